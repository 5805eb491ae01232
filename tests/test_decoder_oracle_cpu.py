"""oracle/decoder_ref.py pinned against the reference's own outputs: tests/golden/ococc_head.npz holds the logits the
imported reference's OccDecoder.occ_forward produced (oracle/gen_golden_ococc.py) for name-hashed weights."""
import os

import numpy as np
import torch

from oracle import decoder_ref as D
from oracle import synth

PREFIX = 'occ_ae_head.occ_decoder.'
SHAPES = {'ln.weight': (1536,), 'ln.bias': (1536,),
          'conv_occ.0.0.weight': (512, 1596), 'conv_occ.0.1.weight': (512,), 'conv_occ.0.1.bias': (512,),
          'conv_occ.1.0.weight': (1024, 512), 'conv_occ.1.1.weight': (1024,), 'conv_occ.1.1.bias': (1024,),
          'conv_occ.2.0.weight': (1024, 1024), 'conv_occ.2.1.weight': (1024,), 'conv_occ.2.1.bias': (1024,),
          'conv_occ.3.weight': (1, 1024), 'conv_occ.3.bias': (1,)}


def decoder_params():
    return synth.synth_state_dict({PREFIX + k: s for k, s in SHAPES.items()}, seed=0)


def test_decoder_oracle_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'ococc_head.npz'))
    P = decoder_params()
    feats = torch.from_numpy(gold['out_fused_roi_feats'])
    xyz = torch.from_numpy(gold['dec_xyz'])
    R, K, _ = xyz.shape
    idx = torch.arange(R).repeat_interleave(K)
    logits = D.decoder(P, PREFIX, feats, xyz.reshape(-1, 3), idx).view(R, K, 1).numpy()
    ref = gold['dec_logits']
    assert np.abs(logits - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    # the bf16 roundings of the fused kernels move the logits by bf16-sized amounts, not more
    lo = D.decoder(P, PREFIX, feats, xyz.reshape(-1, 3), idx, rounding='bf16').view(R, K, 1).numpy()
    assert 1e-5 < np.abs(lo - ref).max() < 3e-2 * max(1.0, np.abs(ref).max())


def test_pos_encode_oracle_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'ococc_head.npz'))
    pe = D.pos_encode(torch.from_numpy(gold['posenc_in']).reshape(-1, 3)).numpy()
    assert np.allclose(pe.reshape(gold['posenc_out'].shape), gold['posenc_out'], atol=1e-6)
