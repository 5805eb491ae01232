"""CPU suite: the SST oracle (oracle/sst_ref.py) is pinned to the imported reference's outputs (tests/golden/sst.npz,
oracle/gen_golden_sst.py) before any HIP kernel is compared with it."""
import os

import numpy as np
import torch

from oracle import sst_ref as S
from oracle import synth

SPARSE, WINDOW = (40, 40, 32), (8, 8, 8)


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'sst.npz'))


def _state_dict(gold):
    shapes = {k: tuple(int(v) for v in s.split(',')) for k, s in zip(gold['param_names'].tolist(), gold['param_shapes'].tolist())}
    return synth.synth_state_dict(shapes, seed=7)


def test_window_partition_and_pos_embed_vs_reference_golden(golden_dir):
    gold = _gold(golden_dir)
    coors = torch.from_numpy(gold['coors'])
    for i in range(2):
        win, ciw = S.window_ids(coors, SPARSE, WINDOW, i == 1)
        assert np.array_equal(win.numpy(), gold[f'batch_win_inds_shift{i}'])
        assert np.array_equal(ciw.numpy(), gold[f'coors_in_win_shift{i}'])
        pe = S.pos_embed(ciw, WINDOW, 128)
        assert np.abs(pe[::4].numpy() - gold[f'pos_flat_shift{i}']).max() < 1e-6


def test_encoder_layer_and_backbone_vs_reference_golden(golden_dir):
    gold = _gold(golden_dir)
    sd = _state_dict(gold)
    coors, feats = torch.from_numpy(gold['coors']), torch.from_numpy(gold['feats'])
    win, ciw = S.window_ids(coors, SPARSE, WINDOW, False)
    pre = 'block_list.0.encoder_list.0.'
    P = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    one = S.encoder_layer(feats, S.pos_embed(ciw, WINDOW, 128), win, P)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    e1 = rel(one.numpy(), gold['one_layer'])
    out = S.sst_blocks(feats, coors, sd, SPARSE, WINDOW)
    e2 = rel(out.numpy(), gold['out'])
    print(f'oracle (no rounding) vs imported reference: one layer {e1:.2e}, two shifted blocks {e2:.2e}')
    assert e1 < 1e-5 and e2 < 1e-5


def test_written_out_backward_is_the_gradient_of_the_forward(golden_dir):
    gold = _gold(golden_dir)
    sd = _state_dict(gold)
    coors, feats = torch.from_numpy(gold['coors'])[:600], torch.from_numpy(gold['feats'])[:600]
    win, ciw = S.window_ids(coors, SPARSE, WINDOW, False)
    pos = S.pos_embed(ciw, WINDOW, 128)
    pre = 'block_list.0.encoder_list.1.'
    P = {k[len(pre):]: v.double().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(600, 128, generator=g, dtype=torch.float64)
    for act in ('gelu', 'relu'):
        x = feats.double().requires_grad_(True)

        y = S.encoder_layer(x, pos, win, P, act=act, detach=False)
        names = list(P)
        grads = torch.autograd.grad(y, [x] + [P[n] for n in names], dy)
        _, c = S.encoder_layer(x.detach(), pos, win, {k: v.detach() for k, v in P.items()}, act=act, keep=True)
        mine = S.encoder_layer_backward(dy, c)
        assert float((mine['dx'] - grads[0]).abs().max()) < 1e-9 * float(grads[0].abs().max())
        for n, gr in zip(names, grads[1:]):
            assert float((mine[n] - gr).abs().max()) <= 1e-9 * float(gr.abs().max()) + 1e-12, n


def test_window_attention_core_equals_autograd_and_its_rounded_form_stays_close():
    """the attention core of the oracle (forward and hand-written backward) against torch autograd on the same float64
    formula; the 'window' rounding mode (bf16 at the per-window kernels' store points) moves it by a few bf16 steps"""
    from oracle import sst_ref
    g = torch.Generator().manual_seed(5)
    nW, T, H, D = 6, 23, 8, 16
    q, k, v, do = (torch.randn(nW, T, H * D, generator=g).bfloat16().double() for _ in range(4))
    key_len = torch.tensor([23, 1, 7, 16, 22, 10])
    o, dq, dk, dv = sst_ref.window_attention_core(q, k, v, key_len, H, dout=do)
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum('wthd,wshd->whts', qa.view(nW, T, H, D), ka.view(nW, T, H, D)) * D ** -0.5
    mask = torch.arange(T)[None, :] >= key_len[:, None]
    ref = torch.einsum('whts,wshd->wthd', torch.softmax(s.masked_fill(mask[:, None, None, :], float('-inf')), -1),
                       va.view(nW, T, H, D)).reshape(nW, T, H * D)
    ref.backward(do)
    for got, exp in ((o, ref.detach()), (dq, qa.grad), (dk, ka.grad), (dv, va.grad)):
        assert float((got - exp).abs().max()) < 1e-12
    ow, dqw, dkw, dvw = sst_ref.window_attention_core(q, k, v, key_len, H, dout=do, rounding='window')
    for got, exp in ((ow, o), (dqw, dq), (dkw, dk), (dvw, dv)):
        rel = float((got - exp).norm() / exp.norm())
        assert 1e-5 < rel < 6e-3, rel          # rounded, and by bf16 steps only


def test_cosine_encoder_layer_vs_reference_golden(golden_dir):
    """the oracle's scaled-cosine attention (cosine=(tau, tau_min)) without roundings equals the imported reference's
    EncoderLayer with layer_cfg cosine / non_shared_tau (tests/golden/sst.npz cos_out, oracle/gen_golden_sst.py;
    CosineMultiheadAttention, cosine_msa.py:123-185) -- before the HIP paths are compared with its rounded forms"""
    gold = _gold(golden_dir)
    shapes = {k: tuple(int(v) for v in s.split(',')) for k, s in zip(gold['cos_param_names'].tolist(), gold['cos_param_shapes'].tolist())}
    sd = synth.synth_state_dict(shapes, seed=9)
    tau = torch.linspace(0.05, 0.4, 8)
    coors, feats = torch.from_numpy(gold['coors']), torch.from_numpy(gold['feats'])
    win, ciw = S.window_ids(coors, SPARSE, WINDOW, False)
    P = {k: v for k, v in sd.items() if k != 'win_attn.self_attn.tau'}
    out = S.encoder_layer(feats, S.pos_embed(ciw, WINDOW, 128), win, P, cosine=(tau, 0.01))
    err = float(np.abs(out.numpy() - gold['cos_out']).max() / np.abs(gold['cos_out']).max())
    print(f'cosine oracle (no rounding) vs imported reference: {err:.2e}')
    assert err < 1e-5
    # a shared tau is the same thing with one value
    one = S.encoder_layer(feats[:500], S.pos_embed(ciw[:500], WINDOW, 128), win[:500], P, cosine=(torch.tensor([0.2]), 0.01))
    per = S.encoder_layer(feats[:500], S.pos_embed(ciw[:500], WINDOW, 128), win[:500], P, cosine=(torch.full((8,), 0.2), 0.01))
    assert torch.equal(one, per)
