"""Pure-PyTorch CPU restatement of the configs[1] occupancy-encoder step, all host cores -- the timed
``cpu_baseline`` of bench.py ("our CPU restatement, pure PyTorch CPU, fp32,
torch.set_num_threads(os.cpu_count())").  TEST INFRASTRUCTURE ONLY.

It is the reference's CPU path written out with the tensor library the reference itself runs on:
  dynamic voxelise   floor((p - min) / voxel), clamped            mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41
  DynamicScatter     torch.unique(dim=0) + index_add / counts      mmdet3d/ops/voxel/src/scatter_points_cuda.cu:199-241
  rulebook           the C oracle (geometry.h:247-297 restated, pinned to the reference's own code by
                     tests/golden/rulebook.npz)
  indiceConv         centre offset = torch.mm; every other offset: index_select -> torch.mm -> index_add_
                     (the reference: SparseGatherFunctor -> torch::mm_out -> SparseScatterAddFunctor,
                     spconv_ops.h:300-354); backward through autograd = the same three steps transposed (:363-456)
  LN + GELU          torch.nn.functional.layer_norm / gelu          ops/sparse_block.py:216-289
  optimizer          torch.optim.AdamW                              configs/_base_/schedules/cosine_2x.py:2-8
fp32 throughout, as the reference trains."""
import numpy as np
import torch

from . import oracle as O


class _IndiceConv(torch.autograd.Function):
    """One sub-manifold convolution in the reference's per-offset formulation (forward and backward)."""

    @staticmethod
    def forward(ctx, x, w, pin, pout, num):
        kvol = w.shape[0]
        centre = int(np.argmax(num))
        out = torch.mm(x, w[centre])
        for k in range(kvol):
            if k == centre or num[k] == 0:
                continue
            out.index_add_(0, pout[k], torch.mm(x.index_select(0, pin[k]), w[k]))
        ctx.save_for_backward(x, w)
        ctx.rule = (pin, pout, num, centre)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        pin, pout, num, centre = ctx.rule
        dw = torch.zeros_like(w)
        dw[centre] = torch.mm(x.t(), dy)
        dx = torch.mm(dy, w[centre].t())
        for k in range(w.shape[0]):
            if k == centre or num[k] == 0:
                continue
            xg, dyg = x.index_select(0, pin[k]), dy.index_select(0, pout[k])
            dw[k] = torch.mm(xg.t(), dyg)
            dx.index_add_(0, pin[k], torch.mm(dyg, w[k].t()))
        return dx, dw, None, None, None


def geometry(xyz, feats, batch_idx, batch_size, voxel_size=(0.2, 0.2, 0.2), rng=(-4, -4, -4, 4, 4, 4)):
    vs = torch.tensor(voxel_size)
    lo = torch.tensor(rng[:3])
    grid = [int(round((rng[3 + i] - rng[i]) / voxel_size[i])) for i in range(3)]
    c = torch.floor((xyz - lo) / vs).to(torch.int64)
    c = torch.minimum(torch.maximum(c, torch.zeros(3, dtype=torch.int64)), torch.tensor(grid) - 1)
    coors = torch.stack([batch_idx.long(), c[:, 2], c[:, 1], c[:, 0]], 1)
    vcoors, inv, counts = torch.unique(coors, dim=0, return_inverse=True, return_counts=True)
    vfeats = torch.zeros(vcoors.shape[0], feats.shape[1]).index_add_(0, inv, feats) / counts[:, None]
    pairs, num = O.subm_rulebook(vcoors.numpy().astype(np.int32), batch_size, grid[::-1])
    pin = [torch.from_numpy(pairs[k, 0, :num[k]].astype(np.int64)) for k in range(27)]
    pout = [torch.from_numpy(pairs[k, 1, :num[k]].astype(np.int64)) for k in range(27)]
    return vfeats, vcoors, pin, pout, num


class EncoderCPU(torch.nn.Module):
    def __init__(self, weights, gammas, betas, eps=1e-3):
        super().__init__()
        self.w = torch.nn.ParameterList([torch.nn.Parameter(torch.as_tensor(w, dtype=torch.float32).reshape(27, w.shape[-2], w.shape[-1]).clone())
                                         for w in weights])
        self.g = torch.nn.ParameterList([torch.nn.Parameter(torch.as_tensor(g, dtype=torch.float32).clone()) for g in gammas])
        self.b = torch.nn.ParameterList([torch.nn.Parameter(torch.as_tensor(b, dtype=torch.float32).clone()) for b in betas])
        self.eps = eps

    def forward(self, xyz, feats, batch_idx, batch_size):
        h, _, pin, pout, num = geometry(xyz, feats, batch_idx, batch_size)
        for w, g, b in zip(self.w, self.g, self.b):
            h = _IndiceConv.apply(h, w, pin, pout, num)
            h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(h, (h.shape[1],), g, b, self.eps))
        return h


def make_step(xyz, feats, batch_idx, batch_size, weights, gammas, betas, lr=1e-4):
    """-> step(): one fwd + bwd (from a fixed upstream gradient, as bench.py) + AdamW on the CPU."""
    model = EncoderCPU(weights, gammas, betas)
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    with torch.no_grad():
        n = model(xyz, feats, batch_idx, batch_size).shape[0]
    d = torch.randn(n, model.w[-1].shape[-1], generator=torch.Generator().manual_seed(1234)) / n

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(xyz, feats, batch_idx, batch_size)
        out.backward(d)
        opt.step()
        return out
    return step, model
