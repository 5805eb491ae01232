"""The occupancy decoder's training forward + backward at ~1 M query rows (configs[2] at 64 tracklets: 2048 RoIs x 512
queries), kernel by kernel: the one-launch forward, the one-launch backward (csrc/mlp_layer.hip: occ_mlp_bwd_kernel) and
the weight-gradient contractions, in the three backward modes of occ/fused_mlp.py: 'fused' (one launch on what the
forward parked), 'recompute' (one launch, nothing saved by the forward) and 'chain' (the forward leaves z / statistics /
y row-major, the backward is the operator chain).  usage: python tools/probe/decoder_train_bench.py [rows] [dropout]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.occ import fused_mlp as fm  # noqa: E402


class Probe(object):
    def __init__(self):
        self.items = []

    def wrap(self, name, flops, launch):
        a, b = L.Timer(), L.Timer()
        a.record()
        launch()
        b.record()
        self.items.append((name, a, b, flops))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    K = 2048
    W = [(torch.randn(n, k, device=dev) / k ** 0.5).requires_grad_(True) for k, n in ((60, 512), (512, 1024), (1024, 1024))]
    gam = [torch.ones(n, device=dev, requires_grad=True) for n in (512, 1024, 1024)]
    bet = [torch.zeros(n, device=dev, requires_grad=True) for n in (512, 1024, 1024)]
    pe = torch.randn(rows, 64, device=dev).to(torch.bfloat16)
    pe[:, 60:] = 0
    roi = torch.randn(K, 512, device=dev, requires_grad=True)
    idx = (torch.arange(rows, device=dev, dtype=torch.int32) // (rows // K + 1)).contiguous()
    hw, hb = torch.randn(1, 1024, device=dev, requires_grad=True), torch.zeros(1, device=dev, requires_grad=True)
    dl = torch.randn(rows, 1, device=dev) / rows
    thr = int(round(p * 65536))
    for mode in ('fused', 'recompute', 'chain'):
        fm.BACKWARD_MODE = mode
        cache = fm.DecoderWeights()

        def step():
            out = fm.occ_mlp_train(pe, roi, idx, W[0], W[1], W[2], gam, bet, 1e-3, hw, hb, thr, (11, 22, 33), cache)
            out.backward(dl)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        probe = Probe()
        fm.set_probe(probe)
        t0 = time.perf_counter()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 5
        for _ in range(n):
            step()
        b.record()
        torch.cuda.synchronize()
        fm.set_probe(None)
        per = {}
        for name, x, y, f in probe.items:
            t = per.setdefault(name, [0.0, 0.0, 0])
            t[0] += x.elapsed_ms(y)
            t[1] += f
            t[2] += 1
        print(f'backward mode {mode} rows={rows} dropout={p}: forward + backward {a.elapsed_time(b) / n:7.3f} ms per step '
              f'(host {1e3 * (time.perf_counter() - t0) / n:7.3f} ms)')
        for k, t in per.items():
            print(f'    {k:34s} {t[0] / t[2]:7.3f} ms  {t[1] / t[0] / 1e9:7.1f} TF/s')


if __name__ == '__main__':
    main()
