#!/usr/bin/env python3
"""Host-side profile of the configs[2] training step (the step is host-bound at B = 4: ~1.8 k launches behind ~16 ms of
kernels).  cProfile over a few steps, printed by cumulative and by own time -- where the Python / dispatcher time
goes, module by module.

usage: python tools/host_profile_b4.py [tracklets=4] [steps=10]"""
import cProfile
import io
import os

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')  # before the HIP runtime loads: objectcentricocccompletion_amd/graph.py
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from objectcentricocccompletion_amd import heads, point_pool, roi_head  # noqa: F401
    from objectcentricocccompletion_amd.occ.occ_base import OccDecoder
    from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
    from objectcentricocccompletion_amd.optim import AdamW
    from objectcentricocccompletion_amd.registry import DETECTORS
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = ococcnet_model_cfg()
    cfg['train_cfg']['random_shift_frame_inds'] = False
    model = DETECTORS.build(cfg).to(dev).train()
    for m in model.modules():
        if isinstance(m, OccDecoder):
            m.compute_dtype = torch.bfloat16
    params = [p for p in model.parameters() if p.requires_grad]
    opt = AdamW(params, lr=1e-6)
    batch = synthetic_training_batch(B, 32, pts_per_frame=64, occ_queries=512, seed=0, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
        total.backward()
        opt.step()

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(steps):
        t = time.perf_counter()
        step()
        per.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    per.sort()
    print(f'unprofiled: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step   (host issue time per step: median '
          f'{per[len(per) // 2] * 1e3:.2f}, min {per[0] * 1e3:.2f} ms)')
    # host time the phases take to ISSUE (no synchronisation inside) and device time between their event marks
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    host = [0.0, 0.0, 0.0]
    torch.cuda.synchronize()
    for i in range(steps):
        t = time.perf_counter()
        ev[i][0].record()
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
        ev[i][1].record()
        t1 = time.perf_counter()
        total.backward()
        ev[i][2].record()
        t2 = time.perf_counter()
        opt.step()
        ev[i][3].record()
        t3 = time.perf_counter()
        host[0] += t1 - t
        host[1] += t2 - t1
        host[2] += t3 - t2
    torch.cuda.synchronize()
    dev_ms = [sum(e[j].elapsed_time(e[j + 1]) for e in ev) / steps for j in range(3)]
    print('host ms  fwd %.2f  bwd %.2f  opt %.2f   |  device span ms  fwd %.2f  bwd %.2f  opt %.2f' % (
        host[0] / steps * 1e3, host[1] / steps * 1e3, host[2] / steps * 1e3, *dev_ms))
    # forward and backward apart (the backward runs in autograd's thread: cProfile sees only its Python callbacks
    # through the main thread's run_backward call)
    if os.environ.get('OCOCC_HOST_PROFILE') == 'torch':   # the autograd thread too: per-operator / per-node CPU time
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=60, max_name_column_width=70))
        print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=60, max_name_column_width=70))
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    for key, n in (('cumulative', 70), ('tottime', 45)):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(n)
        print(s.getvalue()[:14000])


if __name__ == '__main__':
    main()
