#!/usr/bin/env python3
"""Training entry point with the CLI of the reference's tools/train.py (:27-99):
    python tools/train.py <config> [--work-dir DIR] [--resume-from CKPT] [--no-validate]
                          [--cfg-options k=v ...] [--launcher {none,pytorch}] [--seed N]
One process per GPU (torchrun-style env), RCCL gradient all-reduce (bf16 buckets launched from gradient hooks, so
they overlap the backward pass), and the optimisation recipe of configs/_base_/schedules/cosine_2x.py +
configs/ococc/ococcnet.py:468-478: AdamW (betas 0.9/0.999, weight decay 0.05, none on parameters whose name contains
"norm"), gradient clipping at norm 10, the cyclic learning-rate policy (x100 over the first tenth of the run, then down
to x1e-3, cosine segments) applied every iteration.  Checkpoints carry the reference's parameter names plus the
optimizer state and the iteration, so --resume-from continues the moments and the schedule.

Data: by default Waymo-shaped synthetic tracklets generated in memory; with --data-root DIR the tracklet dataset
and the ococcnet.py train pipeline (objectcentricocccompletion_amd/dataset.py, pipelines.py) read files in the
reference's on-disk formats (tools/make_synthetic_dataset.py writes a small set of them; a real
data/waymo tree prepared by the reference's converters has the same layout), sharded over the ranks."""
import argparse
import os

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')  # before the HIP runtime loads: objectcentricocccompletion_amd/graph.py
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser(description='Train OcOccNet (MI355X)')
    ap.add_argument('config')
    ap.add_argument('--work-dir', default='work_dirs/ococcnet')
    ap.add_argument('--resume-from')
    ap.add_argument('--no-validate', action='store_true')
    ap.add_argument('--cfg-options', nargs='+', default=[])
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('--iters', type=int, default=20, help='iterations to run in this invocation')
    ap.add_argument('--max-iters', type=int, default=None,
                    help='length of the whole run the learning-rate cycle spans (default: start + --iters)')
    ap.add_argument('--synthetic', action='store_true', default=True)
    ap.add_argument('--data-root', default=None, help='tree with tracklet_data/*.pkl, poses.pkl, occ_gt/ (see the docstring)')
    ap.add_argument('--proposals', default='tracklet_data/synth_training.pkl')
    ap.add_argument('--candidates', default='tracklet_data/synth_training_gt_candidates.pkl')
    ap.add_argument('--occ-root', default='occ_gt')
    return ap.parse_args()


def main():
    args = parse_args()
    from objectcentricocccompletion_amd import config, heads, point_pool, roi_head  # noqa: F401
    from objectcentricocccompletion_amd.dist import GradBuckets, broadcast_parameters, init_dist
    from objectcentricocccompletion_amd.registry import DETECTORS
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    import ast
    from objectcentricocccompletion_amd.optim import AdamW, cyclic_lr, param_groups_from_cfg
    from objectcentricocccompletion_amd.sir import check_barriers
    cfg = config.fromfile(args.config)
    opts = {}
    for kv in args.cfg_options:
        k, v = kv.split('=', 1)
        try:
            v = ast.literal_eval(v)   # numbers, lists, dicts, True/False/None; anything else stays a string
        except (ValueError, SyntaxError):
            pass
        opts[k] = v
    config.merge_from_dict(cfg, opts)
    rank, world, local_rank = init_dist() if args.launcher == 'pytorch' else (0, 1, 0)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(args.seed)
    model = DETECTORS.build(cfg['model']).to(dev)
    # optimizer: cosine_2x.py:2-8 merged with the lr override of ococcnet.py:468-470
    ocfg = dict(type='AdamW', lr=1e-5, betas=(0.9, 0.999), weight_decay=0.05,
                paramwise_cfg=dict(custom_keys={'norm': dict(decay_mult=0.)}))
    ocfg.update(cfg.get('optimizer', {}))
    assert ocfg.pop('type') == 'AdamW', 'the recipe is AdamW'
    base_lr, wd = float(ocfg['lr']), float(ocfg['weight_decay'])
    groups = param_groups_from_cfg(model.named_parameters(), wd, ocfg.get('paramwise_cfg'))
    opt = AdamW(groups, lr=base_lr, betas=tuple(ocfg['betas']), weight_decay=wd, device_lr=True)
    lr_cfg = dict(policy='cyclic', target_ratio=(100, 1e-3), cyclic_times=1, step_ratio_up=0.1)
    lr_cfg.update(cfg.get('lr_config') or {})
    assert lr_cfg.pop('policy') == 'cyclic', 'the recipe is the cyclic policy'
    clip = (cfg.get('optimizer_config') or {}).get('grad_clip', dict(max_norm=10, norm_type=2))
    start = 0
    if args.resume_from:
        ck = torch.load(args.resume_from, map_location=dev)
        model.load_state_dict(ck['state_dict'])
        start = ck.get('meta', {}).get('iter', 0)
        if 'optimizer' in ck:
            opt.load_state_dict(ck['optimizer'])
    broadcast_parameters(model)
    opt.init_state()
    saved_max = ck.get('meta', {}).get('max_iters') if args.resume_from else None
    max_iters = args.max_iters or saved_max or (start + args.iters)   # a resumed run stays on the ONE cycle it began
    if args.max_iters and saved_max and args.max_iters != saved_max and rank == 0:
        print(f'warning: --max-iters {args.max_iters} differs from the checkpoint\'s {saved_max}: the cyclic LR '
              f'schedule changes shape mid-run', flush=True)
    buckets = GradBuckets(model.parameters(), overlap=True)
    samples = cfg.get('data', {}).get('samples_per_gpu', 4)
    ds = None
    if args.data_root:
        import numpy as np
        from objectcentricocccompletion_amd import dataset  # noqa: F401 (registers)
        from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_train_pipeline
        from objectcentricocccompletion_amd.pipelines import collate_tracklets
        from objectcentricocccompletion_amd.registry import DATASETS
        j = lambda p: os.path.join(args.data_root, p)
        ds = DATASETS.build(dict(type='WaymoTrackletDatasetWithOcc', data_root=args.data_root, ann_file=j(args.candidates),
                                 tracklet_proposals_file=j(args.proposals), occ_anno_root=j(args.occ_root),
                                 pose_file=j('poses.pkl'), pipeline=ococcnet_train_pipeline(), classes=['Car'],
                                 min_tracklet_points=100, min_tracklet_length=32))
        np.random.seed(args.seed + rank)   # the augmentations of the pipeline draw from the global numpy state: per rank
        per_step = world * samples
        if rank == 0:
            print(f'{len(ds)} tracklets under {args.data_root}', flush=True)
    model.train()
    for it in range(start, start + args.iters):
        if ds is not None:
            # tracklets sharded over the ranks: ONE permutation per epoch, the same on every rank
            # (RandomState(seed + epoch)), of which rank r takes its `samples` slots of every step
            pos = it * per_step
            epoch, off = divmod(pos, max(len(ds) - len(ds) % per_step, per_step))
            order = np.random.RandomState(args.seed + 1000003 * epoch).permutation(len(ds))
            idx = [int(order[(off + rank * samples + b) % len(ds)]) for b in range(samples)]
            batch = collate_tracklets([ds[i] for i in idx], dev)
        else:
            batch = synthetic_training_batch(samples, 32, seed=args.seed + it * world + rank, device=dev)
        t0 = time.perf_counter()
        lr = cyclic_lr(base_lr, it, max_iters, **lr_cfg)
        for gi, g in enumerate(opt.param_groups):
            opt.set_lr(lr * g.get('lr_mult', 1.0), gi)
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        total = sum(v.mean() for k, v in losses.items() if k.startswith('loss'))
        total.backward()
        buckets.all_reduce()
        if clip:
            torch.nn.utils.clip_grad_norm_(model.parameters(), **clip)
        opt.step()
        check_barriers()   # (no synchronisation: a host load; raises when a one-launch SIR layer's grid barrier gave up)
        if rank == 0:
            torch.cuda.synchronize()
            print(f'iter {it + 1}: lr {lr:.3e} loss {float(total):.4f} cls {float(losses["loss_rcnn_cls"]):.4f} '
                  f'bbox {float(losses["loss_rcnn_bbox"]):.4f} occ {float(losses["loss_rcnn_occ"].mean()):.4f} '
                  f'({(time.perf_counter() - t0) * 1e3:.1f} ms)', flush=True)
    torch.cuda.synchronize()
    check_barriers()   # nothing is written from a run whose last layers could not gather their grids
    if rank == 0:
        os.makedirs(args.work_dir, exist_ok=True)
        torch.save(dict(state_dict=model.state_dict(), optimizer=opt.state_dict(),
                        meta=dict(iter=start + args.iters, max_iters=max_iters)),
                   os.path.join(args.work_dir, 'latest.pth'))


if __name__ == '__main__':
    main()
