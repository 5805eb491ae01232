#!/usr/bin/env python3
"""Benchmark of the OcOccNet hot path on MI355X.

Workload (BASELINE.json configs[1]): 64 synthetic object grids per GPU, 0.2 m voxels,
40^3 cells, 2000 random points each; SubMConv3d-only occupancy encoder
(voxelise -> scatter-mean -> rulebook -> 3 x [SubMConv3d 3^3 -> LN -> GELU], channels
16->32->64->128), bf16 features, forward + backward + optimizer step.
Metric: object-grids / second (whole job, all ranks).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; tracklets (object grids) are sharded across ranks, the only
collective is the gradient all-reduce over RCCL ("weak" scaling).  Rank 0 prints ONE JSON
line with the contract fields plus "roofline" (dominant kernel, HIP-event timed inside
the timed region) and "cpu_baseline" (the oracle port on one host core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

T_START = time.time()

# ROCm 7.2's graph "packet capture" fast path faults when device memory is allocated between two
# replays of a large graph (tools/graph_bisect.py); must be off before the HIP runtime loads.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRIDS_PER_GPU = 64
POINTS_PER_GRID = 2000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The roofline object is reported on the LONGEST convolution kernel of the step: every conv launch is bracketed with HIP
# events (spconv.ops.KernelProbe, keyed family<kd,ncols>) and the key with the largest average launch time is the one
# reported, with its own algorithmic bytes (conv_algorithmic_bytes).  `traffic` comes from the committed PMC passes of
# that kernel (tools/pmc_hbm.sh), keyed the same way; null for a kernel no PMC pass exists for.
PMC_JSON = {'stationary<64,128>': 'r02_pmc_gather_gemm_stream_64_128.json',
            'sorted<64,128>': 'r04_pmc_gather_gemm_sorted_64_128.json',
            'sorted_ln<64,128>': 'r06_pmc_sorted_ln_64_128.json',
            'sorted_lnbwd<128,64>': 'r05_pmc_sorted_lnbwd_128_64.json',
            'tile_lnbwd<128,64>': 'r06_pmc_tile_lnbwd_128_64.json'}
KERNEL_NAMES = {'stationary': 'gather_gemm_stream_kernel', 'stationary_ln': 'gather_gemm_kernel (+LN epilogue)',
                'sorted': 'gather_gemm_sorted_kernel', 'sorted_ln': 'gather_gemm_sorted_kernel (+LN epilogue)', 'sorted_lnbwd': 'gather_gemm_sorted_kernel (+LN-backward epilogue)',
                'tile': 'subm_tile_conv_kernel', 'tile_ln': 'subm_tile_conv_kernel (+LN epilogue)',
                'tile_lnbwd': 'subm_tile_conv_kernel (+LN-backward epilogue)'}


def conv_algorithmic_bytes(key, n_vox, n_pairs):
    """compulsory bytes of one sub-manifold convolution launch, Nact*kd*s + Nact*nc*s + P*8 + 27*kd*nc*s
    with s = 2 (bf16) -- plus what a fused epilogue must move: LN forward writes the activation beside the conv output
    and the row statistics (Nact*nc*2 + Nact*8); LN backward reads the block's conv output and statistics
    (Nact*nc*2 + Nact*8).  Returns (bytes, description)."""
    family, dims = key.split('<')
    kd, nc = (int(v) for v in dims.rstrip('>').split(','))
    conv = n_vox * kd * 2 + n_vox * nc * 2 + n_pairs * 8 + 27 * kd * nc * 2
    extra, note = 0, ''
    if family.endswith('_ln'):
        extra, note = n_vox * nc * 2 + n_vox * 8, ' + LN/GELU epilogue: writes the activation and row statistics'
    elif family.endswith('_lnbwd'):
        extra, note = n_vox * nc * 2 + n_vox * 8, " + LN-backward epilogue: reads the block's conv output and row statistics"
    return conv + extra, 'gathers %d channels, writes %d channels%s' % (kd, nc, note)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed steps (default 500 for the 0.35 ms submconv step -- a 17 ms timed region of 50 steps '
                         'moved by 10 %% with one scheduling hiccup of the host -- and 50 for the other workloads)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default 20 / 10)')
    ap.add_argument('--grids', type=int, default=GRIDS_PER_GPU)
    ap.add_argument('--points', type=int, default=POINTS_PER_GRID)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the short runs of the other workloads (ococcnet at 4 and 64 tracklets, sst) whose JSON lines '
                         'the default run attaches under "also"')
    ap.add_argument('--workload', default='submconv', choices=['submconv', 'ococcnet', 'sst', 'decode'],
                    help='submconv = BASELINE.json configs[1] (the quoted metric); ococcnet = configs[2]; sst = configs[4] per-GPU share')
    ap.add_argument('--tracklets', type=int, default=4)
    ap.add_argument('--sst-grids', type=int, default=32,
                    help='sst workload: object grids on this GPU (32 = one GPU\'s share of configs[4]\'s 256 objects over 8 GPUs; 256 = the\n'
                         'whole configs[4] batch on one GPU)')
    ap.add_argument('--f32-decoder', action='store_true',
                    help='ococcnet workload: keep the occupancy-decoder MLP in f32 (default: bf16 GEMMs, f32 accumulate)')
    ap.add_argument('--split-graph', action='store_true',
                    help='use the N>1 launch plan (fwd+bwd graph, eager all-reduce, optimizer graph) at N=1 too')
    ap.add_argument('--pipeline', action='store_true',
                    help='compute the NEXT batch\'s geometry on a forked stream of the step\'s graph instead of at the head of '
                         'its own step (graph.PipelinedStep; measured no faster: the geometry kernels contend with the '
                         'convolutions they run beside, 0.346-0.360 vs 0.335 ms/step)')
    ap.add_argument('--no-graph', action='store_true',
                    help='launch every kernel eagerly from Python instead of replaying the captured HIP graph')
    ap.add_argument('--dump-params', default=None,
                    help='test hook: every rank saves its parameters (one flat f32 vector) to <path>.rank<r>.pt when the timed '
                         'steps are done, i.e. after exactly warmup + steps optimizer steps (tests/test_gpu_bench_ranks.py)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 500 if args.workload == 'submconv' else 50
    if args.warmup is None:
        args.warmup = 20 if args.workload == 'submconv' else 10
    return args


def dump_params(args, rank, params):
    if args.dump_params:
        torch.cuda.synchronize()
        torch.save(torch.cat([p.detach().float().reshape(-1) for p in params]).cpu(), f'{args.dump_params}.rank{rank}.pt')


def host_cores(limit=16):
    """Threads the CPU baseline may use: the smallest of the process's CPU affinity, its cgroup CPU quota and `limit`.
    (os.cpu_count() reports the machine's hardware threads; a box that hands this process a few of them -- cpuset or
    cpu.max -- makes torch's intra-op pool with 16 spinning threads two orders of magnitude slower than with 4.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, limit))


def cpu_baseline(sample_grids, points, model):
    """the pure-PyTorch CPU restatement of the same step (oracle/encoder_torch_cpu.py: the reference's
    CPU formulation -- per-offset gather / torch.mm / scatter-add, fp32 -- on every host core) on a bounded sample of
    the same workload; baseline only."""
    from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
    from oracle.encoder_torch_cpu import make_step
    # at most 16 threads: the step is a few hundred small operators; with the 256 hardware threads of the GPU box
    # torch's intra-op pool spends its time handing them out (measured there: 0.06 grids/s with 256 threads, 4.5
    # minutes per step, against 119 grids/s with 8 threads in the build container)
    cores = host_cores()
    torch.set_num_threads(cores)
    xyz, feats, bidx = synthetic_object_grids(sample_grids, points, seed=0, device='cpu')
    ws = [l[0].weight.detach().float().cpu() for l in model.conv_layers]
    gs = [l[1].weight.detach().float().cpu() for l in model.conv_layers]
    bs = [l[1].bias.detach().float().cpu() for l in model.conv_layers]
    step, _ = make_step(xyz, feats, bidx, sample_grids, ws, gs, bs)
    t0 = time.perf_counter()
    step()  # warm-up (page-in, thread pool) -- and the measurement itself if the host is this slow
    first = time.perf_counter() - t0
    reps, t0 = 0, time.perf_counter()
    while first < 8.0 and (reps < 3 or time.perf_counter() - t0 < 10.0):
        step()
        reps += 1
        if time.perf_counter() - t0 > 20.0:
            break
    if reps == 0:
        reps, t0 = 1, time.perf_counter() - first
    dt = (time.perf_counter() - t0) / reps
    return {'value': round(sample_grids / dt, 2), 'unit': 'object-grids/s', 'cores': cores,
            'kind': 'port',
            'sample': f'{sample_grids} of the {GRIDS_PER_GPU} grids x {points} points, geometry + fwd + bwd + AdamW in fp32, '
                      f'{reps} repetitions, torch {torch.__version__.split("+")[0]} CPU with {cores} threads, '
                      'oracle/encoder_torch_cpu.py'}


def pmc_traffic(kernel):
    """HBM bytes per launch of the reported kernel from the committed PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/PMC_JSON[kernel]); None when no pass exists for it."""
    if kernel not in PMC_JSON:
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', PMC_JSON[kernel])) as f:
            return json.load(f)['traffic_bytes_per_launch']
    except Exception:
        return None


def occ_base_fused_train():
    from objectcentricocccompletion_amd.occ import occ_base
    return bool(occ_base.FUSED_MLP and occ_base.FUSED_TRAIN_MLP)


def bench_ococcnet(args, world, rank, dev):
    """configs[2]: full ococcnet.py model on synthetic Waymo-shaped tracklets, B tracklets x 32 frames
    (= B*32 object grids) per GPU per step, K = 512 occupancy queries, fwd + bwd + AdamW, fp32 as the
    reference trains."""
    from objectcentricocccompletion_amd import heads, point_pool, roi_head  # noqa: F401
    from objectcentricocccompletion_amd.dist import GradBuckets, broadcast_parameters
    from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
    from objectcentricocccompletion_amd.registry import DETECTORS
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    torch.manual_seed(0)
    cfg = ococcnet_model_cfg()
    cfg['train_cfg']['random_shift_frame_inds'] = False  # host RNG + in-place shift: keep steps identical
    model = DETECTORS.build(cfg).to(dev).train()
    if not args.f32_decoder:
        from objectcentricocccompletion_amd.occ.occ_base import OccDecoder
        for m in model.modules():
            if isinstance(m, OccDecoder):
                m.compute_dtype = torch.bfloat16
    broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    from objectcentricocccompletion_amd.optim import AdamW
    opt = AdamW(params, lr=1e-6)  # one fused launch over the 269 parameter tensors (6 chunks of 48)
    buckets = GradBuckets(params)
    B, L = args.tracklets, 32
    batch = synthetic_training_batch(B, L, pts_per_frame=64, occ_queries=512, seed=rank, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
        total.backward()
        buckets.all_reduce()
        opt.step()
        return total

    # roofline probe: the occupancy decoder's forward (the FLOP hot spot, SURVEY.md 8a A11: per query point
    # 2 x (60 x 512 + 512 x 1024 + 1024 x 1024 + 1024) = 3.21 MFLOP with the first layer factorised per RoI, plus
    # 2 x 1536 x 512 once per RoI), HIP events on the stream its GEMMs are launched on (torch's current stream)
    decoder = model.roi_head.bbox_head.occ_ae_head.occ_decoder
    dec_events, dec_shapes = [], []
    real_forward = decoder.forward

    def timed_forward(feats, xyz, idx, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_forward(feats, xyz, idx, *a, **k)
        e1.record()
        dec_events.append((e0, e1))
        dec_shapes.append((feats.shape[0], xyz.shape[0]))
        return out

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    decoder.forward = timed_forward
    # (the decoder's own kernels -- the one-launch forward and the one-launch backward with recompute -- bracketed one by one)
    from objectcentricocccompletion_amd.occ import fused_mlp as fm
    kprobe = BlockProbe()
    fm.set_probe(kprobe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    decoder.forward = real_forward
    fm.set_probe(None)
    from objectcentricocccompletion_amd.sir import check_barriers
    check_barriers()   # a line from steps in which a one-launch SIR layer could not gather its grid is not a measurement: raise
    dump_params(args, rank, params)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # kernel time of one step (device-side): HIP events around a step with the queue already full
    torch.cuda.synchronize()
    if rank == 0:
        dec_ms = sum(a.elapsed_time(b) for a, b in dec_events) / max(len(dec_events), 1)
        rois, queries = dec_shapes[0] if dec_shapes else (0, 0)
        flops = queries * 2.0 * (60 * 512 + 512 * 1024 + 1024 * 1024 + 1024) + rois * 2.0 * 1536 * 512
        peak = 157.3 if args.f32_decoder else 2500.0
        res = {
            'metric': 'object-grids/sec (fwd+bwd)', 'value': round(world * B * L * args.steps / dt, 1),
            'unit': 'object-grids/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': ('f32' if args.f32_decoder else 'f32 (occupancy-decoder MLP in bf16)')
            + (' (library products on bf16 operands, f32 accumulation: OCOCC_GEMM_DTYPE=bf16)'
               if os.environ.get('OCOCC_GEMM_DTYPE', '') == 'bf16' else ''),
            'data': 'synthetic',
            'config': {'workload': f'configs[2]: full ococcnet.py model (66.55 M parameters), {B} tracklets x {L} '
                                   f'frames = {B * L} object grids/GPU/step, 64 points/frame, K=512 occupancy '
                                   'queries, fwd+bwd+AdamW, all-reduce of 266 MB gradients at N>1',
                       'grids_per_gpu': B * L, 'parallelism': f'dp{world}', 'launch': 'eager, ~1.0 k launches per step at 4 tracklets (the temporal transformer and the head\'s tail '
                       'replayed as HIP-graph pairs; one launch per SIR layer and direction up to 28 k points)'},
            'roofline': {'kernel': 'OccDecoder forward (MLP 60|1536 -> 512 -> 1024 -> 1024 -> 1 over all query points: '
                                   + ('library f32 GEMMs + LN/GELU kernels)' if args.f32_decoder else
                                      'positional encoding, weight fragments, per-RoI GEMM and the one-launch bf16 MLP kernel in its '
                                      'training instantiation' + (': keeps nothing but the logits, the backward recomputes)'
                                                                  if fm.RECOMPUTE_BACKWARD else
                                                                  ', which also writes z / row statistics / y of every layer -- 10 KB '
                                                                  'per query row)') if occ_base_fused_train()
                                      else 'library bf16 GEMMs + fused LN/GELU kernels)'),
                         'bound': 'mfma', 'achieved': round(flops / (dec_ms * 1e-3) / 1e12, 2) if dec_ms else None,
                         'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(flops / (dec_ms * 1e-3) / 1e12 / peak, 4) if dec_ms else None, 'traffic': None,
                         'algorithmic_flops_per_launch': flops, 'avg_launch_ms': round(dec_ms, 4),
                         'launches_timed': len(dec_events), 'query_points': queries, 'rois': rois,
                         # per own kernel: the forward launch alone, and the backward launch (forward again + two
                         # input-gradient GEMMs per tile: 2 x (60 x 512 + 2 x 512 x 1024 + 2 x 1024 x 1024) flop per query)
                         'per_kernel': {k: dict(v, frac=round(v['tflops'] / peak, 4)) for k, v in probe_summary(kprobe)[2].items()}},
            'cpu_baseline': None}
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline_ococcnet(L)
        print(json.dumps(res), flush=True)


def cpu_baseline_ococcnet(frames):
    """The product's module graph for the same step on the host cores, with its HIP leaf operators swapped for torch /
    oracle restatements (oracle/cpu_port.py, fp32, torch CPU): ONE tracklet per step (a quarter of the GPU batch; the
    CPU path is linear in the tracklets), fwd + bwd + AdamW; baseline only.  At most 16 threads: the step is a few
    thousand small operators, and with the 256 hardware threads of the GPU box torch's intra-op pool spends its time
    handing them out (measured there: 0.11 grids/s with 256 threads)."""
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    from oracle import cpu_port
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = cpu_port.build_detector_cpu(seed_weights=False).train()
    model.roi_head.train_cfg['random_shift_frame_inds'] = False
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-6)
    batch = synthetic_training_batch(1, frames, pts_per_frame=64, occ_queries=512, seed=0, device='cpu')

    def step():
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        (losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()).backward()
        opt.step()

    with cpu_port.cpu_ops():
        step()
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or time.perf_counter() - t0 < 10.0:   # (bounded: one repetition if it is slower than that)
            step()
            reps += 1
        dt = (time.perf_counter() - t0) / reps
    return {'value': round(frames / dt, 2), 'unit': 'object-grids/s', 'cores': cores, 'kind': 'port',
            'sample': f'1 tracklet x {frames} frames ({frames} object grids) per step, fwd+bwd+AdamW in fp32, {reps} '
                      f'repetitions, torch {torch.__version__.split("+")[0]} CPU with {cores} threads, oracle/cpu_port.py'}


class BlockProbe(object):
    """HIP events around the fused forward kernels (recorded on the launch stream)."""

    def __init__(self):
        self.items = []

    def wrap(self, name, flops, launch):
        a, b = _L().Timer(), _L().Timer()
        a.record()
        launch()
        b.record()
        self.items.append((name, a, b, flops))


def _L():
    from objectcentricocccompletion_amd import _lib
    return _lib


def probe_summary(probe):
    """(total ms, total flops, per-kernel detail) of the launches a BlockProbe bracketed"""
    per = {}
    for name, a, b, f in probe.items:
        t = per.setdefault(name, [0.0, 0.0, 0])
        t[0] += a.elapsed_ms(b)
        t[1] += f
        t[2] += 1
    ms = sum(t[0] for t in per.values())
    fl = sum(t[1] for t in per.values())
    detail = {k: {'launches': t[2], 'avg_us': round(t[0] / t[2] * 1e3, 1), 'tflops': round(t[1] / (t[0] * 1e-3) / 1e12, 1)}
              for k, t in per.items() if t[0] > 0}
    return ms, fl, detail


def bench_sst(args, world, rank, dev):
    """configs[4], one GPU's share: 32 object grids of 80x80x64 cells at 0.1 m (the reference's window
    partition asserts z < x, sst_ops.py:283, so the cube is cut to 6.4 m in z; ~8 000 active voxels each),
    voxelise -> scatter-mean -> Linear(16->128) -> SSTInputLayerV2 (3-D windows 8x8x8, drop levels
    30/60/100 tokens) -> 2 BasicShiftBlockV2 (d_model 128, 8 heads, ffn 256) on the fused encoder-layer kernels,
    fwd + bwd + AdamW.  The roofline line is the forward of the encoder layers (attention block + FFN block kernels)
    against the dense bf16 MFMA peak, flops counted on the real tokens and windows."""
    from objectcentricocccompletion_amd.dist import GradBuckets, broadcast_parameters
    from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
    from objectcentricocccompletion_amd.optim import AdamW
    from objectcentricocccompletion_amd.sst import sst_modules as sm
    from objectcentricocccompletion_amd.voxel import dynamic_scatter, voxelization
    torch.manual_seed(0)
    G, P = args.sst_grids, 8200
    shape = (64, 80, 80)   # (D, H, W) = (z, y, x) cells
    rng = [-4, -4, -3.2, 4, 4, 3.2]
    drop = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)),
            2: dict(max_tokens=100, drop_range=(60, 100000))}
    inp = sm.SSTInputLayerV2(drop, (8, 8, 8), (80, 80, 64), shuffle_voxels=False, debug=False, mute=True).to(dev)
    model = sm.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2, dropout=0.0,
                     activation='gelu', num_attached_conv=0, to_bev=False, debug=False,
                     layer_cfg=dict(compute_dtype=torch.bfloat16)).to(dev).train()
    from objectcentricocccompletion_amd.linear import Linear as TallLinear
    embed = TallLinear(16, 128).to(dev)  # voxel encoder stand-in (the package's Linear: row-sliced weight gradient)
    broadcast_parameters(model)
    broadcast_parameters(embed)
    params = list(model.parameters()) + list(embed.parameters())
    opt = AdamW(params, lr=1e-4)
    buckets = GradBuckets(params)
    xyz, feats, bidx = synthetic_object_grids(G, P, seed=rank, device=dev)
    xyz[:, 2] *= 0.8

    from objectcentricocccompletion_amd import _lib as L_

    def step():
        opt.zero_grad(set_to_none=True)
        zyx = voxelization(xyz, [0.1, 0.1, 0.1], rng, -1, -1)
        coors = torch.cat([bidx.view(-1, 1).to(torch.int32), zyx], 1)
        vfeats, vcoors = dynamic_scatter(feats, coors, 'mean', grid_shape=[G] + list(shape))
        info = inp(embed(vfeats), vcoors.long(), batch_size=G)
        out = model(info)[0]['voxel_feats']
        out.backward(d_out)
        buckets.all_reduce()
        opt.step()
        return out

    with torch.no_grad():
        zyx = voxelization(xyz, [0.1, 0.1, 0.1], rng, -1, -1)
        n_act = dynamic_scatter(feats, torch.cat([bidx.view(-1, 1).to(torch.int32), zyx], 1), 'mean',
                                grid_shape=[G] + list(shape))[0].shape[0]
    gen = torch.Generator(device=dev).manual_seed(99 + rank)
    d_out = (torch.randn(n_act, 128, generator=gen, device=dev) / n_act).to(torch.bfloat16)
    for _ in range(args.warmup):
        step()
    from objectcentricocccompletion_amd.sst import fused_block as fb
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # the roofline's kernel times: HIP events around the fused forward launches in a few MORE steps, behind the timed
    # region (two event records per launch inside it cost the step 2 ms of 10: the records serialise the queue)
    probe = BlockProbe()
    fb.set_probe(probe)
    for _ in range(min(args.steps, 10)):
        step()
    torch.cuda.synchronize()
    fb.set_probe(None)
    dump_params(args, rank, params)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        ms, fl, detail = probe_summary(probe)
        tflops = fl / (ms * 1e-3) / 1e12 if ms else None
        print(json.dumps({
            'metric': 'object-grids/sec (fwd+bwd)', 'value': round(world * G * args.steps / dt, 1),
            'unit': 'object-grids/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': ('configs[4] per-GPU share' if G == 32 else 'configs[4], the WHOLE 256-object batch on one GPU' if G == 256
                                    else 'configs[4] shape') + f': {G} grids x 80x80x64 cells at 0.1 m, {P} random points each, '
                                   'SST path (windows 8x8x8, drop levels 30/60/100, d_model 128, 8 heads, ffn 256, '
                                   '2 BasicShiftBlockV2), fwd+bwd+AdamW', 'grids_per_gpu': G, 'active_voxels': int(n_act),
                       'parallelism': f'dp{world}', 'launch': 'eager launches'},
            # forward of an encoder layer = window_attn_block_fwd_kernel + token_ffn_block_fwd_kernel: algorithmic flops of
            # the projections, the attention of every real window and the FFN on the real tokens, over the two kernels' time
            'roofline': {'kernel': 'encoder-layer forward: window_attn_block_fwd_kernel + token_ffn_block_fwd_kernel',
                         'bound': 'mfma', 'achieved': round(tflops, 2) if tflops else None, 'peak': 2500.0,
                         'unit': 'TFLOP/s', 'frac': round(tflops / 2500.0, 5) if tflops else None, 'traffic': None,
                         'launches_timed': len(probe.items), 'per_kernel': detail,
                         'timed_in': 'extra steps behind the timed region'},
            'cpu_baseline': None}), flush=True)


def bench_decode(args, world, rank, dev):
    """SURVEY 8(f) row 2, the step after the path: dense-grid decode of the occupancy of B x 32 RoIs
    (OccDecoder.get_occ, ococcnet decoder 1536 -> 512 -> 1024 -> 1024 -> 1, 0.2 m cells over the box enlarged by
    0.5 m: ~20 k cells per vehicle-sized RoI), inference only, bf16 decoder MLP.  Metric: object grids decoded/s."""
    from objectcentricocccompletion_amd.occ.occ_base import OccDecoder
    torch.manual_seed(0)
    dec = OccDecoder(1536, [512, 1024, 1024], pos_encode_L=10, norm_cfg=dict(type='LN', eps=1e-3), act='gelu',
                     occ_dropout=0.1, use_ln=True).to(dev).eval()
    dec.compute_dtype = None if args.f32_decoder else torch.bfloat16
    g = torch.Generator().manual_seed(7 + rank)
    R = args.tracklets * 32
    rois = torch.zeros(R, 8)
    rois[:, 0] = torch.arange(R) // 32
    rois[:, 1:4] = torch.randn(R, 3, generator=g) * torch.tensor([30., 30., 1.])
    rois[:, 4:7] = torch.rand(R, 3, generator=g) * torch.tensor([0.4, 0.8, 0.4]) + torch.tensor([1.8, 4.2, 1.5])
    rois[:, 7] = (torch.rand(R, generator=g) * 2 - 1) * 3.14
    rois, feats = rois.to(dev), torch.randn(R, 1536, generator=g).to(dev)
    run = lambda: dec.get_occ(feats, rois, 0.2, [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], transform=True)
    for _ in range(args.warmup):
        occ = run()
    from objectcentricocccompletion_amd.occ import fused_mlp as fm
    probe = BlockProbe()
    fm.set_probe(probe)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        occ = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    fm.set_probe(None)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        from objectcentricocccompletion_amd.occ import occ_ops
        cells = int(occ_ops.dense_voxel_centers_batched(rois[:, 4:7], 0.2, [1.0] * 3, [0.5] * 3)[2].sum())
        flops = cells * 2.0 * (60 * 512 + 512 * 1024 + 1024 * 1024 + 1024) + R * 2.0 * 1536 * 512
        ms, fl, detail = probe_summary(probe)
        if ms:   # own kernels ran (bf16): rate of the dominant kernel alone
            roof = {'kernel': ' + '.join(sorted(detail)), 'bound': 'mfma', 'achieved': round(fl / (ms * 1e-3) / 1e12, 1),
                    'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(fl / (ms * 1e-3) / 1e12 / 2500.0, 4), 'traffic': None,
                    'launches_timed': len(probe.items), 'per_kernel': detail,
                    'whole_step_tflops': round(flops * args.steps / dt / 1e12, 1)}
        else:    # f32 run: library GEMMs, whole-step rate
            roof = {'kernel': 'decoder MLP GEMMs (library, f32)', 'bound': 'mfma',
                    'achieved': round(flops * args.steps / dt / 1e12, 1), 'peak': 157.3, 'unit': 'TFLOP/s',
                    'frac': round(flops * args.steps / dt / 1e12 / 157.3, 4), 'traffic': None}
        print(json.dumps({
            'metric': 'object-grids/sec (dense occupancy decode)', 'value': round(world * R * args.steps / dt, 1),
            'unit': 'object-grids/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if args.f32_decoder else 'bf16', 'data': 'synthetic',
            'config': {'workload': f'dense-grid decode (occ_base.py:238-342) of {R} RoIs/GPU, {cells} cells of 0.2 m, '
                                   'ococcnet decoder MLP, inference', 'grids_per_gpu': R, 'cells': cells,
                       'occupied': int(sum(len(t) for s_ in occ for t in s_)), 'parallelism': f'dp{world}'},
            # the dominant kernel, timed by HIP events on its stream: occ_mlp_fwd_kernel = the whole per-cell MLP
            # (2 (64 x 512 + 512 x 1024 + 1024 x 1024) flops per cell, 60 -> 64 input columns padded) in one launch
            'roofline': roof,
            'cpu_baseline': None}), flush=True)


def also_workloads():
    """Short runs (10 timed steps after 8 warm-up steps) of the other workloads, each in a child process of its own, so that the one JSON line the
    driver records also carries configs[2] (at the config's 4 tracklets per GPU, at 16 and at 64: SURVEY 8d), configs[4]'s SST path and the
    dense-grid decode of 64 tracklets' RoIs (8 M cells: the decoder kernel back to back, at sustained clocks).
    Not part of the timed region above; a failure is recorded, it never fails the run."""
    import subprocess
    out = {}
    here = os.path.abspath(__file__)
    for key, extra in (('ococcnet_b4', ['--workload', 'ococcnet', '--tracklets', '4']),
                       ('ococcnet_b16', ['--workload', 'ococcnet', '--tracklets', '16']),
                       ('ococcnet_b64', ['--workload', 'ococcnet', '--tracklets', '64']),
                       ('ococcnet_b64_bf16_operands', ['--workload', 'ococcnet', '--tracklets', '64']),
                       ('sst', ['--workload', 'sst']),
                       ('sst_all_256_objects', ['--workload', 'sst', '--sst-grids', '256']),
                       ('decode_b64', ['--workload', 'decode', '--tracklets', '64'])):
        steps = '30' if key in ('ococcnet_b4', 'ococcnet_b16') else ('5' if key == 'sst_all_256_objects' else '10')   # (the short steps: more of them, the host's load shows)
        cmd = [sys.executable, here, '--steps', steps, '--warmup', '8', '--no-cpu-baseline'] + extra
        env = dict(os.environ)
        if key.endswith('_bf16_operands'):
            # the opt-in product mode of objectcentricocccompletion_amd/gemm.py (library products on bf16-rounded operands, f32
            # accumulation): reported beside the f32 line, never instead of it
            env['OCOCC_GEMM_DTYPE'] = 'bf16'
        t_child = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            d = json.loads(line[-1]) if line else {'error': (r.stderr or 'no output')[-300:]}
            out[key] = {k: d[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config', 'roofline', 'error')
                        if k in d}
            out[key]['wall_s'] = round(time.time() - t_child, 1)
        except Exception as e:   # noqa: BLE001 (timeout, missing device, ...)
            out[key] = {'error': repr(e)[:300]}
    out['conv_sweep'] = conv_sweep()
    return out


def conv_sweep(sizes=(256, 512)):
    """The per-kernel roofline table of configs[1]'s step at 4 x and 8 x the benchmark's 64 grids (the same density, the same
    kernel choices; 512 grids = 1.02 M rows is what the pattern-order records hold): does a convolution kernel's fraction of the
    HBM roofline climb once a compute unit works through several rounds of tiles (the 64-grid step is then bound by its single
    round: the tail), or does it stay (the kernel is)?  Each size a child process: graph replays for the step time, the event
    pairs around the kernels in the eager steps behind them, as in the line of record."""
    import subprocess
    here = os.path.abspath(__file__)
    out = {}
    for grids in sizes:
        cmd = [sys.executable, here, '--grids', str(grids), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-also']
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            d = json.loads(line[-1]) if line else {'error': (r.stderr or 'no output')[-300:]}
            if 'roofline' in d:
                out[str(grids)] = {'ms_per_step': d['ms_per_step'], 'object_grids_per_s': d['value'],
                                   'active_voxels': d['config']['active_voxels'],
                                   'per_kernel': {k: {'us': round(v['avg_launch_ms'] * 1e3, 1), 'frac': v['frac']}
                                                  for k, v in d['roofline']['per_kernel'].items()}}
            else:
                out[str(grids)] = d
        except Exception as e:   # noqa: BLE001
            out[str(grids)] = {'error': repr(e)[:300]}
    return out


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # test hook: OCOCC_BENCH_BACKEND=gloo with OCOCC_BENCH_SHARE_GPU=1 runs several ranks on ONE GPU, to exercise
    # the N>1 launch plan (two graphs + eager all-reduce) on a 1-GPU box; the driver's runs use RCCL, one GPU each
    backend = os.environ.get('OCOCC_BENCH_BACKEND', 'nccl')
    if os.environ.get('OCOCC_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload in ('ococcnet', 'sst', 'decode'):
        {'ococcnet': bench_ococcnet, 'sst': bench_sst, 'decode': bench_decode}[args.workload](args, world, rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    from objectcentricocccompletion_amd.spconv import ops as sp_ops

    torch.manual_seed(0)  # identical initial weights on every rank
    model = SubMOccEncoder(grouped_points=True).to(dev)
    params = [p for p in model.parameters()]
    use_graph = not args.no_graph
    from objectcentricocccompletion_amd.optim import AdamW
    opt = AdamW(params, lr=1e-4)  # fused multi-tensor HIP kernel, device-side step counter
    opt.init_state()
    from objectcentricocccompletion_amd.dist import GradBuckets, broadcast_parameters
    from objectcentricocccompletion_amd.graph import GraphedStep
    broadcast_parameters(model)
    buckets = GradBuckets(params)
    B, P = args.grids, args.points
    xyz, feats, bidx = synthetic_object_grids(B, P, seed=rank, device=dev)

    # Upstream gradient of the encoder output, as a downstream head would hand it back
    # (fixed synthetic bf16 tensor; the voxel count of the fixed synthetic input is fixed).
    with torch.no_grad():
        ref_out = model(xyz, feats, bidx, B)
        n_act = ref_out.features.shape[0]
        n_pairs = int(ref_out.indice_dict['subm1'][3].sum().item())   # (reporting only: the roofline's algorithmic bytes)
        # (reporting only) matrix-instruction over-issue of the output-stationary kernel: it multiplies every 16-row block
        # that has ANY neighbour at an offset (the block masks), useful are the rulebook pairs
        over_issue = None
        rb = getattr(ref_out.indice_dict['subm1'][2], '_ococc', None)
        if rb is not None and rb.tables.get((False, 'fwd'), (None, None))[1] is not None:
            mask = rb.tables[(False, 'fwd')][1].to(torch.int64) & 0xffffffff
            active = int(((mask[:, None] >> torch.arange(27, device=mask.device)[None, :]) & 1).sum().item())
            over_issue = round(active * 16 / max(n_pairs, 1), 2)
        del ref_out

    def sorted_kernel_in_use():
        """does a rulebook built now carry the neighbour-pattern row order?  (and the order's over-issue: 16-row blocks of
        SLOTS with any neighbour at an offset)"""
        with torch.no_grad():
            o = model(xyz, feats, bidx, B)
            rb2 = getattr(o.indice_dict['subm1'][2], '_ococc', None)
            if rb2 is None or not sp_ops._sorted_regime(rb2):
                return False, None
            table, _, rows = rb2.tables[(False, 'fwd')]
            rec, _ = sp_ops.row_order(rb2, table, rows)
            m = rec[:, 1].to(torch.int64) & 0xffffffff
            m = torch.cat([m, m.new_zeros((-m.numel()) % 16)]).view(-1, 16)
            blk = m[:, 0]
            for j in range(1, 16):
                blk = blk | m[:, j]
            active = int(((blk[:, None] >> torch.arange(27, device=blk.device)[None, :]) & 1).sum().item())
            return True, round(active * 16 / max(n_pairs, 1), 2)
    # The sub-manifold convolutions pick their kernel from the rulebook density (compact-then-multiply below ~2-3 pairs
    # per row).  Nothing is set here: spconv.ops.density measured it on the device while that first forward built its
    # rulebook (an asynchronous copy behind an event); harvest it now, before the warm-up steps and the graph capture.
    torch.cuda.synchronize()
    sp_ops.density.poll()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    d_act = (torch.randn(n_act, 128, generator=gen, device=dev) / n_act).to(torch.bfloat16)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        out = model(xyz, feats, bidx, B)
        out.features.backward(d_act)
        buckets.all_reduce()  # data parallel: bucketed gradient all-reduce over RCCL / xGMI (no-op at N=1)
        opt.step()
        return out

    # Fixed-capacity form for graph capture: one voxel row per point, rows past the active count are
    # inert (-1 coordinates, no rulebook pair) and receive a zero upstream gradient.
    d_cap = torch.zeros(B * P, 128, dtype=torch.bfloat16, device=dev)
    d_cap[:n_act] = d_act

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        out = model(xyz, feats, bidx, B, static=True)
        # (sp_ops.overlap_wgrad() would put the weight gradients on a parallel graph branch; measured slower,
        # 0.505-0.527 vs 0.489 ms/step: the overlapped kernels contend for the same gather path)
        out.features.backward(d_cap)
        return out

    def whole_step():
        out = fwd_bwd()
        opt.step()
        return out

    # every convolution kernel of the step, HIP events on the launch stream.  Event-record
    # nodes inside a captured graph are rejected by this ROCm (hipEventRecordExternal: invalid argument),
    # so in graph mode the events go around the same kernels in eager steps run right after the timed
    # replays (same process, same inputs); the rocprofv3 trace of the replays is the cross-check.
    # Graph mode: each event pair brackets 8 back-to-back launches of the kernel (they only write their outputs), so the
    # event-record overhead (~10 us around one 40 us launch) does not end up in the average.
    probe = sp_ops.KernelProbe(repeat=8 if use_graph else 1)

    graph_note = 'eager launches'
    pipelined = False
    if use_graph:
        try:
            if world == 1 and not args.split_graph and args.pipeline:
                # the geometry of the NEXT batch (voxelise, scatter-mean, rulebook: points only, no weights) is
                # recorded on a forked stream of the same graph and runs beside the convolutions (graph.PipelinedStep)
                from objectcentricocccompletion_amd.graph import PipelinedStep

                def train_on(geom):
                    opt.zero_grad(set_to_none=True)
                    out = model(geometry=geom)
                    out.features.backward(d_cap)
                    opt.step()
                    return out

                if os.environ.get('OCOCC_PIPE_FORK', 'head') == 'tail':
                    # fork BEHIND the backward pass: the next batch's geometry runs beside the end of the step --
                    # parameter-gradient sums (held back from the end-of-backward callback), AdamW, next weight operands
                    from objectcentricocccompletion_amd import _deferred
                    model.prepare_weights(grad=True)

                    def fwd_bwd_held(geom):
                        opt.zero_grad(set_to_none=True)
                        out = model(geometry=geom, weights_ready=True)
                        with _deferred.hold():
                            out.features.backward(d_cap)
                        return out

                    def end_of_step():
                        _deferred.flush()
                        opt.step()
                        model.prepare_weights(grad=True)
                    g_pipe = PipelinedStep(lambda: model.geometry(xyz, feats, bidx, B, static=True), fwd_bwd_held,
                                           tail=end_of_step)
                elif os.environ.get('OCOCC_PIPE_FORK', 'head') == 'backward':
                    def fwd_only(geom):
                        opt.zero_grad(set_to_none=True)
                        return model(geometry=geom)

                    def bwd_opt(out):
                        out.features.backward(d_cap)
                        opt.step()
                        return out
                    g_pipe = PipelinedStep(lambda: model.geometry(xyz, feats, bidx, B, static=True), bwd_opt, forward=fwd_only)
                else:
                    g_pipe = PipelinedStep(lambda: model.geometry(xyz, feats, bidx, B, static=True), train_on)
                step = g_pipe.replay
                pipelined = True
            elif world == 1 and not args.split_graph:
                g_all = GraphedStep(whole_step, warmup=3)

                def step():
                    return g_all.replay()
            else:
                # the gradient all-reduce stays an eager RCCL call between two graphs
                # graph 1: forward + backward + gradients packed into the flat buckets; eager: the bucket
                # all-reduce (the only RCCL call); graph 2: unpack (average, copy back) + AdamW
                def fwd_bwd_pack():
                    out = fwd_bwd()
                    buckets.pack()
                    return out

                def unpack_step():
                    buckets.unpack()
                    opt.step()

                g_fb = GraphedStep(fwd_bwd_pack, warmup=3)
                g_opt = GraphedStep(unpack_step, warmup=0, pool=g_fb.pool())

                def step():
                    out = g_fb.replay()
                    buckets.reduce()
                    g_opt.replay()
                    return out
            if pipelined:
                graph_note = ('one hipGraphLaunch per step; the next batch\'s geometry (voxelise, scatter-mean, rulebook) '
                              'runs on a forked stream of the same graph')
            elif world == 1 and not args.split_graph:
                graph_note = 'one hipGraphLaunch per step'
            else:
                graph_note = 'two HIP graphs + eager RCCL all-reduce'
        except Exception as e:  # noqa: BLE001 -- report and fall back to eager launches, never to another device
            print(f'[bench] HIP graph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            use_graph = False
            probe.repeat = 1
    if not use_graph:
        step = eager_step

    for _ in range(args.warmup):
        out = step()
    if not use_graph:
        sp_ops.set_probe(probe)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    sp_ops.set_probe(None)
    dump_params(args, rank, params)
    # host cost of queueing one step on an idle device (one hipGraphLaunch, or the eager launch sequence): when it
    # approaches ms_per_step the number above is the host's, not the GPU's
    host_us = []
    for _ in range(5):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host_us.append((time.perf_counter() - h0) * 1e6)
    torch.cuda.synchronize()
    host_us = sorted(host_us)[len(host_us) // 2]
    if use_graph:
        sp_ops.set_probe(probe)
        for _ in range(min(args.steps, 20)):
            eager_step()
        torch.cuda.synchronize()
        sp_ops.set_probe(None)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        n_vox = int(n_act)
        sorted_used, sorted_over_issue = sorted_kernel_in_use()
        # the longest convolution kernel of the step, and every other one beside it
        per_kernel = {}
        for key in probe.keys():
            ms = probe.mean_ms(key)
            if ms:
                nbytes, what = conv_algorithmic_bytes(key, n_vox, n_pairs)
                per_kernel[key] = {'avg_launch_ms': round(ms, 5), 'launches_timed': probe.count(key),
                                   'algorithmic_bytes_per_launch': nbytes,
                                   'frac': round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        top = max(per_kernel, key=lambda k: per_kernel[k]['avg_launch_ms']) if per_kernel else None
        kern_ms = per_kernel[top]['avg_launch_ms'] if top else None
        alg_bytes, what = conv_algorithmic_bytes(top, n_vox, n_pairs) if top else (None, '')
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms else None
        family = top.split('<')[0] if top else ''
        kd_top, nc_top = (int(v) for v in top.split('<')[1].rstrip('>').split(',')) if top else (0, 0)
        res = {
            'metric': 'object-grids/sec (fwd+bwd)',
            'value': round(world * B * args.steps / dt, 1),
            'unit': 'object-grids/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16',
            'data': 'synthetic',
            'config': {
                'workload': 'configs[1]: SubMConv3d-only occupancy encoder fwd+bwd+AdamW, '
                            f'{B} object grids/GPU x {P} random points, 0.2 m voxels, 40^3 grid, '
                            'channels 16-32-64-128, LN+GELU, bf16 features',
                'grids_per_gpu': B, 'points_per_grid': P, 'active_voxels': n_vox,
                'rulebook_pairs': n_pairs, 'parallelism': f'dp{world}', 'launch': graph_note,
                'host_us_to_queue_one_step': round(host_us, 1),
            },
            'roofline': {
                'kernel': ('%s<%d,%d> (SubMConv3d %d->%d %s: %s)' % (
                    KERNEL_NAMES.get(family, family), kd_top, nc_top, *((kd_top, nc_top, 'forward') if kd_top < nc_top
                                                                       else (nc_top, kd_top, 'input gradient')), what)) if top else None,
                'key': top,
                'chosen_as': 'the convolution kernel with the largest average launch time in this run',
                'bound': 'hbm',
                'achieved': round(achieved, 1) if achieved else None,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                'traffic': pmc_traffic(top),
                # (`traffic` is NOT measured in this run: it is the committed counter pass of the same kernel on the same workload)
                'traffic_source': ('profiles/' + PMC_JSON[top]) if top in PMC_JSON and pmc_traffic(top) is not None else None,
                'algorithmic_bytes_per_launch': alg_bytes,
                'mfma_over_issue_pattern_order': sorted_over_issue if sorted_used else None,   # 16-row blocks multiplied x 16 / rulebook pairs
                'mfma_over_issue_voxel_order': over_issue,
                'avg_launch_ms': round(kern_ms, 5) if kern_ms else None,
                'launches_timed': probe.count(top) if top else 0,
                'timed_in': 'timed region' if not use_graph else
                            'eager steps after the timed graph replays, 8 back-to-back launches per event pair',
                'per_kernel': per_kernel,
            },
        }
        wall = {'to_result_s': round(time.time() - T_START, 1)}
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: the other ranks would sit in the barrier)
            t0 = time.time()
            res['cpu_baseline'] = cpu_baseline(16, P, model)
            wall['cpu_baseline_s'] = round(time.time() - t0, 1)
        if world == 1 and not args.no_also:
            t0 = time.time()
            res['also'] = also_workloads()
            wall['also_s'] = round(time.time() - t0, 1)
        res['wall'] = wall
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
