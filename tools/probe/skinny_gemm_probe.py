#!/usr/bin/env python3
"""ARCHIVED EXPERIMENT (numbers in the header of tools/probe/skinny_gemm.hip.txt).  The few-row f32 products of tools/probe/skinny_gemm.hip.txt (three-way split operands on the bf16 matrix cores, the weight streamed
once) against the library's f32 GEMMs: error vs float64 and device time per launch (HIP events around 50 back-to-back
launches) for the token-stack / head shapes of configs[2] at 128 and 256 rows.  Builds its own copy of the translation
unit.  usage: python tools/probe/skinny_gemm_probe.py [workgroup target ...]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libskinny_probe.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-x', 'hip',
                os.path.join(ROOT, 'tools', 'probe', 'skinny_gemm.hip.txt'), os.path.join(csrc, 'capi.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.ococc_skinny_workspace_bytes.restype = i64
lib.ococc_skinny_workspace_bytes.argtypes = [i64, i64]
lib.ococc_skinny_linear_f32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, vp, i64, vp, i64, vp]
lib.ococc_skinny_dgrad_f32.argtypes = [vp, i64, vp, i64, i64, i64, i64, vp, i64, vp, i64, vp]
lib.ococc_skinny_wgrad_f32.argtypes = [vp, i64, vp, i64, i64, i64, i64, vp, i64, vp]
lib.ococc_last_error.restype = ctypes.c_char_p
dev = torch.device('cuda:0')
stream = lambda: torch.cuda.current_stream().cuda_stream


def check(rc):
    assert rc == 0, lib.ococc_last_error()


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


targets = [int(v) for v in sys.argv[1:]] or [288]
shapes = [(3072, 1536), (1536, 1536), (512, 1536), (1536, 512), (2048, 3072), (2048, 2048), (1536, 2048), (512, 512)]
rel = lambda a, e: float((a.double() - e).norm() / e.norm())
for target in targets:
    check(lib.ococc_skinny_set_target(target))
    print(f'--- workgroup target {target}')
    for M in (128, 256, 100):
        for N, K in shapes:
            g = torch.Generator(device=dev).manual_seed(N + K + M)
            x = torch.randn(M, K, device=dev, generator=g)
            w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
            b = torch.randn(N, device=dev, generator=g)
            dy = torch.randn(M, N, device=dev, generator=g)
            ws = torch.zeros(int(lib.ococc_skinny_workspace_bytes(N, K)), dtype=torch.uint8, device=dev)
            y, dx, dw = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
            f = lambda: check(lib.ococc_skinny_linear_f32(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), M, N, K, y.data_ptr(), N,
                                                          ws.data_ptr(), ws.numel(), stream()))
            d = lambda: check(lib.ococc_skinny_dgrad_f32(dy.data_ptr(), N, w.data_ptr(), K, M, N, K, dx.data_ptr(), K, ws.data_ptr(),
                                                         ws.numel(), stream()))
            wg = lambda: check(lib.ococc_skinny_wgrad_f32(dy.data_ptr(), N, x.data_ptr(), K, M, N, K, dw.data_ptr(), K, stream()))
            f(); d(); wg()
            torch.cuda.synchronize()
            xd, wd, dyd = x.double(), w.double(), dy.double()
            e = (rel(y, xd @ wd.t() + b.double()), rel(dx, dyd @ wd), rel(dw, dyd.t() @ xd))
            t_own = (bench(f), bench(d), bench(wg))
            t_lib = (bench(lambda: torch.addmm(b, x, w.t())), bench(lambda: torch.mm(dy, w)), bench(lambda: torch.mm(dy.t(), x)))
            if M == 100:   # (a ragged row count: correctness only)
                print(f'M {M:4d} N {N:5d} K {K:5d}: err {e[0]:.1e} {e[1]:.1e} {e[2]:.1e}')
                continue
            print(f'M {M:4d} N {N:5d} K {K:5d}: err {e[0]:.1e} {e[1]:.1e} {e[2]:.1e}   own us fwd {t_own[0]:6.1f} dgrad {t_own[1]:6.1f} '
                  f'wgrad {t_own[2]:6.1f}   library {t_lib[0]:6.1f} {t_lib[1]:6.1f} {t_lib[2]:6.1f}', flush=True)
