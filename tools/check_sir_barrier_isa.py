#!/usr/bin/env python3
"""Static check of the grid barrier of the one-launch SIR layer (csrc/sir_fused_impl.hpp: bar_arrive).

A workgroup may announce itself at a grid barrier only when every one of its waves has drained its outstanding
vector-memory operations: the no-return atomics on the segment maxima, the arg-max rows and the collected gradients
are what the other workgroups read behind the barrier.  `__syncthreads()` alone compiles to a bare `s_barrier` on
gfx950, which does not wait for them (round-5 advisor finding); bar_arrive therefore issues `s_waitcnt vmcnt(0)`
in front of it.  This script compiles the three tile-size translation units to gfx950 assembly and checks, for every
arrival (a RETURNING `global_atomic_add ... sc0` behind an `s_barrier`), that walking back from that `s_barrier` an
`s_waitcnt vmcnt(0)` is met before any vector-memory instruction or branch target.

usage: tools/check_sir_barrier_isa.py   (exit code 1 when an arrival is not covered)
"""
import os, re, subprocess, sys, tempfile

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', 'objectcentricocccompletion_amd', 'csrc')
MEM = ('global_', 'buffer_', 'flat_', 'scratch_')
fail = arrivals = 0
jobs = {}
for mb in (1, 2, 4):
    src = os.path.join(csrc, f'sir_fused_mb{mb}.hip')
    out = os.path.join(tempfile.gettempdir(), f'ococc_sir_fused_mb{mb}_{os.getpid()}.s')
    jobs[mb] = (out, subprocess.Popen(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '-x', 'hip', src, '-o', out,
                                       '--cuda-device-only'], stderr=subprocess.DEVNULL))
for mb, (out, job) in jobs.items():
    if job.wait() != 0:
        sys.exit(f'hipcc failed on sir_fused_mb{mb}.hip')
    lines = open(out).read().split('\n')
    os.unlink(out)
    kernel = '?'
    for i, l in enumerate(lines):
        m = re.match(r'(_ZN\S*sir_fused_(fwd|bwd)_kernel\S*):', l)
        if m:
            kernel = m.group(1)
        t = l.strip()
        if not (t.startswith('global_atomic_add ') and t.endswith('sc0')):
            continue
        # the workgroup barrier this atomic stands behind (none: the census counter at kernel start)
        bar = None
        for j in range(i - 1, max(i - 80, 0), -1):
            u = lines[j].strip()
            if u.startswith('s_barrier'):
                bar = j
                break
            if u.startswith('s_endpgm') or re.match(r'_ZN', u):
                break
        if bar is None:
            continue
        arrivals += 1
        ok = False
        for j in range(bar - 1, max(bar - 40, 0), -1):
            u = lines[j].strip()
            if not u or u[0] == ';':
                continue
            if re.match(r's_waitcnt\b.*vmcnt\(0\)', u):
                ok = True
                break
            if u.startswith(MEM) or re.match(r'\.LBB\d+_\d+:', u):
                break
        if not ok:
            fail += 1
            print(f'mb{mb} {kernel}: arrival at line {i + 1} -- no s_waitcnt vmcnt(0) directly in front of the s_barrier at line {bar + 1}')
print(f'grid-barrier arrivals checked: {arrivals}')
print(f'arrivals without vmcnt(0) in front of the barrier: {fail}')
sys.exit(1 if fail or arrivals == 0 else 0)
