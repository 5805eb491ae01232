#!/usr/bin/env python3
"""Static check of the gather_gemm_stream_kernel ISA (run after editing the kernel).

The per-offset loop issues its loads from inline asm, so the compiler does not know that the
destination registers are still in flight.  Any compiler-generated instruction (copy, cast, spill)
that touches such a register between the load and the explicit s_waitcnt reads or clobbers garbage.
This script compiles sparse_conv.hip to assembly and walks every instantiation with a model of the
in-order vmcnt queue (prologue, then the loop twice so that the state at the back edge meets the loop
header), reporting vector instructions that touch a register with a load in flight, and scratch
(spill) traffic inside the loop.  MFMAs are left out: their out-of-line blocks follow the loop in the
text, not in execution order, and in the source every MFMA sits behind stream_wait_vm + stream_tie.

usage: tools/check_stream_isa.py   (exit code 1 when something is found)
"""
import os, re, subprocess, sys, tempfile

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, '..', 'objectcentricocccompletion_amd', 'csrc', 'sparse_conv.hip')
out = os.path.join(tempfile.gettempdir(), 'ococc_sparse_conv.s')
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '-x', 'hip', src, '-o', out,
                '--cuda-device-only'], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
fail = 0
for m in re.finditer(r'\n(_ZN12_GLOBAL__N_125gather_gemm_stream_kernel(\w+)):', s):
    a = m.end(); b = s.index('s_endpgm', a); body = s[a:b].split('\n')
    bars = [i for i, l in enumerate(body) if 's_barrier' in l]
    head = next(i for i, l in enumerate(body) if 'This Inner Loop Header' in l)
    # the latch block (wait + barrier) sits right in front of the loop header in the text ...
    latch = max(i for i in bars if i < head) - 1
    # ... and the loop (with its out-of-line MFMA blocks) ends at the explicit vmcnt(0) behind it
    end = next(i for i, l in enumerate(body) if i > head and re.search(r's_waitcnt vmcnt\(0\)\s*$', l))
    seq = body[:end] + body[latch:end]
    queue, bad, scratch = [], [], 0
    for idx, l in enumerate(seq):
        t = l.strip()
        if not t or t[0] in ';.': continue
        op = t.split()[0]
        regs = set()
        for mm in re.finditer(r'v\[(\d+):(\d+)\]', t): regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for mm in re.finditer(r'\bv(\d+)\b', t): regs.add(int(mm.group(1)))
        inflight = set().union(*queue) if queue else set()
        if op == 's_waitcnt':
            mm = re.search(r'vmcnt\((\d+)\)', t)
            if mm:
                n = int(mm.group(1))
                queue = queue[len(queue) - n:] if 0 < n < len(queue) else ([] if n == 0 else queue)
            continue
        if regs & inflight and not op.startswith('v_mfma'): bad.append(t[:90])
        if 'scratch_' in op and idx >= latch: scratch += 1
        if op.startswith(('buffer_load', 'global_load', 'scratch_load')):
            dst = set()
            if 'lds' not in op:
                mm = re.match(r'\S+\s+v\[(\d+):(\d+)\]', t) or re.match(r'\S+\s+v(\d+)()', t)
                if mm: dst = set(range(int(mm.group(1)), int(mm.group(2) or mm.group(1)) + 1))
            queue.append(dst)
        elif op.startswith(('global_store', 'buffer_store', 'scratch_store')):
            queue.append(set())
    tag = m.group(2)[:22]
    print(f'{tag:24s} touches of in-flight registers: {len(bad):3d}   scratch ops in the loop: {scratch}')
    for t in bad[:4]: print('     ', t)
    fail |= bool(bad) or bool(scratch)
sys.exit(1 if fail else 0)
