#!/usr/bin/env python3
"""Static check of the gather_gemm_stream_kernel ISA (run after editing the kernel).

The per-offset loop issues its loads from inline asm, so the compiler does not know that the
destination registers are still in flight.  Any compiler-generated instruction (copy, cast, spill)
that touches such a register between the load and the explicit s_waitcnt reads or clobbers garbage,
and a spill inside the loop would add scratch traffic to the hand-counted vmcnt queue.

This script compiles sparse_conv.hip to assembly and walks every instantiation with a model of the
in-order vmcnt queue: the prologue, then the main text of the loop twice (so that the state at the
back edge meets the loop header).  Blocks the compiler parks behind the loop (rarely taken MFMA
groups, entered by a branch from the main text) are checked with the queue state at that branch and
must not contain memory operations.  Reported: vector instructions that touch a register with a
load in flight (MFMAs excepted: in the source every MFMA sits behind stream_wait_vm + stream_tie),
and scratch traffic inside the loop.

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
    # The second pass starts at the loop header -- or, when the block in front of the header is a latch
    # (a label there is branched to from inside the loop; its wait + barrier belong to the loop), at that block.
    latch = head
    prev_labels = [i for i in range(head) if re.match(r'\.LBB\d+_\d+:', body[i])]
    if prev_labels:
        li = prev_labels[-1]
        lab = body[li].split(':')[0]
        if any(re.search(r's_cbranch\S*\s+' + re.escape(lab) + r'\b|s_branch\s+' + re.escape(lab) + r'\b', l)
               for l in body[head:]):
            latch = li
    # ... and the loop (with its out-of-line MFMA blocks) ends at the explicit vmcnt(0) behind it
    end = next(i for i, l in enumerate(body) if i > head and re.search(r's_waitcnt vmcnt\(0\)\s*$', l))
    # Text order is execution order up to the back edge; behind it the compiler parks out-of-line blocks
    # (rarely taken MFMA groups) that are entered by a branch from the main text and branch back.  They are
    # checked with the queue state at the branch that enters them and must not contain memory operations.
    # (the main text of the loop ends at the first unconditional branch behind the header: the back edge itself
    # or the jump over the parked blocks to the code behind the loop)
    back = next(i for i in range(head, end) if body[i].strip().startswith('s_branch'))
    labels = {body[i].split(':')[0]: i for i in range(len(body)) if re.match(r'\.LBB\d+_\d+:', body[i])}

    def regs_of(t):
        r = set()
        for mm in re.finditer(r'v\[(\d+):(\d+)\]', t): r.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for mm in re.finditer(r'\bv(\d+)\b', t): r.add(int(mm.group(1)))
        return r

    queue, bad, scratch = [], [], 0

    def check_out_of_line(start, inflight):
        # (a block without the 'in Loop' note is the loop's EXIT: there the compiler may reload what it spilled
        # around the loop -- into registers that are not the target of a load still in flight, checked below; the
        # explicit vmcnt(0) follows before anything is used)
        leaving = 'in Loop' not in body[start]
        for l in body[start + 1:end]:
            t = l.strip()
            if not t or t[0] == ';': continue
            if t[0] == '.': continue
            op = t.split()[0]
            reload = leaving and op.startswith('scratch_load')
            if (op.startswith(('buffer_', 'global_', 'scratch_')) and not reload) or (op == 's_waitcnt' and 'vmcnt' in t):
                bad.append('memory operation in an out-of-line block: ' + t[:60])
            if regs_of(t) & inflight and not op.startswith('v_mfma'): bad.append(t[:90])
            if op == 's_branch': return

    for lo, hi in ((0, back + 1), (latch, back + 1)):
        for idx in range(lo, hi):
            t = body[idx].strip()
            if not t or t[0] in ';.': continue
            op = t.split()[0]
            inflight = set().union(*queue) if queue else set()
            if op == 's_waitcnt':
                mm = re.search(r'vmcnt\((\d+)\)', t)
                if mm:
                    n = int(mm.group(1))
                    queue = queue[len(queue) - n:] if 0 < n < len(queue) else ([] if n == 0 else queue)
                continue
            if op.startswith('s_cbranch') or op == 's_branch':
                tgt = t.split()[-1]
                if tgt in labels and labels[tgt] > back: check_out_of_line(labels[tgt], inflight)
                continue
            if regs_of(t) & inflight and not op.startswith('v_mfma'): bad.append(t[:90])
            if 'scratch_' in op and idx >= head: scratch += 1
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
