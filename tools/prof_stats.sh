#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]   (run on the GPU box through gpurun)
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-also --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
n=max(int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name'] or 'multi_tensor_apply' in r['Name'])
tot=0
note=False
for r in rows:
    calls=int(r['Calls'])/n
    per=int(r['TotalDurationNs'])/n/1e3
    tag=' '
    if 'gather_gemm_stream_kernel<64, 128' in r['Name'] and abs(calls-round(calls))>1e-6:
        # bench.py's roofline probe: 20 eager steps after the timed region launch this kernel 8x back to back
        per=float(r['AverageNs'])/1e3; calls=1.0; tag='*'; note=True
    tot+=per
    if per>float(__import__("os").environ.get("PROF_MIN_US","6")): print(f"{r['Name'][:86]:86s} {calls:5.1f}{tag}{float(r['AverageNs'])/1e3:7.1f} {per:7.1f}")
print('steps',n,'kernel us/step',round(tot,1))
if note: print('* once per step; the other launches in the trace are the roofline probe (8 back-to-back launches per event pair, after the timed region)')
PY
tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json
# replay-only accounting: the same trace cut to the graph replays (tools/replay_table.py)
python3 $GRAFT_REPO_ROOT/tools/replay_table.py $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_trace.csv ${PROF_MIN_US:-6} | tee $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_replay_table.txt
# (the raw trace is tens of MB; the table and the stats csv are what is kept)
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_trace.csv
