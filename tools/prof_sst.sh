#!/bin/bash
# usage (GPU box): tools/prof_sst.sh <tag>   kernel time per call of --workload sst
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload sst --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/${tag}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:8]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
print('total kernel ms', tot/1e6)
PY
