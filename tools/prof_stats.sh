#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]   (run on the GPU box through gpurun)
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
n=max(int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name'] or 'multi_tensor_apply' in r['Name'])
tot=0
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per
    if per>6: print(f"{r['Name'][:86]:86s} {int(r['Calls'])/n:5.1f} {float(r['AverageNs'])/1e3:7.1f} {per:7.1f}")
print('steps',n,'kernel us/step',round(tot,1))
PY
tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json
