#!/bin/bash
# usage: tools/prof_any.sh <tag> <steps-in-run> [bench args...] -- rocprofv3 kernel stats of any bench workload
tag=$1; n=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
n=$n
tot=0
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per
for r in rows[:45]:
    per=int(r['TotalDurationNs'])/n/1e3
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/n:7.1f} {float(r['AverageNs'])/1e3:8.1f} {per:9.1f}")
print('kernel us/step',round(tot,1), 'launches/step', sum(int(r['Calls']) for r in rows)/n)
PY
tail -c 400 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json
