#!/bin/bash
# usage: tools/prof_ococcnet.sh <tag>   (on the GPU box): kernel time per step of --workload ococcnet
tag=$1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload ococcnet --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
n=13
tot=0; calls=0
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per; calls+=int(r['Calls'])/n
    if per>150: print(f"{r['Name'][:100]:100s} {int(r['Calls'])/n:7.1f} {float(r['AverageNs'])/1e3:7.1f} {per:8.1f}")
print('kernel us/step',round(tot,1),'launches/step',round(calls,1))
PY
head -c 200 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/bench.json
