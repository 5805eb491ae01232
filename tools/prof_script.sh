#!/bin/bash
# usage: tools/prof_script.sh <tag> <script.py> [args...]  -- rocprofv3 kernel stats of any python script (on the GPU box)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 "$@" > $out/stdout.txt 2>$out/stderr.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/${tag}_kernel_stats.csv')))
for r in rows[:30]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
