#!/bin/bash
# usage: tools/prof_gaps.sh <tag> [bench args...]   (GPU box): kernel trace of a short run -> tools/trace_gaps.py
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/gaps_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-also --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/gaps_$tag/bench.json 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $(find /tmp/gaps_$tag -name "*kernel_trace.csv" | head -1) | tee $GRAFT_REPO_ROOT/gpurun_out/gaps_$tag/gaps.txt
