#!/bin/bash
# usage: tools/prof_replay.sh <tag> [bench args...]   (on the GPU box through gpurun): rocprofv3 kernel trace of bench.py cut to the
# graph replays (tools/replay_table.py) -> gpurun_out/<tag>_replay_table.txt and <tag>_kernel_stats.csv; the raw trace is dropped
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-also --no-cpu-baseline "$@" > $out/bench.json 2>/dev/null
trace=$(find $out -name "${tag}_kernel_trace.csv" | head -1)
stats=$(find $out -name "${tag}_kernel_stats.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/replay_table.py $trace ${PROF_MIN_US:-0} > $GRAFT_REPO_ROOT/gpurun_out/${tag}_replay_table.txt
cp $stats $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
cp $out/bench.json $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_rocprof.json
rm -rf $out
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/${tag}_replay_table.txt
