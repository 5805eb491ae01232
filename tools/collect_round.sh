O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O/r02
python bench.py > $O/r02/bench_graph.json 2> $O/r02/bench_graph.err; tail -c 900 $O/r02/bench_graph.json
python bench.py --no-graph --no-cpu-baseline --steps 50 > $O/r02/bench_eager.json 2>/dev/null; cut -c1-200 $O/r02/bench_eager.json
PROF_MIN_US=0.05 bash tools/prof_stats.sh r02d --steps 50 --warmup 10 > $O/r02/table.txt 2>&1; tail -3 $O/r02/table.txt | cut -c1-200
# (PMC traffic of the dominant kernel: tools/pmc_hbm.sh r02d gather_gemm_stream -- kernel unchanged since profiles/r02_pmc_gather_gemm_stream_64_128.json)
timeout 300 python bench.py --workload ococcnet > $O/r02/ococc_b4.json 2>/dev/null; cut -c1-250 $O/r02/ococc_b4.json
timeout 300 python bench.py --workload ococcnet --tracklets 64 --no-cpu-baseline --steps 30 > $O/r02/ococc_b64.json 2>/dev/null; cut -c1-250 $O/r02/ococc_b64.json
timeout 300 python bench.py --workload ococcnet --tracklets 64 --no-cpu-baseline --f32-decoder --steps 20 > $O/r02/ococc_b64_f32.json 2>/dev/null; cut -c1-250 $O/r02/ococc_b64_f32.json
bash tools/prof_any.sh r02oc64 7 --workload ococcnet --tracklets 64 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-200
bash tools/prof_any.sh r02sst 13 --workload sst --steps 10 --warmup 3 2>&1 | tail -3 | cut -c1-300
rm -f $O/prof_*/*kernel_trace.csv
