cd /tmp && export TMPDIR=/tmp
for r in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_rel$r; mkdir -p $out
  OCOCC_SIR_BATCH_REL=$r rocprofv3 --kernel-trace --stats --output-format csv -d $out -o rel$r -- python3 $GRAFT_REPO_ROOT/bench.py --workload ococcnet --tracklets 64 --steps 10 --warmup 3 --no-cpu-baseline --no-also > $out/bench.json 2>/dev/null
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/rel${r}_kernel_stats.csv')))
n=13
tot=sum(int(x['TotalDurationNs']) for x in rows)
print('batch_rel=$r kernel ms/step', round(tot/n/1e6,2), 'launches', round(sum(int(x['Calls']) for x in rows)/n))
for x in rows:
    if 'rel_chains' in x['Name'] or 'wgrad_multi' in x['Name']: print('   ', x['Name'][:70], round(int(x['Calls'])/n,1), round(float(x['AverageNs'])/1e3,1))
PY
  rm -f $out/*kernel_trace.csv
done
