#!/bin/bash
# usage (GPU box): tools/prof_sir_fused.sh <tracklets>   SIR kernel time per step of --workload ococcnet, one launch per layer against one per block
b=$1
cd /tmp && export TMPDIR=/tmp
for fused in 1 0; do
  tag=r05_sir_fused${fused}_b$b
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
  mkdir -p $out
  OCOCC_SIR_FUSED=$fused rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload ococcnet --tracklets $b --steps 10 --warmup 3 --no-cpu-baseline --no-also > $out/bench.json 2>/dev/null
  python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$out/${tag}_kernel_stats.csv')))
n=13
tot=0; calls=0; sir=0; sirc=0
print('==== fused=$fused B=$b')
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per; calls+=int(r['Calls'])/n
    name=r['Name']
    if re.search(r'point_mlp|sir_fused|segment_argmax|join_cols|place_cols|fill_kernel|shortcut', name):
        sir+=per; sirc+=int(r['Calls'])/n
        print(f"{name[:100]:100s} {int(r['Calls'])/n:7.1f} {float(r['AverageNs'])/1e3:8.1f} {per:9.1f}")
print('SIR kernels us/step',round(sir,1),'launches',round(sirc,1),'| all kernels us/step',round(tot,1),'launches/step',round(calls,1))
PY
done
