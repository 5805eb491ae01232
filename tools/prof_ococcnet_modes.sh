#!/bin/bash
# usage (GPU box): tools/prof_ococcnet_modes.sh <tracklets> <mode>...   kernel time per step of --workload ococcnet per decoder backward mode
b=$1; shift
cd /tmp && export TMPDIR=/tmp
for mode in "$@"; do
  tag=r05_${mode}_b$b
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
  mkdir -p $out
  OCOCC_DECODER_BACKWARD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload ococcnet --tracklets $b --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2>/dev/null
  python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$out/${tag}_kernel_stats.csv')))
n=13
tot=0; calls=0
print('==== $mode B=$b')
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per; calls+=int(r['Calls'])/n
    name=r['Name']
    if name.startswith('Cijk'): name='Cijk '+re.search(r'MT\d+x\d+x\d+',name).group(0)+(' BBS' if '_BBS_' in name else ' S')+name[4:24]
    if per>400: print(f"{name[:90]:90s} {int(r['Calls'])/n:7.1f} {float(r['AverageNs'])/1e3:8.1f} {per:9.1f}")
print('kernel us/step',round(tot,1),'launches/step',round(calls,1))
PY
done
