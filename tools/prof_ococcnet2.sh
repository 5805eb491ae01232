#!/bin/bash
# usage: tools/prof_ococcnet2.sh <tag> <tracklets> [min us per step to list]   (on the GPU box): kernel time per step of --workload ococcnet,
# grouped: own HIP / hipBLASLt / ATen + copies; keeps gpurun_out/<tag>_kernel_stats.csv
tag=$1; trk=$2; minus=${3:-150}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload ococcnet --tracklets $trk --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2>/dev/null
stats=$(find $out -name "${tag}_kernel_stats.csv" | head -1)
cp $stats $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv')))
n=13
tot=0; calls=0; grp={'own':[0,0],'hipblaslt':[0,0],'aten+copies':[0,0]}
for r in rows:
    per=int(r['TotalDurationNs'])/n/1e3; tot+=per; c=int(r['Calls'])/n; calls+=c
    name=r['Name']
    g='hipblaslt' if name.startswith('Cijk') else ('own' if '(anonymous namespace)::' in name and 'at::' not in name else 'aten+copies')
    grp[g][0]+=per; grp[g][1]+=c
    if per>$minus: print(f"{name[:100]:100s} {c:7.1f} {float(r['AverageNs'])/1e3:7.1f} {per:8.1f}")
print('kernel us/step',round(tot,1),'launches/step',round(calls,1))
for g,(t,c) in grp.items(): print(f'  {g:12s} {t:9.1f} us  {100*t/tot:5.1f} %  {c:7.1f} launches')
PY
python3 -c "import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print('ms_per_step under the profiler',d['ms_per_step'])"
rm -rf $out
