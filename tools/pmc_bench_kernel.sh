#!/bin/bash
# usage (GPU box): tools/pmc_bench_kernel.sh <tag> <kernel-substring> [bench.py args]
# HBM-side bytes per launch of ONE kernel of the configs[1] step (bench.py's graph replays and probe steps), collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate bounded passes, FETCH_SIZE x 2 on gfx950.
tag=$1; kern=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o h$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5 "$@" > /dev/null 2>$out/herr$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections, json
acc=collections.defaultdict(list)
for f in sorted(glob.glob('$out/h*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if '''$kern''' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m={k: sum(v)/len(v) for k,v in acc.items()}
print({k:(v,len(acc[k])) for k,v in m.items()})
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    res={'kernel': '''$kern''', 'FETCH_SIZE_KB_raw': m['FETCH_SIZE'], 'WRITE_SIZE_KB': m['WRITE_SIZE'],
         'fetch_correction': 'x2: gfx950 FETCH_SIZE reports half of 16 B/lane reads (MI355X_MICROARCH.md, HBM section)',
         'traffic_bytes_per_launch': int(m['FETCH_SIZE']*2*1024 + m['WRITE_SIZE']*1024),
         'TCC_HIT_sum': m.get('TCC_HIT_sum'), 'TCC_MISS_sum': m.get('TCC_MISS_sum'), 'launches_counted': len(acc['FETCH_SIZE']),
         'collected': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate passes (tools/pmc_bench_kernel.sh) over bench.py --steps 20 --warmup 5 (configs[1]: 64 grids x 2000 points)'}
    json.dump(res, open('$out/hbm.json','w'), indent=1)
    print(json.dumps(res))
PY
