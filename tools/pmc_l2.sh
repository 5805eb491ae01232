#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <kernel-substring> <script.py> [args...]
# L2 / vector-L1 counters of one kernel (one counter set per run): hit rate, requests, where the waves wait.
tag=$1; kern=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 "$@" > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('$out/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if '$kern' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f'  {k:34s} mean {sum(v)/len(v):16.0f}  min {min(v):14.0f} max {max(v):14.0f} (n={len(v)})')
PY
