#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_r4e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TD_[A-Z_0-9a-z]+)\b" | sort -u | head -150 > $out/counters.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  OCOCC_STAGED_CONV=0 timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_conv.py --iters 3 --cin 64 --cout 128 --mode fwd > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('$out/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'gather_gemm_stream_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(f'{k:40s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
