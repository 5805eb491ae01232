#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <kernel-substring> [prof_conv args]  -- SQ / LDS counters of one conv kernel
# (each counter set in its own bounded rocprofv3 pass; FETCH_SIZE/WRITE_SIZE passes are separate: tools/pmc_hbm.sh)
tag=$1; kern=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_conv.py --iters 3 "$@" > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('$out/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if '$kern' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(f'{k:40s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
