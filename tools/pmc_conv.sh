#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <kernel-substring> [prof_conv args]  -- SQ / LDS / TA counters of one conv kernel
tag=$1; kern=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_conv.py --iters 3 "$@" > /dev/null 2>$out/err$i.txt
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
