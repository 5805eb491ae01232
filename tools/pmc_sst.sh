#!/bin/bash
# MFMA utilisation of the window-attention kernels inside the SST bench step (bounded PMC passes)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sst
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload sst --steps 3 --warmup 1 > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
for kern in ('window_attn_fwd_kernel', 'window_attn_bwd_kernel'):
    acc=collections.defaultdict(list)
    for f in sorted(glob.glob('$out/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(kern)
    for k,v in sorted(acc.items()):
        print(f'  {k:34s} mean {sum(v)/len(v):16.0f}  sum {sum(v):18.0f} (n={len(v)})')
PY
