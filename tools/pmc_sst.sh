#!/bin/bash
# SQ / LDS counters of the fused SST encoder-layer kernels inside the SST bench step (bounded PMC passes, one counter
# set per run as the guide prescribes).  usage: tools/pmc_sst.sh [out-tag]
tag=${1:-pmc_sst}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload sst --steps 2 --warmup 1 > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
for kern in ('window_attn_block_fwd_kernel', 'token_ffn_block_fwd_kernel', 'window_attn_block_bwd_kernel', 'token_ffn_block_bwd_kernel', 'token_wgrad_kernel'):
    acc=collections.defaultdict(list)
    for f in sorted(glob.glob('$out/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(kern)
    for k,v in sorted(acc.items()):
        print(f'  {k:34s} mean {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
