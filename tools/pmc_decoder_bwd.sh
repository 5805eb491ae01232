#!/bin/bash
# SQ counters of occ_mlp_bwd_kernel (tools/probe/decoder_train_bench.py), one counter set per pass.
# usage: tools/pmc_decoder_bwd.sh [out-tag] [lib-override]
tag=${1:-pmc_dec_bwd}
if [ -n "$2" ]; then export OCOCC_LIB_PATH=$GRAFT_REPO_ROOT/$2; fi
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/probe/decoder_train_bench.py 262144 0.1 > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
for kern in ('occ_mlp_bwd_kernel', 'occ_mlp_fwd_kernel'):
    acc=collections.defaultdict(list)
    for f in sorted(glob.glob('$out/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(kern)
    for k,v in sorted(acc.items()):
        print(f'  {k:34s} mean {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
