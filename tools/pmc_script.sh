#!/bin/bash
# usage: tools/pmc_script.sh <tag> <kernel-substring[,substring...]> <script.py> [args...]
# SQ counters (one counter set per run) of the kernels whose name contains one of the substrings, for any python script.
tag=$1; kern=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o p$i -- python3 "$@" > /dev/null 2>$out/err$i.txt || echo "pass $i failed/timeout"
done
python3 - <<PY
import csv, glob, collections
for kern in '$kern'.split(','):
    acc=collections.defaultdict(list)
    for f in sorted(glob.glob('$out/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(kern)
    for k,v in sorted(acc.items()):
        print(f'  {k:34s} mean {sum(v)/len(v):16.0f}  min {min(v):14.0f} max {max(v):14.0f} (n={len(v)})')
PY
