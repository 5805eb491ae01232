#!/bin/bash
# configs[2] steps with the SIRLayer as one launch per direction (OCOCC_SIR_FUSED=1, default) against one launch per block
# (=0), at 4 / 16 / 64 tracklets; at 64 also with the row threshold lifted (131 k points through the point_mlp kernels).
out=gpurun_out/ab_sir_fused.txt
: > $out
for rep in 1 2; do
for b in 4 16 64; do
  for fused in 1 0; do
    for maxrows in 80000 100000000; do
      if [ $b != 64 ] && [ $maxrows != 80000 ]; then continue; fi
      line=$(OCOCC_SIR_FUSED=$fused OCOCC_POINT_LAYER_MAX_ROWS=$maxrows python bench.py --workload ococcnet --tracklets $b --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
      echo "rep=$rep B=$b fused=$fused max_rows=$maxrows ms/step=$line" | tee -a $out
    done
  done
done
done
