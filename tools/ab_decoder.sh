# A/B: the decoder's backward modes (occ/fused_mlp.py BACKWARD_MODE) inside the configs[2] step
for mode in fused recompute chain; do
  for b in 4 64; do
    OCOCC_DECODER_BACKWARD=$mode python bench.py --workload ococcnet --tracklets $b --steps $([ $b = 64 ] && echo 10 || echo 30) --warmup 5 --no-cpu-baseline > gpurun_out/r05g_dec_${mode}_b$b.json 2> gpurun_out/r05g_dec_${mode}_b$b.err
    python - <<PY
import json
d=json.load(open('gpurun_out/r05g_dec_${mode}_b$b.json'))
print('$mode B=$b ms/step', d['ms_per_step'], 'decoder fwd ms', d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k:(v['avg_us'],v['tflops']) for k,v in d['roofline'].get('per_kernel',{}).items()})
PY
  done
done
