for v in "default:" "s128:64x32:2.0,32x64:2.0" "sboth:32x64:2.0" "sall:"; do
  name=${v%%:*}; shapes=${v#*:}
  if [ "$name" = default ]; then unset OCOCC_TILE_SHAPES; else export OCOCC_TILE_SHAPES="$shapes"; fi
  python bench.py --no-also --no-cpu-baseline > gpurun_out/r05b_ab_$name.json 2> gpurun_out/r05b_ab_$name.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r05b_ab_$name.json'))
print('$name', d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['roofline']['per_kernel'].items()})
PY
done
