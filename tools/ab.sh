# A/B timing of two builds of the library on ONE box (box-to-box clocks differ by a few per cent):
#   build variant A, cp golds-rl-gym_amd/lib/libgoldsrl.so golds-rl-gym_amd/lib/A.so; same for B; then
#   gpurun -- 'bash tools/ab.sh [bench args]'      runs A B A B and prints ms per update
L=golds-rl-gym_amd/lib
for v in A B A B; do
  cp $L/$v.so $L/libgoldsrl.so
  python bench.py --no-cpu-baseline --no-extras --steps 3 "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python -c "
import json,sys;d=json.load(open('gpurun_out/ab_$v.json'));r=d['roofline']
print('$v', round(d['ms_per_step'],1),'ms/update', round(d['value']), 'env-steps/s; gemm', round(r.get('achieved',0),1), 'TF', round(r.get('gemm_ms_total',0),1),'ms')"
done
