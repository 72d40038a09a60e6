# A/B of one build under two environments on ONE box (box-to-box clocks differ by a few per cent):
#   gpurun -- 'bash tools/ab_env.sh "GRL_NET_EXPAND2=lds" "GRL_NET_EXPAND2=gemm" [bench args]'   runs A B A B, prints ms per update
A=$1; B=$2; shift; shift
for v in A B A B; do
  if [ $v = A ]; then E="$A"; else E="$B"; fi
  env $E python bench.py --no-cpu-baseline --no-extras --steps 3 "$@" > gpurun_out/abe_$v.json 2> gpurun_out/abe_$v.err || { tail -5 gpurun_out/abe_$v.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/abe_$v.json'));r=d['roofline']
print('$v ($E)', round(d['ms_per_step'],1),'ms/update', round(d['value']), 'env-steps/s; gemm', round(r.get('achieved',0),1), 'TF', round(r.get('gemm_ms_total',0),1),'ms; single-stream update', r['measured'].split('(')[1].split(')')[0])"
done
