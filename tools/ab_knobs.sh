# times one bench configuration under several environments on ONE box; prints ms per update and one GEMM family
# usage: bash tools/ab_knobs.sh <family> "<ENV=.. ENV=..>" "<...>" ...
FAM=$1; shift
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > gpurun_out/abk.json 2> gpurun_out/abk.err || { tail -5 gpurun_out/abk.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/abk.json'));r=d['roofline'];f=r['by_family'].get('$FAM',{})
print('[$E]', round(d['ms_per_step'],1),'ms/update; $FAM', round(f.get('ms',0),1),'ms', round(f.get('achieved',0),1),'TF; all gemm', round(r['gemm_ms_total'],1),'ms')"
done; done
