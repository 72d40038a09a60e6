#!/bin/bash
# A/B of one library build under two environments on ONE box: tools/ab_env.sh "VAR=a" "VAR=b" [bench args]; alternates a b a b
A="$1"; B="$2"; shift 2
for i in 1 2; do
  for E in "$A" "$B"; do
    echo "== $E"
    env $E python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2), d['ms_per_step_spread'], 'host_enqueue', d.get('host_enqueue_ms_per_update'), d.get('host',{}).get('per_rank'))"
  done
done
