#!/bin/bash
# Several environments of one library build on ONE box, two alternating rounds: tools/ab_multi.sh "VAR=a" "VAR=b X=c" ... [-- bench args]
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2; do
  for E in "${ENVS[@]}"; do
    echo "== $E"
    env $E python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2), d['ms_per_step_spread']['min'], d['ms_per_step_spread']['max'], 'host_enqueue', d.get('host_enqueue_ms_per_update'))"
  done
done
