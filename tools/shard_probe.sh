#!/bin/bash
# Strong-scaling shard (32 768 / 8 = 4 096 envs per GPU): the full update at several chunk sizes, on one box.
# usage: tools/shard_probe.sh [envs] [chunk sizes...]
E=${1:-4096}; shift
CH=${@:-"0 20480 10240 5120"}
out=gpurun_out/shard_probe_$E.txt
: > $out
for c in $CH; do
  if [ "$c" = "0" ]; then unset GRL_NET_CHUNK; else export GRL_NET_CHUNK=$c; fi
  for rep in 1 2; do
    echo "== chunk=$c rep=$rep" >> $out
    python bench.py --envs $E --steps 10 --warmup 2 --no-legs --no-flat-configs --no-cpu-baseline --no-extras 2>>$out | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', j['ms_per_step'], j['ms_per_step_spread'], 'host_enqueue', j.get('host_enqueue_ms_per_update'))" >> $out || exit 1
  done
done
cat $out
