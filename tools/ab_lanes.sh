# update time against the number of streams ("lanes") the chunks are dealt to, on one box
for L in 4 6 8 3 5 4; do
  GRL_NET_LANES=$L python3 bench.py --no-cpu-baseline --no-extras --steps 3 2>/dev/null > gpurun_out/ab_lanes_$L.json
  python3 -c "import json;d=json.loads(open('gpurun_out/ab_lanes_$L.json').read().strip().splitlines()[-1]);print('lanes $L', round(d['ms_per_step'],1))"
done
