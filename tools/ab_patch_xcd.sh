# tile order of the dense1 patch weight gradient (GRL_PATCH_WGRAD_XCD = 1: default since the end of round 3, a row slice per XCD; 2: the J tiles of an I tile per XCD; 0: launch order) on one box
for V in 2 1 0 2 1; do
  GRL_PATCH_WGRAD_XCD=$V python3 bench.py --no-cpu-baseline --no-extras --steps 3 2>/dev/null > gpurun_out/ab_pxcd_$V.json
  python3 -c "import json;d=json.loads(open('gpurun_out/ab_pxcd_$V.json').read().strip().splitlines()[-1]);print('wgrad xcd order $V', round(d['ms_per_step'],1), round(d['roofline']['by_family']['dense1_patch_wgrad']['ms'],1))"
done
