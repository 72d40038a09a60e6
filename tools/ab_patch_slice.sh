# rows per weight-gradient slice of the dense1 patch GEMM under the support masks (smaller slices: tighter unions, more slab tiles), one box
for V in 1024 512 2048 256 1024; do
  GRL_PATCH_SLICE=$V python3 bench.py --no-cpu-baseline --no-extras --steps 3 2>/dev/null > gpurun_out/ab_pslice_$V.json
  python3 -c "import json;d=json.loads(open('gpurun_out/ab_pslice_$V.json').read().strip().splitlines()[-1]);f=d['roofline']['by_family']['dense1_patch_wgrad'];print('slice rows $V', round(d['ms_per_step'],1), round(f['ms'],1), round(f['executed_share_of_the_5x5_patch'],3))"
done
