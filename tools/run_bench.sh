set -e
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || (cat gpurun_out/bench_r01.err; exit 1)
cat gpurun_out/bench_r01.json
