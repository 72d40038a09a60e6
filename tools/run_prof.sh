# rocprofv3 passes for the round's profile evidence; outputs under gpurun_out/prof_* (copy summaries to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 $ARGS > gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 $ARGS > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 $ARGS > gpurun_out/prof_write.log 2>&1
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20
