# rocprofv3 passes for the round's profile evidence (per-kernel stats + HBM traffic counters in separate passes).
# usage: bash tools/run_prof.sh <tag> <bench args...>      outputs: gpurun_out/<tag>_summary.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ARGS="bench.py --no-cpu-baseline --no-extras --single-stream $@"
echo "stats pass" >> gpurun_out/prof_${TAG}.progress
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 $ARGS > gpurun_out/prof_${TAG}_stats.log 2>&1
echo "fetch pass" >> gpurun_out/prof_${TAG}.progress
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -- python3 $ARGS > gpurun_out/prof_${TAG}_fetch.log 2>&1
echo "write pass" >> gpurun_out/prof_${TAG}.progress
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -- python3 $ARGS > gpurun_out/prof_${TAG}_write.log 2>&1
python3 tools/summarize_prof.py $TAG
