# rocprofv3 passes for the round's profile evidence; outputs under gpurun_out/prof_* (copy summaries to profiles/)
# usage: bash tools/run_prof.sh <tag> <bench args...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ARGS="bench.py --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 $ARGS > gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -- python3 $ARGS > gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -- python3 $ARGS > gpurun_out/prof_${TAG}_write.log 2>&1
python3 tools/summarize_prof.py $TAG
