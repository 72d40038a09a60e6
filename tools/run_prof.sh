# The round's rocprofv3 evidence on the GPU box (kernel-trace/stats and every PMC group in its own pass; the program directly after `--`).
#   part A   one full PAAC update at 8 192 envs (ONE 81 920-sample chunk per step, the timed configuration's chunk), single stream:
#            per-kernel stats, FETCH_SIZE, WRITE_SIZE, SQ counters.  A second argument `interior` runs it on the interior_policy state
#            distribution (bench.py --interior) and writes the *_interior files the bench's interior leg reads.
#   part B   the TIMED configuration (32 768 envs, four streams): kernel trace (busy union, concurrency), then serialised PMC
#            passes (SQ busy / MFMA busy, FETCH_SIZE, WRITE_SIZE)
#   part C   env-only (random policy): the Swarm step kernel's duration and HBM traffic
# usage: bash tools/run_prof.sh <round, e.g. 05> A|B|C [interior]        outputs under gpurun_out/r<round>_*; copy what is judged to profiles/
set -e
RND=$1; PART=$2; VAR=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SFX=""; XARG=""
if [ "$VAR" = "interior" ]; then SFX="_interior"; XARG="--interior"; fi
P=gpurun_out/r${RND}${SFX}
note() { echo "$(date +%T) $1" >> gpurun_out/r${RND}_prof.progress; }
if [ "$PART" = "A" ]; then
  ARGS="bench.py --envs 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-extras --single-stream $XARG"
  note "A$SFX stats";  rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_A_stats -- python3 $ARGS > ${P}_A_stats.log 2>&1
  note "A$SFX fetch";  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${P}_A_fetch -- python3 $ARGS > ${P}_A_fetch.log 2>&1
  note "A$SFX write";  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${P}_A_write -- python3 $ARGS > ${P}_A_write.log 2>&1
  note "A$SFX sq";     rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d ${P}_A_sq -- python3 $ARGS > ${P}_A_sq.log 2>&1
  note "A$SFX done"
  python3 tools/summarize_prof.py $RND A $VAR
  rm -rf ${P}_A_stats ${P}_A_fetch ${P}_A_write ${P}_A_sq      # raw traces stay on the box (gpurun_out is capped at 64 MiB)
elif [ "$PART" = "B" ]; then
  ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $XARG"
  note "B trace";  rocprofv3 --kernel-trace --output-format csv -d ${P}_B_trace -- python3 $ARGS > ${P}_B_trace.log 2>&1
  ARGS1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $XARG"
  note "B sq";     rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d ${P}_B_sq -- python3 $ARGS1 > ${P}_B_sq.log 2>&1
  note "B fetch";  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${P}_B_fetch -- python3 $ARGS1 > ${P}_B_fetch.log 2>&1
  note "B write";  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${P}_B_write -- python3 $ARGS1 > ${P}_B_write.log 2>&1
  note "B done"
  python3 tools/summarize_prof.py $RND B $VAR
  rm -rf ${P}_B_trace ${P}_B_sq ${P}_B_fetch ${P}_B_write
else
  ARGS="bench.py --policy random --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
  note "C stats";  rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_C_stats -- python3 $ARGS > ${P}_C_stats.log 2>&1
  note "C fetch";  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${P}_C_fetch -- python3 $ARGS > ${P}_C_fetch.log 2>&1
  note "C write";  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${P}_C_write -- python3 $ARGS > ${P}_C_write.log 2>&1
  note "C done"
  python3 tools/summarize_prof.py $RND C
  rm -rf ${P}_C_stats ${P}_C_fetch ${P}_C_write
fi
