# per-kernel durations for one full update at E=4096 (one 40 960-sample chunk per step): fast rocprofv3 stats pass
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --envs ${KS_ENVS:-8192} --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream > gpurun_out/ks.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-96s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
