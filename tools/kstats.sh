# Per-kernel durations of full PAAC updates at 8 192 envs (one 81 920-sample chunk per step, single stream):
# rocprofv3 --kernel-trace --stats; prints every kernel above 0.3 %, the GEMM / non-GEMM split and the launches per update.
# usage: bash tools/kstats.sh TAG [env assignments | --bench-flags ...]     -> gpurun_out/ks_TAG.csv (+ the table on stdout)
set -e
TAG=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks_$TAG && mkdir -p gpurun_out
EXTRA=""
for kv in "$@"; do case "$kv" in --*) EXTRA="$EXTRA $kv";; *) export "$kv";; esac; done
UPD=3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$TAG -- python3 bench.py --envs 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-extras --single-stream $EXTRA > gpurun_out/ks_$TAG.log 2>&1
python3 - "$TAG" "$UPD" <<'PY'
import csv, glob, sys
tag, upd = sys.argv[1], int(sys.argv[2])
f = glob.glob("gpurun_out/ks_%s/*/*_kernel_stats.csv" % tag)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
gemm = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_rowk" in r["Name"] or "gemm_tn" in r["Name"])
calls = sum(int(r["Calls"]) for r in rows)
with open("gpurun_out/ks_%s.csv" % tag, "w") as o:
    o.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        o.write('"%s",%s,%.3f,%.2f,%.2f\n' % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("%s: %d updates in the pass (2 timed + the roofline pass): %.1f ms of kernels per update, GEMM %.1f (%.1f %%), non-GEMM %.1f (%.1f %%), %d launches per update"
      % (tag, upd, tot / 1e6 / upd, gemm / 1e6 / upd, 100 * gemm / tot, (tot - gemm) / 1e6 / upd, 100 * (tot - gemm) / tot, calls // upd))
for r in rows:
    pct = 100 * float(r["TotalDurationNs"]) / tot
    if pct < 0.3: continue
    n = r["Name"]
    n = n.split("(")[0][-60:] if "gemm" not in n else n[10:100]
    print("%-92s calls %5s avg %8.1f us %5.2f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, pct))
PY
rm -rf gpurun_out/ks_$TAG
