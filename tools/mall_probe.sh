# Does the cache hierarchy behind L2 (the 256 MB Infinity Cache) serve the producer -> consumer traffic of the update?
# Two rocprofv3 kernel-stat passes of the same single-stream update at 8 192 envs: as shipped, and with a 512 MiB write in front
# of every GEMM and of the largest byte-moving helpers (GRL_NET_EVICT=1: the kernel then finds nothing of its inputs in L2 / MALL).
# Prints per-kernel average durations side by side.  usage: gpurun -- 'bash tools/mall_probe.sh'   -> gpurun_out/mall_probe.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ARGS="bench.py --envs 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream"
rm -rf gpurun_out/mp0 gpurun_out/mp1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mp0 -- python3 $ARGS > gpurun_out/mp0.log 2>&1
export GRL_NET_EVICT=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mp1 -- python3 $ARGS > gpurun_out/mp1.log 2>&1
python3 - <<'PY' | tee gpurun_out/mall_probe.txt
import csv, glob
def load(d):
    f = glob.glob(d + "/*/*_kernel_stats.csv")[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
a, b = load("gpurun_out/mp0"), load("gpurun_out/mp1")
print("%-100s %6s %10s %10s %7s" % ("kernel", "calls", "warm us", "evicted us", "ratio"))
ta = tb = 0.0
for k, (c, us) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:34]:
    if k not in b or "evict_kernel" in k:
        continue
    print("%-100s %6d %10.1f %10.1f %7.3f" % (k[:100], c, us, b[k][1], b[k][1] / us))
    if "gemm_" in k:
        ta += c * us; tb += b[k][0] * b[k][1]
print("all GEMM launches: warm %.1f ms, evicted %.1f ms, ratio %.3f" % (ta / 1e3, tb / 1e3, tb / ta))
PY
rm -rf gpurun_out/mp0 gpurun_out/mp1
