set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_flat
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_flat -- python3 tools/bench_trade_quick.py > gpurun_out/tr_flat.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tr_flat/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[(k, r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-42s grid %8s calls %4d avg %9.1f us total %8.1f ms" % (k, g, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
PY
rm -rf gpurun_out/tr_flat
