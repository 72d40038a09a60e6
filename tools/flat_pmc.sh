# HBM-side bytes of the flat-net kernels (rollout, training forward, backward): bash tools/flat_pmc.sh   (on the GPU box)
# FETCH_SIZE and WRITE_SIZE in separate passes (gfx950: FETCH_SIZE x2 correction, unit KB), as tools/kpmc.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=gpurun_out/flatpmc
rm -rf $D
RX="flat_rollout_kernel|flat_backward_fast_kernel|flat_forward_fast_kernel"
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc FETCH_SIZE --output-format csv -d $D/mem -- python3 tools/flat_update_times.py 1 > $D.mem.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc WRITE_SIZE --output-format csv -d $D/memw -- python3 tools/flat_update_times.py 1 > $D.memw.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("mem", "memw"):
    for f in glob.glob("gpurun_out/flatpmc/%s/*/*counter_collection.csv" % part):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    for name, v in sorted(c.items()):
        v = sorted(v)
        scale = 2.0 if name == "FETCH_SIZE" else 1.0
        print("%-62s %-11s launches %3d  MB per launch: min %.1f  max %.1f" % (k, name, len(v), v[0] * scale / 1024, v[-1] * scale / 1024))
PY
rm -rf $D
