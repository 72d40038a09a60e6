# SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES per GEMM kernel (own PMC pass) + durations, one 8 192-env update on a single stream
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --envs 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --single-stream"
rm -rf gpurun_out/tn_sq gpurun_out/tn_st
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/tn_sq -- python3 $ARGS > gpurun_out/tn_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tn_st -- python3 $ARGS > gpurun_out/tn_st.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tn_sq/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
st = {r["Name"]: r for r in csv.DictReader(open(glob.glob("gpurun_out/tn_st/*/*_kernel_stats.csv")[0]))}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if "gemm" in k:
        print("%-70s conflict/wave_cycles %.4f  avg %8.1f us" % (k[:70], v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_WAVE_CYCLES"], 1), float(st[k]["AverageNs"]) / 1e3 if k in st else -1))
PY
rm -rf gpurun_out/tn_sq gpurun_out/tn_st
