# SQ counters of the MFMA GEMM kernels for one full update at E=4096 (own pass, kernel-trace only)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq && mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --envs 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/pmc_sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"]
with open("gpurun_out/pmc_sq_summary.csv", "w") as out:
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
        out.write('"%s",%d,' % (k, cnt[k]) + ",".join("%.4g" % (agg[k][n] / max(cnt[k], 1)) for n in names) + "\n")
for line in open("gpurun_out/pmc_sq_summary.csv").read().splitlines()[:26]:
    print(line[:60].ljust(60), line.split('",')[-1] if '",' in line else "")
PY
