# SQ counters of the GEMM microbenchmark kernels (tools/ubench/gemm_f16x3): where the wave cycles go
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_ub
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_ub/$tag -- tools/ubench/gemm_f16x3 > gpurun_out/pmc_ub_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_ub_$tag.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ub/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rowk" not in k: continue
        m = re.search(r"(rowk\w*)<([^>]*)>", k)
        name = (m.group(1) + "<" + m.group(2) + ">") if m else k[:60]
        key = (name, r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: "%.3g" % (sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
rm -rf gpurun_out/pmc_ub
