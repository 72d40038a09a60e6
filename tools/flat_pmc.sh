set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_flat
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_flat -- python3 tools/bench_trade_quick.py > gpurun_out/pmc_flat.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_flat/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "flat_forward" in r["Kernel_Name"] and int(r["Grid_Size"]) == 128 * 256:
        agg["fwd8192"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg["fwd8192"]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(cs["SQ_WAVE_CYCLES"]))
PY
rm -rf gpurun_out/pmc_flat
