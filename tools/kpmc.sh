# Counters of ONE kernel family in a full PAAC update at 8 192 envs (single stream): duration, HBM-side fetch/write bytes, SQ busy/wait.
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 ("exceeds the capabilities of the hardware"), and a refused pass can hang: timeouts.
# usage: bash tools/kpmc.sh NAME_REGEX [env assignments | --bench-flags ...]     (on the GPU box; every PMC group in its own pass)
set -e
RX="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
EXTRA=""
for kv in "$@"; do case "$kv" in --*) EXTRA="$EXTRA $kv";; *) export "$kv";; esac; done
mkdir -p gpurun_out
ARGS="bench.py --envs 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --single-stream $EXTRA"
D=gpurun_out/kpmc
rm -rf $D
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc FETCH_SIZE --output-format csv -d $D/mem -- python3 $ARGS > $D.mem.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc WRITE_SIZE --output-format csv -d $D/memw -- python3 $ARGS > $D.memw.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $D/sq -- python3 $ARGS > $D.sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $D/inst -- python3 $ARGS > $D.inst.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for part in ("mem", "memw", "sq", "inst"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/kpmc/%s/*/*counter_collection.csv" % part):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:90]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, c in acc.items():
        print(part, k)
        for name, v in sorted(c.items()):
            print("    %-28s per launch %.4g  (launches %d)" % (name, v / n[(k, name)], n[(k, name)]))
PY
rm -rf $D
