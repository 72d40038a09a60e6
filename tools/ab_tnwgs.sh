cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for M in 512 1024 2048; do
rm -rf gpurun_out/ks
GRL_TN_WGS=$M rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --envs 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --single-stream > gpurun_out/ks.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/ks/*/*_kernel_stats.csv")[0]
t=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "gemm_tn" in n and ("ConvRowsList" in n or "SlotGatherT3P" in n or "DenseRowsIU" in n): print("tn_wgs $M", n[15:75], r["Calls"], round(float(r["AverageNs"])/1e3,1)); t+=float(r["AverageNs"])/1e3
    if "slab_reduce_kernel" in n or "slab_reduce_umask" in n or "slab_reduce_taps" in n: print("tn_wgs $M", n[:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
