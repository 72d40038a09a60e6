"""Condense rocprofv3 CSV output (gpurun_out/prof_<tag>_{stats,fetch,write}) into small files for profiles/."""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
stats = glob.glob("gpurun_out/prof_%s_stats/*/*_kernel_stats.csv" % tag)[0]
rows = list(csv.DictReader(open(stats)))
pmc = {}
for name in ("fetch", "write"):
    f = glob.glob("gpurun_out/prof_%s_%s/*/*_counter_collection.csv" % (tag, name))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[name] = sum(v) / len(v)
with open("gpurun_out/%s_summary.csv" % tag, "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct,FETCH_SIZE_avg_KB,WRITE_SIZE_avg_KB\n")
    for r in rows:
        k = r["Name"]
        f.write('"%s",%s,%.3f,%.2f,%s,%.1f,%.1f\n' % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                     r["Percentage"], pmc.get(k, {}).get("fetch", float("nan")),
                                                     pmc.get(k, {}).get("write", float("nan"))))
print(open("gpurun_out/%s_summary.csv" % tag).read()[:6000])

# aggregate HBM-side traffic of the MFMA GEMM kernels (bench.py's roofline.traffic): FETCH_SIZE x2 (gfx950 wide
# coalesced reads count half, MI355X_MICROARCH.md HBM/rocprofv3 section), WRITE_SIZE as is, unit KB
import json
g_calls = 0
g_bytes = 0.0
g_ns = 0.0
for r in rows:
    k = r["Name"]
    if k.startswith("void grl::gemm_rowk") or k.startswith("void grl::gemm_tn"):
        c = int(r["Calls"])
        g_calls += c
        g_ns += float(r["TotalDurationNs"])
        g_bytes += c * 1024.0 * (2.0 * pmc.get(k, {}).get("fetch", 0.0) + pmc.get(k, {}).get("write", 0.0))
json.dump({"kernels": "gemm_rowk / gemm_tn (all instantiations)", "launches": g_calls, "avg_launch_us": g_ns / 1e3 / max(g_calls, 1),
           "hbm_bytes_per_launch": g_bytes / max(g_calls, 1), "hbm_bytes_total": g_bytes,
           "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is, unit KB; per-kernel averages weighted by calls",
           "source": "tools/run_prof.sh %s (rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)" % tag},
          open("gpurun_out/%s_gemm_traffic.json" % tag, "w"), indent=1)
print(open("gpurun_out/%s_gemm_traffic.json" % tag).read())
