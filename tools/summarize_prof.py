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
