"""Timeline of ONE rollout step of one chunk from a rocprofv3 --kernel-trace CSV: every kernel between two consecutive Swarm step
launches, with its queue, duration and the gap to the previous kernel on that queue.  usage: python tools/trace_step.py <trace dir> [which step = 30]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
steps = [i for i, r in enumerate(rows) if "swarm_kernel<0" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
a, b = steps[k], steps[k + 1]
t0 = rows[a][1]
last = {}
tot = 0
for s, e, n, q in rows[a:b + 1]:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("grl::", "")
    tot += e - s
    print("q%s  +%8.1f us  dur %7.1f  gap %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, n[:100]))
print("step: %.1f us wall, %.1f us of kernels, %d launches" % ((rows[b][1] - t0) / 1e3, tot / 1e3, b - a))
