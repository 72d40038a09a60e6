"""GPU busy time (union of kernel intervals over all streams) against wall time, from a rocprofv3 --kernel-trace CSV of the
default four-stream bench: tells a GPU-bound update from a host-enqueue-bound one."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last update: from the last swarm burn-in-free rollout start; simply take the final 40 % of the trace
t_end = max(r[1] for r in rows)
t_beg = rows[0][0]
# analyse the window [t_end - a ms, t_end - b ms]
hi = t_end - int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else t_end
cut = t_end - int(float(sys.argv[2]) * 1e6)
sel = [r for r in rows if cut <= r[0] < hi]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = sel[-1][1] - sel[0][0] if sel else 0
tot = sum(e - s for s, e, _ in sel)
print("window %.1f ms: kernels %d, sum of durations %.1f ms, union busy %.1f ms (%.1f %% of wall), mean concurrency %.2f" % (
    wall / 1e6, len(sel), tot / 1e6, busy / 1e6, 100.0 * busy / wall, tot / max(busy, 1)))
gaps = []
cur_e = None
for s, e, _ in sel:
    if cur_e is not None and s > cur_e: gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort(reverse=True)
print("idle gaps: %d, total %.1f ms, largest (us): %s" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))
