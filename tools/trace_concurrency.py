"""Four-stream timeline of the timed configuration from a rocprofv3 --kernel-trace CSV: how much of the wall clock runs with 1, 2, 3, 4+
kernels in flight, which kernels are the ones running ALONE (and for how long), per-queue busy shares, and the longest stretches during
which only small-grid kernels are resident.  usage: python tools/trace_concurrency.py <trace dir> [updates in the window = 2]
(window: bench.py --steps 2 --warmup 1 ends every update with adam_kernel; from the warm-up update's to the second timed update's --
the single-stream roofline pass that follows is left out)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f)):
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256) * max(1, int(r.get("Workgroup_Size_Y", 1) or 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), g // max(wg, 1)))
rows.sort()
adams = [e for s, e, k, q, w in rows if k.startswith("grl::adam_kernel")]
nupd = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = adams[0], adams[nupd]
sel = [r for r in rows if lo <= r[0] and r[1] <= hi]
wall = hi - lo
ev = []
for i, (s, e, k, q, w) in enumerate(sel):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live, t_prev = set(), lo
hist = collections.Counter()
alone = collections.Counter()
small_only = 0
for t, d, i in ev:
    dt = t - t_prev
    if dt > 0:
        hist[min(len(live), 5)] += dt
        if len(live) == 1:
            alone[short := re.sub(r"\(.*", "", sel[next(iter(live))][2])[-70:]] += dt
        if live and all(sel[j][4] < 512 for j in live):
            small_only += dt
    t_prev = t
    if d > 0: live.add(i)
    else: live.discard(i)
print("window: %d updates, %.1f ms per update, %d kernels per update" % (nupd, wall / nupd / 1e6, len(sel) // max(nupd, 1)))
print("kernels in flight -> share of the wall: " + ", ".join("%s%d: %.1f %%" % (">=" if k == 5 else "", k, 100.0 * v / wall) for k, v in sorted(hist.items())))
print("wall with only grids of < 512 workgroups resident: %.1f %% (%.1f ms per update)" % (100.0 * small_only / wall, small_only / nupd / 1e6))
print("running ALONE (ms per update):")
for k, v in alone.most_common(14):
    print("   %7.2f  %s" % (v / nupd / 1e6, k))
busy = collections.Counter()
for s, e, k, q, w in sel:
    busy[q] += e - s
print("per queue: kernel-resident share of the wall: " + ", ".join("%s: %.0f %%" % (q, 100.0 * v / wall) for q, v in sorted(busy.items())))
