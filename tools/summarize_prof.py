"""Condense the rocprofv3 passes of tools/run_prof.sh <round> <part> [interior] into the small files committed under profiles/
(r<round>_*; with `interior`: r<round>_*_interior, the state distribution of bench.py's interior_policy leg)."""
import collections
import csv
import glob
import json
import sys

rnd, part = sys.argv[1], sys.argv[2]
sfx = "_interior" if len(sys.argv) > 3 and sys.argv[3] == "interior" else ""
R = "gpurun_out/r%s" % rnd      # output prefix
P = "%s%s_%s" % (R, sfx, part)


def one(pattern):
    f = glob.glob(pattern)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


def counters(tag):
    """{kernel: {counter: [values per dispatch]}} -- a counter's value of one dispatch is the sum over its rows (one per XCD/SE)"""
    f = one("%s_%s/*/*_counter_collection.csv" % (P, tag))
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: list(d.values()) for c, d in cs.items()} for k, cs in per.items()}


def is_gemm(k):
    return k.startswith("void grl::gemm_rowk") or k.startswith("void grl::gemm_tn")


def family(k):
    """the bench line's GEMM families (PT_* tags of net_conv.hip) from a kernel's template arguments"""
    if not is_gemm(k):
        return None
    tn = k.startswith("void grl::gemm_tn")
    if "PatchRows" in k:
        return "dense1_patch_wgrad" if tn else ("dense1_patch_fwd" if "EpiPatchFwd" in k else "dense1_patch_dgrad")
    if "SlotRowsP" in k or "SlotsToPatch" in k:      # conv3's per-agent corrections: slot products (GRL_NET_EXPAND3=prod) or the patch gather
        return "slot_products_fwd"
    if "SlotGatherT3P" in k:
        return "slot_wgrad" if tn else "slot_dgrad"
    if "CorrRows" in k or "T2SlotGather" in k:
        return "conv2_class_corrections"
    if "DenseRowsKU" in k or ("ConvRowsList" in k and not tn) or "GatherConv2ReluRows" in k and not tn:
        return "per_env_fwd"
    if "GatherT3Rows" in k or "GatherT2Rows" in k or "DenseRowsNU" in k:
        return "per_env_dgrad"
    if tn and ("ConvRowsList" in k or "DenseRowsIU" in k or "GatherConv" in k):
        return "per_env_wgrad"
    if tn:
        return "dense_small_wgrad"
    if "EpiGradSum" in k:
        return "dense_small_dgrad"
    if "EpiBiasAct" in k:
        return "dense_small_fwd"
    return "other"


if part in ("A", "C"):
    rows = list(csv.DictReader(open(one(P + "_stats/*/*_kernel_stats.csv"))))
    fetch, write = counters("fetch"), counters("write")
    avg = lambda d, k, c: (sum(d[k][c]) / len(d[k][c])) if k in d and c in d[k] else float("nan")
    out = "%s_%s%s_summary.csv" % (R, "full_update_e8192" if part == "A" else "envonly", sfx)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct,FETCH_SIZE_avg_KB,WRITE_SIZE_avg_KB\n")
        for r in rows:
            k = r["Name"]
            f.write('"%s",%s,%.3f,%.2f,%.2f,%.1f,%.1f\n' % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                          100 * float(r["TotalDurationNs"]) / tot, avg(fetch, k, "FETCH_SIZE"), avg(write, k, "WRITE_SIZE")))
    print(open(out).read()[:3000])
    if part == "A":
        g_calls = sum(int(r["Calls"]) for r in rows if is_gemm(r["Name"]))
        g_ns = sum(float(r["TotalDurationNs"]) for r in rows if is_gemm(r["Name"]))
        g_bytes = sum(int(r["Calls"]) * 1024.0 * (2.0 * avg(fetch, r["Name"], "FETCH_SIZE") + avg(write, r["Name"], "WRITE_SIZE"))
                      for r in rows if is_gemm(r["Name"]))
        updates = 3       # --steps 2 + the one profiled extra update of bench.py
        # the engine's initial reset (every env, with the Swarm burn-in: ONE launch of the reset kernel, milliseconds, before the
        # first update) is in the trace but in no update: out of the per-update figure and of the shares
        setup_ns = sum(float(r["MaxNs"]) for r in rows if "swarm_kernel<2" in r["Name"] and float(r["MaxNs"]) > 50 * float(r["MinNs"]))
        tot -= setup_ns
        non = tot - g_ns
        fam = collections.defaultdict(lambda: [0, 0.0, 0.0])      # launches, ns, bytes
        for r in rows:
            fm = family(r["Name"])
            if fm:
                fam[fm][0] += int(r["Calls"]); fam[fm][1] += float(r["TotalDurationNs"])
                fam[fm][2] += int(r["Calls"]) * 1024.0 * (2.0 * avg(fetch, r["Name"], "FETCH_SIZE") + avg(write, r["Name"], "WRITE_SIZE"))
        by_family = {k: {"launches": v[0], "avg_launch_us": v[1] / 1e3 / max(v[0], 1), "hbm_bytes_per_launch": v[2] / max(v[0], 1),
                         "hbm_TBps": (v[2] / v[1] * 1e9 / 1e12) if v[1] else 0.0} for k, v in fam.items()}
        json.dump({"kernels": "gemm_rowk / gemm_tn (all instantiations)", "launches": g_calls, "avg_launch_us": g_ns / 1e3 / max(g_calls, 1),
                   "hbm_bytes_per_launch": g_bytes / max(g_calls, 1), "hbm_bytes_total": g_bytes, "updates_in_this_pass": updates,
                   "kernel_ms_per_update": tot / 1e6 / updates, "setup_ms_excluded": setup_ns / 1e6, "launches_per_update": sum(int(r["Calls"]) for r in rows) / updates,
                   "by_family": by_family, "gemm_share_of_kernel_time": g_ns / tot, "non_gemm_share_of_kernel_time": non / tot,
                   "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is, unit KB; per-kernel averages weighted by calls",
                   "source": "tools/run_prof.sh %s A %s: bench.py --envs 8192 --steps 2 --warmup 0 --single-stream%s (3 updates: 2 timed + the HIP-event pass)"
                             % (rnd, sfx[1:], " --interior" if sfx else "")},
                  open("%s_gemm_traffic%s.json" % (R, sfx), "w"), indent=1)
        print(open("%s_gemm_traffic%s.json" % (R, sfx)).read())
        sq = counters("sq")
        names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"]
        with open("%s_sq_counters_e8192%s.csv" % (R, sfx), "w") as f:
            f.write("kernel,launches," + ",".join(names) + "\n")
            for k in sorted(sq, key=lambda k: -sum(sq[k].get("SQ_WAVE_CYCLES", [0]))):
                n = len(sq[k].get("SQ_WAVE_CYCLES", [1]))
                f.write('"%s",%d,' % (k, n) + ",".join("%.4g" % (sum(sq[k].get(c, [0])) / max(n, 1)) for c in names) + "\n")
    else:
        k = [r for r in rows if "swarm_kernel<0" in r["Name"]][0]
        name = k["Name"]
        b = 1024.0 * (2.0 * avg(fetch, name, "FETCH_SIZE") + avg(write, name, "WRITE_SIZE"))
        json.dump({"kernel": name, "envs": 32768, "fast_math": False, "avg_us": float(k["AverageNs"]) / 1e3, "hbm_bytes_per_launch": b,
                   "algorithmic_bytes_per_launch": 32768 * 4613, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is, unit KB",
                   "source": "tools/run_prof.sh %s C" % rnd}, open("gpurun_out/swarm_step_traffic.json", "w"), indent=1)
        print(open("gpurun_out/swarm_step_traffic.json").read())
else:
    # ---- the timed configuration: 32 768 envs, four streams
    f = one(P + "_trace/*/*_kernel_trace.csv")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    # the two timed updates end with adam_kernel; the window = from the adam of the warm-up update to the adam ending timed update 2
    adams = [e for s, e, k in rows if k.startswith("grl::adam_kernel")]
    lo, hi = adams[0], adams[2]
    sel = [r for r in rows if lo <= r[0] and r[1] <= hi]
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = hi - lo
    dur = sum(e - s for s, e, _ in sel)
    gemm_dur = sum(e - s for s, e, k in sel if is_gemm(k))
    res = {"config": "bench.py default: Swarm-v0 32 768 envs, T=20, four streams; two timed updates between the Adam launches of the trace",
           "wall_ms_per_update": wall / 2e6, "kernels_per_update": len(sel) / 2, "sum_of_kernel_durations_ms_per_update": dur / 2e6,
           "union_busy_fraction_of_wall": busy / wall, "mean_concurrency": dur / busy, "gemm_share_of_summed_durations": gemm_dur / dur}
    # serialised PMC passes (one update each, --warmup 0 --steps 1 + the profiled extra update = 2 updates)
    sq, fetch, write = counters("sq"), counters("fetch"), counters("write")
    tot = lambda d, c, pred=lambda k: True: sum(sum(v.get(c, [])) for k, v in d.items() if pred(k))
    upd = 2.0
    res["pmc_updates_in_pass"] = upd
    mfma = tot(sq, "SQ_VALU_MFMA_BUSY_CYCLES") / upd
    sqbusy = tot(sq, "SQ_BUSY_CYCLES") / upd
    res["SQ_VALU_MFMA_BUSY_CYCLES_per_update"] = mfma
    res["SQ_BUSY_CYCLES_per_update"] = sqbusy
    # SQ_BUSY_CYCLES is summed over the 32 shader engines (calibrated on single kernels: value / duration = 32 x 2.0-2.3 GHz), each
    # with 32 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES sums SIMD-cycles (16 per v_mfma_f32_16x16x32_bf16): the matrix pipes' share of the
    # SIMD-cycles the GPU was busy is MFMA / (32 x SQ_BUSY)
    res["mfma_busy_fraction_of_busy_simd_cycles_all_kernels"] = mfma / (32.0 * sqbusy) if sqbusy else None
    g_busy = tot(sq, "SQ_BUSY_CYCLES", is_gemm)
    res["mfma_busy_fraction_of_busy_simd_cycles_gemm_kernels"] = tot(sq, "SQ_VALU_MFMA_BUSY_CYCLES", is_gemm) / (32.0 * g_busy) if g_busy else None
    res["mfma_busy_fraction_of_timed_wall_at_2p4GHz"] = mfma / (1024 * 2.4e9 * wall / 2e9)
    f2 = one(P + "_sq/*/*_counter_collection.csv")
    dur_ns, busy_c = 0.0, 0.0
    for r in csv.DictReader(open(f2)):
        if r["Counter_Name"] == "SQ_BUSY_CYCLES" and is_gemm(r["Kernel_Name"]):
            dur_ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy_c += float(r["Counter_Value"])
    res["shader_clock_GHz_during_gemm_kernels"] = busy_c / 32.0 / dur_ns if dur_ns else None
    hbm = 1024.0 * (2.0 * tot(fetch, "FETCH_SIZE") + tot(write, "WRITE_SIZE")) / upd
    res["hbm_side_bytes_per_update"] = hbm
    res["hbm_side_TBps_over_timed_wall"] = hbm / (wall / 2e9) / 1e12
    res["notes"] = ("PMC passes serialise the dispatches, so the counters are per-kernel totals of the same work, not of the overlapped run; "
                    "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KB; L2-miss traffic includes what the 256 MB Infinity Cache serves")
    json.dump(res, open("%s_timed_config_32768%s.json" % (R, sfx), "w"), indent=1)
    print(json.dumps(res, indent=1))
