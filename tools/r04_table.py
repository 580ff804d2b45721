#!/usr/bin/env python3
"""DESIGN.md section 4's table from the committed files: profiles/r04_bench_<workload>.json (the bench line of the stats session),
profiles/r04_kernel_stats_<workload>.csv (rocprofv3 --kernel-trace --stats of the same command) and ik_amd/kernel_stats.json (the PMC
sessions' fold).  Prints markdown rows; nothing is retyped.
    python tools/r04_table.py"""
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["cassie_leg", "ur5", "ur10", "ur5_clamp", "ur10_clamp", "arm7", "ur5_two_tasks", "cassie_full_body", "cassie_demo", "cassie_demo_posture",
         "cassie_demo_pinned", "cassie_demo_pinned_posture", "cassie_demo_pik", "cassie_two_feet_pik", "ur5_pos_then_ori_pik", "cassie_three_feet"]
stats = json.load(open(os.path.join(ROOT, "ik_amd", "kernel_stats.json")))


def rocprof_avg_us(w, kernel_ms):
    """average duration of the timed kernel in the rocprofv3 table: the row whose average is closest to the bench line's"""
    p = os.path.join(ROOT, "profiles", "r04_kernel_stats_%s.csv" % w)
    if not os.path.exists(p):
        return None
    rows = [r for r in csv.DictReader(open(p)) if int(r["Calls"]) >= 10]
    if not rows:
        return None
    best = min(rows, key=lambda r: abs(float(r["AverageNs"]) / 1e6 - kernel_ms))
    return float(best["AverageNs"]) / 1e3, int(best["Calls"])


print("| workload (`bench.py --workload`) | kernel | kernel time (HIP events; rocprofv3 average) | solves/s (`value`) | flop / solve (PMC) | FP64 frac (78.6 TFLOP/s) | HBM traffic / algorithmic | VALU-issuing share | scratch B / lane | CPU port, all threads |")
print("|---|---|---|---|---|---|---|---|---|---|")
for w in ORDER:
    p = os.path.join(ROOT, "profiles", "r04_bench_%s.json" % w)
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r, h, c = d["roofline"], d.get("hbm_roofline") or {}, d["config"]
    k = c["kernel"]
    rec = stats.get(k, {})
    pmc = rec.get("pmc", {})
    share = pmc.get("SQ_ACTIVE_INST_VALU", 0) / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAVE_CYCLES") else None
    ms = r["kernel_ms"]
    rp = rocprof_avg_us(w, ms)
    alg = (h.get("algorithmic_bytes_per_solve") or r.get("algorithmic_bytes_per_solve") or 0) * c["batch_per_gpu"]
    traffic = r.get("traffic")
    cpu = (d.get("cpu_baseline") or {}).get("value")
    print("| %s | `%s` | %.3f ms; %s | %.3g | %s | %s | %s | %s | %s | %s |" % (
        w, k, ms, ("%.1f µs × %d" % rp) if rp else "—", d["value"],
        ("%d" % round(r["flop_per_solve"])) if r.get("flop_per_solve") else "—",
        ("%.3f" % r["frac"]) if r.get("bound") == "fp64_valu" else "—",
        ("%.1f / %.1f MB" % (traffic / 1e6, alg / 1e6)) if traffic else "—",
        ("%.2f" % share) if share else "—",
        rec.get("stamp", {}).get("scratch_size", "—"),
        ("%.3g" % cpu) if cpu else "—"))
