#!/usr/bin/env python3
"""Fold ONE PMC session (tools/pmc_session.sh: one workload, one build) into ik_amd/kernel_stats.json and profiles/.

One build, one counter set: every pass directory must hold exactly ONE rocprofv3 session (one *counter_collection.csv); the
dispatches of the workload's solve kernel must agree on Kernel_Name, Scratch_Size, VGPR_Count and LDS_Block_Size within and across
the passes -- anything else is refused.  The entry is stamped with the kernel's symbol, its register / scratch / LDS figures, the
identity of the device sources the session was run from (tools/source_stamp.py) and, when run inside the git checkout, HEAD;
bench.py replays `traffic` / `flop_per_solve` only from an entry whose source stamp matches the tree it runs from.

HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE
counts a coalesced streaming read at half its bytes (MI355X_MICROARCH.md, section HBM) -- calibrated on the chain kernel's own known
byte count: it reads 28 doubles per problem = 14.68 MB at B = 65536 and FETCH_SIZE reports 7.31 MiB x 2 = 14.97 MB (the rest is the
constant table).

    python tools/pmc_to_stats.py gpurun_out/r03_pmc_cassie_leg profiles/r03_pmc leg
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOLVE = ("dls_", "pik_", "ikgpu_hot_", "ikgpu_lane_")   # (the last two: kernels compiled at run time, rtc.cpp)
NOT_SOLVE = ("pass_through",)


def solve_rows(path):
    rows = [r for r in csv.DictReader(open(path)) if any(s in r["Kernel_Name"] for s in SOLVE) and not any(s in r["Kernel_Name"] for s in NOT_SOLVE)]
    if not rows:
        return rows
    # the workload's solve kernel: the symbol with the most dispatches (bench.py --timed-only launches nothing else of that family)
    top = collections.Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][0]
    return [r for r in rows if r["Kernel_Name"] == top]


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    bench = json.load(open(os.path.join(src, "bench.json")))
    kernel = bench["config"]["kernel"]
    stamp = json.load(open(os.path.join(src, "source_stamp.json")))
    agg = collections.defaultdict(list)
    ident = None
    passes = sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d)))
    for name in passes:
        files = glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)
        if len(files) != 1:
            sys.exit("pass %s of %s holds %d rocprofv3 sessions (expected exactly one): %s" % (name, src, len(files), files))
        rows = solve_rows(files[0])
        if not rows:
            sys.exit("pass %s: no solve-kernel dispatch in %s" % (name, files[0]))
        for r in rows:
            me = (r["Kernel_Name"], r.get("Scratch_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                  r.get("Grid_Size"), r.get("Workgroup_Size"))
            if ident is None:
                ident = me
            if me != ident:
                sys.exit("mixed builds in one session: %s vs %s (pass %s)" % (ident, me, name))
        keep = [k for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                            "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
                            "Start_Timestamp", "End_Timestamp") if k in rows[0]]
        with open(os.path.join(dst, "%s_%s.csv" % (tag, name)), "w") as fh:
            w = csv.DictWriter(fh, fieldnames=keep)
            w.writeheader()
            for r in rows:
                w.writerow({k: r[k] for k in keep})
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in agg.items()}
    traffic = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
    path = os.path.join(ROOT, "ik_amd", "kernel_stats.json")
    stats = json.load(open(path)) if os.path.exists(path) else {}
    e = stats.setdefault(kernel, {})
    e["hbm_traffic_bytes_per_launch"] = traffic
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None
    e["stamp"] = {"device_source_sha16": stamp["device_source_sha16"], "git_head_when_folded": head, "kernel_symbol": ident[0],
                  "scratch_size": int(ident[1] or 0), "vgpr_count": int(ident[2] or 0), "accum_vgpr_count": int(ident[3] or 0),
                  "sgpr_count": int(ident[4] or 0), "lds_block_size": int(ident[5] or 0), "grid_size": int(ident[6] or 0),
                  "workgroup_size": int(ident[7] or 0), "workload": bench["config"]["name"], "session": os.path.basename(os.path.normpath(src))}
    persistent = "coop" in ident[0] or "refill" in ident[0]
    lanes_per_problem = 16 if "coop" in ident[0] else 1
    batch = int(bench["config"]["batch_per_gpu"])   # the session's own bench line says what one launch solves
    if not persistent and int(ident[6] or 0) < batch:
        sys.exit("grid %s smaller than the batch %d" % (ident[6], batch))
    e["pmc"] = {"batch": batch, "FETCH_SIZE_KiB": mean.get("FETCH_SIZE"),
                "WRITE_SIZE_KiB": mean.get("WRITE_SIZE"), "SQ_LDS_BANK_CONFLICT": mean.get("SQ_LDS_BANK_CONFLICT"),
                "SQ_LDS_IDX_ACTIVE": mean.get("SQ_LDS_IDX_ACTIVE"), "SQ_INSTS_VALU": mean.get("SQ_INSTS_VALU"),
                "SQ_ACTIVE_INST_VALU": mean.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": mean.get("SQ_WAVE_CYCLES"),
                "SQ_WAIT_ANY": mean.get("SQ_WAIT_ANY"), "GRBM_GUI_ACTIVE": mean.get("GRBM_GUI_ACTIVE")}
    if "SQ_INSTS_VALU_FMA_F64" in mean:
        # executed FP64 flops per launch, from the hardware instruction counters (wave instructions x 64 lanes, FMA = 2)
        fl = 64.0 * (2.0 * mean["SQ_INSTS_VALU_FMA_F64"] + mean.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                     mean.get("SQ_INSTS_VALU_ADD_F64", 0.0) + mean.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        if lanes_per_problem == 1:
            e["flop_per_solve_measured"] = fl / batch
            e.pop("wave_flop_slots_per_solve", None)
        else:  # phases leave lanes idle: wave instructions x 64 is not work done, so no flop figure (and no valu_roofline)
            e.pop("flop_per_solve_measured", None)
            e["wave_flop_slots_per_solve"] = fl / batch
        e["pmc"].update({k: mean[k] for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64",
                                              "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_VMEM") if k in mean})
    json.dump(stats, open(path, "w"), indent=1)
    with open(os.path.join(dst, "%s_bench.json" % tag), "w") as fh:
        json.dump(bench, fh)
    print(kernel, json.dumps(e["stamp"]), "traffic MB %.2f" % (traffic / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
