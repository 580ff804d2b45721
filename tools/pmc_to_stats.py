#!/usr/bin/env python3
"""Fold rocprofv3 PMC passes (tools/pmc_session.sh) into ik_amd/kernel_stats.json and profiles/.

HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: FETCH_SIZE / WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE counts a coalesced streaming read at half its bytes (MI355X_MICROARCH.md, section HBM) --
calibrated here on this kernel's own known byte count: the chain kernel reads 28 doubles per problem =
14.68 MB at B = 65536 and FETCH_SIZE reports 7.31 MiB... x 2 = 14.97 MB (the rest is the constant table).

    python tools/pmc_to_stats.py gpurun_out/pmc_leg "dls_chain<NJ=7,full>" profiles/r01_pmc leg
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src, kernel, dst, tag):
    os.makedirs(dst, exist_ok=True)
    agg = collections.defaultdict(list)
    for p in sorted(glob.glob(os.path.join(src, "*", "runc", "*counter_collection.csv"))):
        rows = [r for r in csv.DictReader(open(p)) if "dls_" in r["Kernel_Name"] or "pik_" in r["Kernel_Name"]]
        if not rows:
            continue
        keep = [k for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                            "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
                            "Start_Timestamp", "End_Timestamp") if k in rows[0]]
        name = os.path.basename(os.path.dirname(os.path.dirname(p)))
        with open(os.path.join(dst, "%s_%s.csv" % (tag, name)), "w") as fh:
            w = csv.DictWriter(fh, fieldnames=keep)
            w.writeheader()
            for r in rows:
                w.writerow({k: r[k] for k in keep})
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in agg.items()}
    traffic = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
    path = os.path.join(ROOT, "ik_amd", "kernel_stats.json")
    stats = json.load(open(path))
    e = stats.setdefault(kernel, {})
    e["hbm_traffic_bytes_per_launch"] = traffic
    # problems per launch: one per lane, except the cooperative kernel's sixteen lanes per problem
    lanes_per_problem = 16 if rows and "coop" in rows[0]["Kernel_Name"] else 1
    # (the cooperative kernels run persistent workgroups: their grid says nothing about the batch -- bench.py's default it is)
    batch = int(os.environ.get("IKGPU_PMC_BATCH", "65536")) if lanes_per_problem == 16 else (int(rows[0]["Grid_Size"]) if rows else None)
    e["pmc"] = {"batch": batch, "FETCH_SIZE_KiB": mean.get("FETCH_SIZE"),
                "WRITE_SIZE_KiB": mean.get("WRITE_SIZE"), "SQ_LDS_BANK_CONFLICT": mean.get("SQ_LDS_BANK_CONFLICT"),
                "SQ_LDS_IDX_ACTIVE": mean.get("SQ_LDS_IDX_ACTIVE"), "SQ_INSTS_VALU": mean.get("SQ_INSTS_VALU"),
                "SQ_ACTIVE_INST_VALU": mean.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": mean.get("SQ_WAVE_CYCLES"),
                "SQ_WAIT_ANY": mean.get("SQ_WAIT_ANY"), "GRBM_GUI_ACTIVE": mean.get("GRBM_GUI_ACTIVE")}
    if "SQ_INSTS_VALU_FMA_F64" in mean and e["pmc"]["batch"]:
        # executed FP64 flops per launch, from the hardware instruction counters (wave instructions x 64 lanes, FMA = 2)
        fl = 64.0 * (2.0 * mean["SQ_INSTS_VALU_FMA_F64"] + mean.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                     mean.get("SQ_INSTS_VALU_ADD_F64", 0.0) + mean.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        if lanes_per_problem == 1:
            e["flop_per_solve_measured"] = fl / e["pmc"]["batch"]
        else:  # phases leave lanes idle: wave instructions x 64 is not work done, so no flop figure (and no valu_roofline)
            e.pop("flop_per_solve_measured", None)
            e["wave_flop_slots_per_solve"] = fl / e["pmc"]["batch"]
        e["pmc"].update({k: mean[k] for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64",
                                              "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_VMEM") if k in mean})
    json.dump(stats, open(path, "w"), indent=1)
    print(kernel, json.dumps(e["pmc"]), "traffic MB", traffic / 1e6)


if __name__ == "__main__":
    main(*sys.argv[1:5])
