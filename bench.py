#!/usr/bin/env python3
"""bench.py -- IK solves/s of the batched DLS path on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path over one batch already resident in HBM: B independent
fixed-iteration DLS solves (reference ik::dls, ik/ik/dls.cpp:5-78; lambda = 1e-2, step = 1.0,
never-stop visitor) plus, for N > 1, the RCCL all-gather of the solved configurations.  Weak scaling:
B per GPU is fixed.  Default workload = the one the metric is quoted on: Cassie single-leg chain,
B = 65536, 50 iterations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--iters I]
                    [--workload cassie_leg|cassie_full_body|ur5|ur10|cassie_demo|cassie_demo_posture|cassie_demo_pik] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import ik_amd  # noqa: E402
from ik_amd import distributed as ikdist  # noqa: E402
from ik_amd import workload  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_VALU_PEAK_TF = 78.6  # 256 CU x 4 SIMD x 16 FP64 FMA lanes x 2 flop x 2.4 GHz

# Algorithmic HBM bytes per solve: q0 in + targets in + q out + success + iters
#   = 8 nq + 96 T + 8 nq + 1 + 4
WORKLOADS = {
    "cassie_leg": dict(urdf="cassie_fixed", free_flyer=False, frames=["LeftFootFront"], nq=16,
                       text="Cassie single-leg chain (cassie_fixed.urdf, 7 support joints of nq=16), one SE(3) LeftFootFront task"),
    "cassie_full_body": dict(urdf="cassie", free_flyer=True, frames=["LeftFootFront", "RightFootFront", "pelvis"], nq=23,
                             text="Cassie full body (cassie.urdf + free-flyer, nq=23 / nv=22), SE(3) tasks on LeftFootFront, "
                                  "RightFootFront and pelvis (M=18)"),
    "ur5": dict(urdf="ur5", free_flyer=False, frames=["tool0"], nq=6,
                text="UR5 arm (ur5.urdf, nq=6), one SE(3) tool0 task, joint-limit projection after every step"),
    # BASELINE.json's config 5 names a UR10; the reference ships a UR5 only, this model is authored from the public
    # ur_description constants (fixtures/make_ur10_urdf.py) and is NOT a reference file
    "ur10": dict(urdf="ur10", free_flyer=False, frames=["tool0"], nq=6,
                 text="UR10 arm (fixtures/models/ur10.kin.urdf, authored from public constants, not in the reference; nq=6), one SE(3) "
                      "tool0 task, joint-limit projection after every step"),
    # the demo's own task set (reference ik_ros/src/cassie.cpp:45-81): the tree kernel's general build
    "cassie_demo": dict(urdf="cassie", free_flyer=True, frames=["LeftFootFront", "pelvis", "LeftFootFront"], nq=23,
                        tasks=[("frame", "LeftFootFront", 0, "pelvis"), ("frame", "pelvis", 2, "universe"),
                               ("align", "LeftFootFront", 1, "universe")],
                        text="Cassie demo task set (cassie.urdf + free-flyer): LeftFootFront position w.r.t. the pelvis, pelvis "
                             "SE(3) pose, LeftFootFront Y-axis alignment (M=10)"),
    # ... with the posture regulariser the demo declares and leaves commented out (cassie.cpp:63-64,76: all 16 joints, priority 1):
    # the tree kernel's posture build
    "cassie_demo_posture": dict(urdf="cassie", free_flyer=True, frames=["LeftFootFront", "pelvis", "LeftFootFront"], nq=23,
                                tasks=[("frame", "LeftFootFront", 0, "pelvis"), ("frame", "pelvis", 2, "universe"),
                                       ("align", "LeftFootFront", 1, "universe")],
                                posture=dict(nj=16, priority=1, weight=0.05),
                                text="Cassie demo task set (foot position w.r.t. the pelvis, pelvis SE(3) pose, foot Y-axis alignment) + a "
                                     "PostureTask on all 16 joints at priority 1, weight 0.05 (M=26)"),
    # the same tasks through the reference's other solver, ik::pik (reference ik/ik/pik.cpp:31-103): the alignment row at
    # priority 1, solved in the null space of the two pose tasks; damping factor 0.1 per level
    "cassie_demo_pik": dict(urdf="cassie", free_flyer=True, frames=["LeftFootFront", "pelvis", "LeftFootFront"], nq=23,
                            tasks=[("frame", "LeftFootFront", 0, "pelvis"), ("frame", "pelvis", 2, "universe"),
                                   ("align", "LeftFootFront", 1, "universe")],
                            prios=[0, 0, 1], solver="pik", lam=[0.1, 0.1],
                            text="Cassie demo task set as two priority levels (foot position w.r.t. the pelvis + pelvis SE(3) pose; "
                                 "then foot Y-axis alignment), ik::pik with lambda 0.1 per level, generic kernel"),
}


def bytes_per_solve(w):
    posture = 8 * w["posture"]["nj"] if w.get("posture") else 0      # one target value per posture row
    return 8 * w["nq"] + 96 * len(w["frames"]) + posture + 8 * w["nq"] + 1 + 4


def load_kernel_stats():
    path = os.path.join(ROOT, "ik_amd", "kernel_stats.json")
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


def make_inputs(name, model, idx):
    w = WORKLOADS[name]
    lo, hi = model.lowerPositionLimit, model.upperPositionLimit
    if name in ("cassie_full_body", "cassie_demo", "cassie_demo_pik", "cassie_demo_posture"):
        return workload.freeflyer_workload(lo, hi, workload.cassie_nominal(model.names), idx, seed=0, mode="near")
    if name in ("ur5", "ur10"):
        return workload.chain_workload(lo, hi, workload.UR5_NOMINAL, idx, seed=0, mode="near")
    return workload.chain_workload(lo, hi, workload.cassie_nominal(model.names), idx, seed=0, mode="uniform")


def task_specs(w):
    return w.get("tasks") or [("frame", f, 2, "universe") for f in w["frames"]]


def cpu_baseline(model, w, q0_np, tg_np, iters, budget_s=12.0):
    """The CPU oracle (oracle/ik_oracle.c, a port -- the reference itself cannot be built here) timed on
    this host's cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    om = O.OracleModel(model.flat())
    specs = task_specs(w)
    fids = [model.getFrameId(f) for _, f, _, _ in specs]
    prios = w.get("prios") or [0] * len(specs)
    rows = [(model.getFrameId(f), model.getFrameId(r), (3 + t) if kind == "align" else t, p, None) for (kind, f, t, r), p in zip(specs, prios)]
    if w.get("posture"):   # one row per joint: (tangent column, index in q, IKGPU_POSTURE_ROW, priority, [weight, mask])
        po = w["posture"]
        rows += [(model.nv - po["nj"] + k, model.nq - po["nj"] + k, 6, po["priority"], [po["weight"], 1.0]) for k in range(po["nj"])]
    tasks = O.make_tasks(rows)
    if w.get("solver") == "pik":
        prm = O.pik_params(iters, 1.0, -1.0, w["lam"])
        solve = O.pik_batch
    else:
        prm = O.params(iters, 1e-2, 1.0, -1.0)
        solve = O.dls_batch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    probe = min(256, q0_np.shape[0])
    t = time.perf_counter()
    solve(om, tasks, tg_np[:probe], q0_np[:probe], prm, 1)
    r1 = probe / (time.perf_counter() - t)
    sample = int(min(q0_np.shape[0], max(probe, r1 * cores * budget_s)))
    t = time.perf_counter()
    q_ref, ok_ref, it_ref = solve(om, tasks, tg_np[:sample], q0_np[:sample], prm, cores)
    dt = time.perf_counter() - t
    # which of the sampled problems converged on the CPU (SURVEY.md 8d parity bar): the stacked error vanishes
    if w.get("tasks"):
        conv = np.array([np.abs(O.evaluate(om, tasks, tg_np[b], q_ref[b])[0]).max() < 1e-8 for b in range(min(sample, 2048))])
        conv = np.concatenate([conv, np.zeros(sample - conv.size, dtype=bool)])
    else:
        reached = O.fk_batch(om, q_ref, fids)
        conv = np.abs(reached - tg_np[:sample]).reshape(sample, -1).max(axis=1) < 1e-8
    return dict(value=sample / dt, unit="solves/s", cores=cores, kind="port",
                sample="first %d problems of the batch, %d threads, %.2f s wall; 1-thread probe %.0f solves/s"
                       % (sample, cores, dt, r1)), q_ref, sample, conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--workload", default="cassie_leg", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--gather", action="store_true", help="run the all-gather step even at N = 1 (rehearsal)")
    ap.add_argument("--timed-only", action="store_true",
                    help="launch nothing but the warm-up and timed steps (profiling passes: every dispatch is the same launch)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch (WORLD_SIZE=%d)" % (args.gpus, world))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under torch.distributed.run (RANK / WORLD_SIZE set) the process group is always created, also for N = 1,
    # so that a one-GPU rehearsal (`python -m torch.distributed.run --nproc-per-node 1 bench.py --gather`) runs the
    # very same RCCL code path the N > 1 launch takes.
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    distributed = world > 1 or (launched and args.gather)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    w = WORKLOADS[args.workload]
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, w["urdf"] + ".kin.urdf"), free_flyer=w["free_flyer"])
    prios = w.get("prios") or [0] * len(task_specs(w))
    problem = ik_amd.InverseKinematicsProblem(model, max(prios + ([w["posture"]["priority"]] if w.get("posture") else [])))
    for i, ((kind, f, t, r), prio) in enumerate(zip(task_specs(w), prios)):
        if kind == "align":
            problem.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(model, f, ik_amd.AlignAxisType(t), r), prio)
        else:
            problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t), r), prio)
    if w.get("posture"):
        posture = problem.add_posture_task("posture", ik_amd.PostureTask.create(model, w["posture"]["nj"]), w["posture"]["priority"])
        posture.weighting()[:] = w["posture"]["weight"]
    use_pik = w.get("solver") == "pik"
    if use_pik:
        data = ik_amd.pik_data(problem, device=local_rank)
        data.lambda_ = list(w["lam"])
    else:
        data = ik_amd.dls_data(problem, device=local_rank)

    # this rank's shard of the global synthetic batch (weak scaling: B problems per GPU)
    lo, hi = ikdist.shard_range(B * world, rank, world)
    q0_np, qs_np = make_inputs(args.workload, model, np.arange(lo, hi))
    Q0 = torch.from_numpy(np.ascontiguousarray(q0_np.T)).to(dev)
    QS = torch.from_numpy(np.ascontiguousarray(qs_np.T)).to(dev)
    targets = ik_amd.task_frames_fk_batch(problem, QS, data)  # FK(q*) on the device: reachable targets
    if w.get("tasks"):
        # express each frame target in its reference frame (here: the pelvis, task 1's frame); the alignment row asks for
        # the direction the foot's Y axis has at q* (slot 2 holds the same foot frame): reachable, like the frame targets
        Rp, pp = targets[1, :9].reshape(3, 3, B), targets[1, 9:]
        Rf, pf = targets[0, :9].reshape(3, 3, B).clone(), targets[0, 9:].clone()
        targets[0, :9] = torch.einsum("kib,kjb->ijb", Rp, Rf).reshape(9, B)
        targets[0, 9:] = torch.einsum("kib,kb->ib", Rp, pf - pp)
        targets[2, 9:] = Rf[:, 1, :]
        targets[2, :9] = torch.eye(3, dtype=torch.float64, device=dev).reshape(9, 1)
    if w.get("posture"):   # posture target = the generating configuration (double 9 of each row's slot): consistent with the poses
        nj = w["posture"]["nj"]
        targets[-nj:] = 0.0
        targets[-nj:, 9, :] = QS[model.nq - nj:]
    # two buffer sets alternate so that the all-gather of step k overlaps the solve of step k + 1
    bufs = [ikdist.ShardBuffers(model.nq, B, world, dev) for _ in range(2)]
    out = bufs[0].out()
    visitor = ik_amd.never_stop_visitor()
    if use_pik:
        prm = ik_amd.pik_parameters(max_iterations=args.iters, step_length=1.0)
        solve_batch = ik_amd.pik_batch
    else:
        prm = ik_amd.dls_parameters(max_iterations=args.iters, damping=1e-2, step_length=1.0)
        solve_batch = ik_amd.dls_batch

    state = {"k": 0}

    def step():
        buf = bufs[state["k"] % 2]
        state["k"] += 1
        if distributed:
            buf.wait()  # the gather issued two steps ago must have drained this buffer set
        solve_batch(problem, Q0, targets, data, visitor, prm, out=buf.out())
        if distributed:
            buf.all_gather(async_op=True)

    def drain():
        for buf in bufs:
            buf.wait()

    for _ in range(args.warmup):
        step()
    drain()
    # HIP events on the stream the kernels are launched on (torch's current stream is the one handed to the C ABI).
    # N = 1: one pair around the K back-to-back launches -> average launch duration incl. the ~1-2 us launch boundary
    # (agrees with rocprofv3 --kernel-trace within 1 %); N > 1: the collective shares the region, so each rank also
    # times ONE isolated launch after the timed region.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step()
    ev1.record()
    drain()
    out = bufs[(state["k"] - 1) % 2].out()
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if launched:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if distributed:
        ev0.record()
        solve_batch(problem, Q0, targets, data, visitor, prm, out=bufs[0].out())
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = float(ev0.elapsed_time(ev1))
    else:
        kernel_ms = float(ev0.elapsed_time(ev1)) / args.steps

    if rank == 0:
        value = B * world * args.steps / elapsed
        bps = bytes_per_solve(w)
        achieved = bps * B / (kernel_ms * 1e-3) / 1e9
        stats = load_kernel_stats().get(data.kernel, {})
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, gfx950 correction applied: tools/pmc_to_stats.py); measured at B = 65536, linear in B.
        traffic = None
        if stats.get("hbm_traffic_bytes_per_launch") and stats.get("pmc", {}).get("batch"):
            traffic = stats["hbm_traffic_bytes_per_launch"] * B / stats["pmc"]["batch"]
        res = {
            "metric": "IK solves/sec (50-iter %s) at batch=65536" % ("PIK" if use_pik else "DLS"),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("%s, %d fixed PIK iterations, step 1.0, reachable targets FK(q*)" if use_pik else
                                    "%s, %d fixed DLS iterations, damping 1e-2, step 1.0, reachable targets FK(q*)")
                                   % (w["text"], args.iters),
                       "name": args.workload, "batch_per_gpu": B, "global_batch": B * world, "iterations": args.iters,
                       "kernel": data.kernel,
                       "parallelism": "batch-sharded x%d + RCCL all-gather" % world if distributed else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_solve": bps,
                         "note": ("fused on-chip loop: the binding roof is FP64 VALU issue, see valu_roofline"
                                  if data.kernel.startswith(("dls_chain<", "dls_tree<")) else
                                  "generic kernel, workspace in LDS (16 lanes per problem): LDS-latency bound, DESIGN.md 3.3"
                                  if data.kernel.startswith("dls_generic<") else
                                  "generic per-lane kernel, workspace in HBM: bound by that traffic, DESIGN.md 3.4")},
        }
        if stats.get("flop_per_solve_measured") and args.iters == 50:
            flops = stats["flop_per_solve_measured"]
            tf = flops * B / (kernel_ms * 1e-3) / 1e12
            res["valu_roofline"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                                    "frac": tf / FP64_VALU_PEAK_TF, "flop_per_solve": flops,
                                    "counting": "executed FP64 VALU instructions per launch from the SQ_INSTS_VALU_{FMA,MUL,ADD,"
                                                "TRANS}_F64 counters (x 64 lanes, FMA = 2), 50 iterations; profiles/r01_pmc"}
        if args.timed_only:
            args.no_cpu = True
        if world == 1 and not use_pik and not args.timed_only:
            # secondary figures of SURVEY.md 8d, outside the timed region: the library's default stop rule
            # (reference ik/ik/visitor.hpp:15-21, tolerance 1e-4 on the squared priority-0 error; max_iterations 100,
            # reference ik/ik/common.hpp:59-66) and the cold path URDF text -> device handle
            stop = ik_amd.inverse_kinematics_visitor()
            prm_stop = ik_amd.dls_parameters(max_iterations=100, damping=1e-2, step_length=1.0)
            spare = bufs[state["k"] % 2].out()   # not the buffer set holding the timed region's last result
            solve_batch(problem, Q0, targets, data, stop, prm_stop, out=spare)
            ev0.record()
            for _ in range(5):
                Qs, oks, its = solve_batch(problem, Q0, targets, data, stop, prm_stop, out=spare)
            ev1.record()
            torch.cuda.synchronize()
            ms = float(ev0.elapsed_time(ev1)) / 5
            res["default_stop_rule"] = {"value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms,
                                        "stop_sq_tol": stop.tolerance, "max_iterations": 100,
                                        "mean_iterations": float(its.double().mean().item()),
                                        "success_rate": float(oks.double().mean().item())}
            t = time.perf_counter()
            m2 = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, w["urdf"] + ".kin.urdf"), free_flyer=w["free_flyer"])
            t_parse = time.perf_counter() - t
            p2 = ik_amd.InverseKinematicsProblem(m2, w["posture"]["priority"] if w.get("posture") else 0)
            if w.get("posture"):
                p2.add_posture_task("posture", ik_amd.PostureTask.create(m2, w["posture"]["nj"]), w["posture"]["priority"])
            for i, (kind, f, tt, r) in enumerate(task_specs(w)):
                if kind == "align":
                    p2.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(m2, f, ik_amd.AlignAxisType(tt), r))
                else:
                    p2.add_frame_task("t%d" % i, ik_amd.FrameTask.create(m2, f, ik_amd.KinematicType(tt), r))
            d2 = ik_amd.dls_data(p2, device=local_rank)
            res["model_load"] = {"urdf_parse_ms": t_parse * 1e3, "urdf_to_device_handle_ms": (time.perf_counter() - t) * 1e3,
                                 "kernel": d2.kernel}
            del d2, p2, m2
        if not args.no_cpu and world == 1:   # the CPU leg runs on rank 0 at N = 1 only
            tg_np = targets.permute(2, 0, 1).contiguous().cpu().numpy()
            cpu, q_ref, sample, conv = cpu_baseline(model, w, q0_np, tg_np, args.iters)
            res["cpu_baseline"] = cpu
            d = np.abs(out[0].cpu().numpy().T[:sample] - q_ref).max(axis=1)
            res["parity_vs_cpu"] = {"problems": sample, "converged_on_cpu": int(conv.sum()),
                                    "max_abs_dq_rad_converged": float(d[conv].max()) if conv.any() else None,
                                    "max_abs_dq_rad_not_converged": float(d[~conv].max()) if (~conv).any() else None,
                                    "bar_rad": 1e-6}
        print(json.dumps(res))
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
