#!/usr/bin/env python3
"""Stop-rule mode of the chain kernels on batches larger than the machine: lock-step against lane refill, by batch size, resident
waves per CU, target distribution (uniform: ~3 % of the problems never converge; near: all converge) and max_iterations.
    python tools/refill_timing.py [model frame | full_body x | demo x]     (demo: the reference demo's task set on its static lane program)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

name, frame = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cassie_fixed", "LeftFootFront")
DEMO = name == "demo"
FULL_BODY = name == "full_body" or DEMO
if DEMO:
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie.kin.urdf"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t0", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Position, "pelvis"))
    problem.add_frame_task("t1", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
    problem.add_align_axis_task("t2", ik_amd.AlignAxisTask.create(model, "LeftFootFront", ik_amd.AlignAxisType.AxisY, "universe"))
    nominal = workload.cassie_nominal(model.names)
elif FULL_BODY:
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie.kin.urdf"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    for i, f in enumerate(["LeftFootFront", "RightFootFront", "pelvis"]):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
    nominal = workload.cassie_nominal(model.names)
else:
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, name + ".kin.urdf"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else np.zeros(model.nq) if name == "arm7" else workload.cassie_nominal(model.names)


def timed(data, Q0, T, vis, prm, reps=3):
    out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


for build in (("default",) if FULL_BODY else ("default", "general")):
    if build == "general":
        os.environ["IKGPU_CHAIN_HOT"] = "0"
    data = ik_amd.dls_data(problem, device=0)
    os.environ.pop("IKGPU_CHAIN_HOT", None)
    print("==", data.kernel)
    for mode in ("uniform", "near"):
        for B in (65536, 262144, 1048576):
            if FULL_BODY:
                q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), seed=0, mode=mode)
            else:
                q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, mode)
            Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
            T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
            if DEMO:   # targets in the tasks' reference frames, the alignment row asks for the foot's Y axis at q* (as bench.py)
                Rp, pp = T[1, :9].reshape(3, 3, B), T[1, 9:]
                Rf, pf = T[0, :9].reshape(3, 3, B).clone(), T[0, 9:].clone()
                T[0, :9] = torch.einsum("kib,kjb->ijb", Rp, Rf).reshape(9, B)
                T[0, 9:] = torch.einsum("kib,kb->ib", Rp, pf - pp)
                T[2, 9:] = Rf[:, 1, :]
                T[2, :9] = torch.eye(3, dtype=torch.float64, device="cuda").reshape(9, 1)
            ms50, _ = timed(data, Q0, T, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50))
            for max_it in (100, 20):
                prm = ik_amd.dls_parameters(max_iterations=max_it)
                vis = ik_amd.inverse_kinematics_visitor()
                os.environ["IKGPU_REFILL"] = "0"
                ms0, out0 = timed(data, Q0, T, vis, prm)
                mean_it = float(out0[2].double().mean())
                row = "%s B=%7d max_it=%3d mean_it %.2f | 50 fixed %.3f ms | lock-step %.3f ms" % (mode, B, max_it, mean_it, ms50, ms0)
                os.environ["IKGPU_REFILL"] = "1"
                for wpc in ("4", "8"):
                    os.environ["IKGPU_REFILL_WAVES_PER_CU"] = wpc
                    ms1, out1 = timed(data, Q0, T, vis, prm)
                    same = all(torch.equal(x, y) for x, y in zip(out0, out1))
                    row += " | refill %s w/CU %.3f ms%s" % (wpc, ms1, "" if same else " DIFFERENT")
                os.environ.pop("IKGPU_REFILL_WAVES_PER_CU")
                os.environ.pop("IKGPU_REFILL")
                ms2, out2 = timed(data, Q0, T, vis, prm)       # the default policy (two phases above the resident batch)
                same = all(torch.equal(x, y) for x, y in zip(out0, out2))
                row += " | default policy %.3f ms%s" % (ms2, "" if same else " DIFFERENT")
                ideal = mean_it * ms50 / 50      # (ms50 is the 50-iteration time of THIS batch: it already scales with B)
                print(row + " | ideal (mean_it x time per iteration of the batch) %.3f ms" % ideal)
