#!/usr/bin/env python3
"""A derived visitor (step tolerance) on a chain problem: the per-lane interpreter of the generic lane program against its static
build (compiled at the first such solve), B = 65536.   python tools/visitor_timing.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

B = 65536
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "uniform")
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
res = {}
for mode in ("interpreter", "static"):
    if mode == "interpreter":
        os.environ["IKGPU_GENERIC_STATIC"] = "0"
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    vis = ik_amd.inverse_kinematics_visitor(1e-4, step_tolerance=1e-10)
    prm = ik_amd.dls_parameters(max_iterations=50)
    out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)   # (the first solve compiles)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
    b.record()
    torch.cuda.synchronize()
    os.environ.pop("IKGPU_GENERIC_STATIC", None)
    res[mode] = [x.clone() for x in out]
    print("%-12s %s: %.3f ms per launch, mean iterations %.2f" % (mode, data.kernel, a.elapsed_time(b) / 5, out[2].double().mean().item()))
print("max |dq| between the two: %.2e, iteration counts equal: %s" % ((res["static"][0] - res["interpreter"][0]).abs().max().item(),
                                                                     torch.equal(res["static"][2], res["interpreter"][2])))
