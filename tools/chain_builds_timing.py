#!/usr/bin/env python3
"""Every build of the chain kernel on the same inputs: hot / hot-rtc against general, B = 65536, 50 fixed iterations.
    python tools/chain_builds_timing.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

B = 65536
for name, frame in (("cassie_fixed", "LeftFootFront"), ("ur5", "tool0"), ("arm7", "tool"), ("arm7", "l5"), ("cassie_fixed", "lefttarsus")):
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, name + ".kin.urdf"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else np.zeros(model.nq) if name == "arm7" else workload.cassie_nominal(model.names)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, "near")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    res = {}
    for build in ("default", "general"):
        if build == "general":
            os.environ["IKGPU_CHAIN_HOT"] = "0"
        data = ik_amd.dls_data(problem, device=0)
        os.environ.pop("IKGPU_CHAIN_HOT", None)
        T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
        vis, prm = ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50)
        out = None
        for _ in range(3):
            out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
        b.record()
        torch.cuda.synchronize()
        res[build] = (data.kernel, a.elapsed_time(b) / 20, out[0].clone())
    d = (res["default"][2] - res["general"][2]).abs().max().item()
    print("%-13s %-14s | %-32s %.4f ms | %-32s %.4f ms | max |dq| between builds %.2e" % (name, frame, res["default"][0], res["default"][1], res["general"][0], res["general"][1], d))
