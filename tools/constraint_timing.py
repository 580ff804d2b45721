"""Kernel time of constrained DLS problems (ik::FrameConstraint, cooperative generic kernel), B = 65536, 50 iterations, with the
Cholesky-QR basis of the constraint Jacobian and (IKGPU_PIK_PROJECTOR=dense) with the rank-revealing Gram-Schmidt only.
    python tools/constraint_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import ik_amd  # noqa: E402
from test_gpu_constraints import CASES  # noqa: E402
from test_gpu_generic import build  # noqa: E402

B = 65536
for case in ("demo_right_foot_pinned", "demo_right_foot_pinned_with_posture", "demo_everything_on", "leg_with_relative_orientation", "pelvis_with_both_feet_locked"):
    name, ff, specs, cspecs = CASES[case]
    ik, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, 256, seed=21)
    for i, (f, t, r) in enumerate(cspecs):
        problem.add_frame_constraint("c%d" % i, ik.FrameConstraint.create(model, f, ik.KinematicType(t), r))
    rep = B // 256
    Q0 = torch.from_numpy(np.ascontiguousarray(np.tile(q0, (rep, 1)).T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(np.tile(tg, (rep, 1, 1)).transpose(1, 2, 0))).cuda()
    p = ik.dls_parameters(max_iterations=50, damping=1e-1, step_length=0.5)
    res = {}
    for mode in ("cholqr", "dense"):
        os.environ["IKGPU_PIK_PROJECTOR"] = "dense" if mode == "dense" else ""
        data = ik.dls_data(problem, device=0)
        Q = None
        for _ in range(2):
            Q, ok, it = ik.dls_batch(problem, Q0, T, data, ik.never_stop_visitor(), p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            Q, ok, it = ik.dls_batch(problem, Q0, T, data, ik.never_stop_visitor(), p)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = Q
        print("%-32s %-7s %s  %.2f ms per launch" % (case, mode, data.kernel, e0.elapsed_time(e1) / 3))
    print("%-32s max |dq| between the two: %.2e" % (case, (res["cholqr"] - res["dense"]).abs().max().item()))
