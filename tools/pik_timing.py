"""Kernel time of ik::pik (reference ik/ik/pik.cpp:31-103) by level split and device form, B = 65536, 50 iterations: the lane program
compiled for the problem (device/pik_solver.hpp static_pik) against the cooperative interpreter (IKGPU_PIK_STATIC=0).
    python tools/pik_timing.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import ik_amd  # noqa: E402
from test_gpu_generic import build  # noqa: E402
from test_gpu_pik import PIK_CASES  # noqa: E402

B = 65536
cases = sys.argv[1:] or ["ur5_pos_then_ori", "ur5_full_then_elbow", "fixed_two_feet", "feet_then_pelvis", "demo_two_levels"]
for case in cases:
    name, ff, specs, edit, _ = PIK_CASES[case]
    ik, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, 256, seed=21, xml_edit=edit)
    levels = problem.max_priority_level() + 1
    rep = B // 256
    Q0 = torch.from_numpy(np.ascontiguousarray(np.tile(q0, (rep, 1)).T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(np.tile(tg, (rep, 1, 1)).transpose(1, 2, 0))).cuda()
    p = ik.pik_parameters(max_iterations=50, step_length=0.5)
    v = ik.never_stop_visitor()
    res = {}
    for form in ("coop", "static") + (("tree",) if case.startswith("demo_two_levels") else ()):
        os.environ.pop("IKGPU_PIK_STATIC", None)
        os.environ["IKGPU_PIK_KERNEL"] = "generic"      # (the demo's two levels would run on the tree kernel otherwise)
        if form == "coop":
            os.environ["IKGPU_PIK_STATIC"] = "0"
        if form == "tree":
            os.environ.pop("IKGPU_PIK_KERNEL")
        data = ik.pik_data(problem, device=0)
        data.lambda_ = [0.1] * levels
        if form == "static":
            os.environ["IKGPU_PIK_KERNEL"] = "static"   # (past the tree kernel's two-level build, onto the compiled lane program)
        kernel = data.kernel
        for _ in range(2):
            Q, ok, it = ik.pik_batch(problem, Q0, T, data, v, p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            Q, ok, it = ik.pik_batch(problem, Q0, T, data, v, p)
        e1.record()
        torch.cuda.synchronize()
        res[form] = Q
        print("%-22s %-6s %-48s %9.3f ms per launch   max |dq| vs coop %.2e" % (case, form, kernel, e0.elapsed_time(e1) / 3,
                                                                               (Q - res.get("coop", Q)).abs().max().item()), flush=True)
os.environ.pop("IKGPU_PIK_KERNEL", None)
