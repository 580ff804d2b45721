"""Kernel time of the forms of the generic DLS kernel (cooperative LDS / per-lane HBM / per-lane LDS / the lane program specialised at run time) on problems that
plan onto it, B = 65536, 50 iterations.
    python tools/generic_forms.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import ik_amd  # noqa: E402
from test_gpu_generic import CASES, build  # noqa: E402

B = 65536
os.environ["IKGPU_DLS_KERNEL"] = "generic"
cases = sys.argv[1:] or ["shared_joints", "com_of_the_arm", "moving_reference_prismatic", "demo_task_set", "com_under_feet"]
for case in cases:
    name, ff, specs, edit = CASES[case]
    ik, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, 256, seed=21, xml_edit=edit)
    rep = B // 256
    Q0 = torch.from_numpy(np.ascontiguousarray(np.tile(q0, (rep, 1)).T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(np.tile(tg, (rep, 1, 1)).transpose(1, 2, 0))).cuda()
    p = ik.dls_parameters(max_iterations=50, damping=1e-1, step_length=0.5)
    res = {}
    for form in (os.environ.get("FORMS", "coop,lane,lds,static").split(",")):
        os.environ.pop("IKGPU_GENERIC_KERNEL", None)
        os.environ.pop("IKGPU_GENERIC_STATIC", None)
        if form in ("lane", "lds"):
            os.environ["IKGPU_GENERIC_KERNEL"] = form
        elif form == "coop":
            os.environ["IKGPU_GENERIC_STATIC"] = "0"
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
        res[form] = Q
        print("%-28s %-6s %s  %.3f ms per launch   max |dq| vs coop %.2e" % (case, form, data.kernel, e0.elapsed_time(e1) / 3,
                                                                          (Q - res.get("coop", Q)).abs().max().item()), flush=True)
