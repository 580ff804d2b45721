#!/usr/bin/env python3
"""Per-phase cycle counts of the cooperative generic kernels (debug build: make -C ik_amd/csrc KERNEL_EXTRA=-DIKGPU_COOP_PROFILE).
Runs the demo task set once and prints workgroup 0's cycles per phase and iteration.  `python tools/coop_profile.py pik`: ik::pik with
the alignment row on a second level (the bench's cassie_demo_pik workload) instead of ik::dls."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import capi, workload  # noqa: E402

model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie.kin.urdf"), free_flyer=True)
PIK = len(sys.argv) > 1 and sys.argv[1] == "pik"
CASE = sys.argv[2] if len(sys.argv) > 2 else None      # a case of tests/test_gpu_generic.py instead of the demo task set
os.environ["IKGPU_DLS_KERNEL"] = "generic"
if CASE:
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    from test_gpu_generic import CASES, build
    name, ff, specs, edit = CASES[CASE]
    ik, O, model, problem, data, om, ot, q0c, tgc = build(name, ff, specs, 256, seed=21, xml_edit=edit)
    B, iters = 65536, 50
    rep = B // 256
    Q0 = torch.from_numpy(np.ascontiguousarray(np.tile(q0c, (rep, 1)).T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(np.tile(tgc, (rep, 1, 1)).transpose(1, 2, 0))).cuda()
    p = ik_amd.dls_parameters(max_iterations=iters, damping=1e-1, step_length=0.5)
    L = capi.lib()
    out = (C.c_longlong * 16)()
    for _ in range(2):
        ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
        torch.cuda.synchronize()
        L.ikgpu_debug_coop_profile(out, 1)
    names = ["local transforms", "chains (oMi)", "task blocks", "task Jacobian columns", "Gram", "Cholesky", "back substitution", "dq", "integrate",
             "pik: de, Jbar", "pik: Gram", "pik: Cholesky + dq", "pik: row-space basis", "joint Jacobian (Jw), CoM"]
    tot = sum(out[:14])
    print(CASE, data.kernel)
    for n, v in zip(names, out[:14]):
        if v:
            print("%-24s %9.0f cycles / iteration  %5.1f %%" % (n, v / iters, 100.0 * v / tot))
    sys.exit(0)
problem = ik_amd.InverseKinematicsProblem(model, 1 if PIK else 0)
problem.add_frame_task("fl", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Position, "pelvis"))
problem.add_frame_task("pelvis", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
problem.add_align_axis_task("align", ik_amd.AlignAxisTask.create(model, "LeftFootFront", ik_amd.AlignAxisType.AxisY), 1 if PIK else 0)
data = ik_amd.pik_data(problem) if PIK else ik_amd.dls_data(problem)
if PIK:
    data.lambda_ = [0.1, 0.1]
B, iters = 65536, 50
q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), seed=0)
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
T = torch.zeros((3, 12, B), dtype=torch.float64, device="cuda")
T[:, [0, 4, 8]] = 1.0
T[0, 11], T[1, 11], T[2, 9] = -0.8, 1.0, 1.0
p = ik_amd.pik_parameters(max_iterations=iters, step_length=1.0) if PIK else ik_amd.dls_parameters(max_iterations=iters, damping=1e-2, step_length=1.0)
solve = ik_amd.pik_batch if PIK else ik_amd.dls_batch
L = capi.lib()
out = (C.c_longlong * 16)()
solve(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
torch.cuda.synchronize()
L.ikgpu_debug_coop_profile(out, 1)
solve(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
torch.cuda.synchronize()
L.ikgpu_debug_coop_profile(out, 1)
names = ["local transforms", "chains (oMi)", "task blocks", "task Jacobian columns", "Gram", "Cholesky", "back substitution", "dq", "integrate",
         "pik: de, Jbar", "pik: Gram", "pik: Cholesky + dq", "pik: row-space basis", "joint Jacobian (Jw), CoM"]
tot = sum(out[:14])
for n, v in zip(names, out[:14]):
    print("%-24s %9.0f cycles / iteration  %5.1f %%" % (n, v / iters, 100.0 * v / tot))
print("%-24s %9.0f cycles / iteration (s_memtime, workgroup 0)" % ("total", tot / iters))
