"""Kernel time of a fixed-base two-chain problem (cassie_fixed, both feet, SE(3) tasks), B = 65536, 50 iterations: the tree
kernel with its base block dropped against the generic kernel (IKGPU_DLS_KERNEL=generic).
    python tools/fixed_base_timing.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
for f in ("LeftFootFront", "RightFootFront"):
    problem.add_frame_task(f, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
B = 65536
q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "near")
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
p = ik_amd.dls_parameters(max_iterations=50)
v = ik_amd.never_stop_visitor()
res = {}
for mode in ("tree", "generic"):
    if mode == "generic":
        os.environ["IKGPU_DLS_KERNEL"] = "generic"
    data = ik_amd.dls_data(problem, device=0)
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    for _ in range(2):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    e1.record()
    torch.cuda.synchronize()
    res[mode] = Q
    print("%s: %s  %.3f ms per launch, %.3g solves/s, max |q - q*| = %.2e" % (mode, data.kernel, e0.elapsed_time(e1) / 5, B / (e0.elapsed_time(e1) / 5e3),
                                                                          (Q - QS).abs().max().item()))
print("tree vs generic: max |dq| = %.2e" % (res["tree"] - res["generic"]).abs().max().item())
