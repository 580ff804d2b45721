"""Kernel time of tree-kernel variants other than the hot build (Cassie full body, B = 65536, 50 iterations): weighted tasks,
Position-type foot tasks.    python tools/tree_variants_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

B = 65536
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie.kin.urdf"), free_flyer=True)
q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), seed=0, mode="near")
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
K = ik_amd.KinematicType
for label, types, w in (("three Full tasks, unit weights (hot)", (K.Full, K.Full, K.Full), None),
                        ("three Full tasks, pelvis weighted", (K.Full, K.Full, K.Full), [1, 1, 1, 0.3, 0.3, 0.3]),
                        ("feet Position, pelvis Full", (K.Position, K.Position, K.Full), None)):
    problem = ik_amd.InverseKinematicsProblem(model)
    for f, t in zip(("LeftFootFront", "RightFootFront", "pelvis"), types):
        task = problem.add_frame_task(f, ik_amd.FrameTask.create(model, f, t))
        if w is not None and f == "pelvis":
            task.weighting()[:] = w
    data = ik_amd.dls_data(problem, device=0)
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    p = ik_amd.dls_parameters(max_iterations=50)
    for _ in range(2):
        ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
    e1.record()
    torch.cuda.synchronize()
    print("%-40s %-36s %.4f ms  max|q - q*| %.1e" % (label, data.kernel, e0.elapsed_time(e1) / 10, (Q - QS).abs().max().item()))
