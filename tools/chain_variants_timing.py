"""Kernel time of chain-kernel variants other than the hot builds (B = 65536, 50 iterations): weighted Full task, Position task,
on the Cassie leg and the UR5 -- used to A/B the table placement (scalar loads vs LDS) for the builds that spill SGPRs.
    python tools/chain_variants_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

B = 65536
for urdf, frame, nominal in (("cassie_fixed", "LeftFootFront", None), ("ur5", "tool0", workload.UR5_NOMINAL)):
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, urdf + ".kin.urdf"))
    nom = workload.cassie_nominal(model.names) if nominal is None else nominal
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(B), 0, "near")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
    for label, kt, w in (("full, unit weights (hot)", ik_amd.KinematicType.Full, None), ("full, weighted", ik_amd.KinematicType.Full, [1, 2, 0.5, 1.5, 1, 3]),
                         ("position", ik_amd.KinematicType.Position, None), ("orientation", ik_amd.KinematicType.Orientation, None)):
        problem = ik_amd.InverseKinematicsProblem(model)
        t = problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, kt))
        if w is not None:
            t.weighting()[:] = w
        data = ik_amd.dls_data(problem, device=0)
        T = ik_amd.task_frames_fk_batch(problem, QS, data)
        p = ik_amd.dls_parameters(max_iterations=50)
        for _ in range(3):
            ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
        e1.record()
        torch.cuda.synchronize()
        print("%-13s %-26s %-28s %.4f ms" % (urdf, label, data.kernel, e0.elapsed_time(e1) / 20))
