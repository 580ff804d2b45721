import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import ik_amd
from ik_amd import workload
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
for build in ("default", "general"):
    if build == "general": os.environ["IKGPU_CHAIN_HOT"] = "0"
    data = ik_amd.dls_data(problem, device=0)
    os.environ.pop("IKGPU_CHAIN_HOT", None)
    for B in (64, 65536):
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "uniform")
        Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
        T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
        vis, prm = ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50)
        out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t
        print(data.kernel, "B", B, "issue per call %.1f us, total per call %.1f us" % (t_issue / 200 * 1e6, t_all / 200 * 1e6))
