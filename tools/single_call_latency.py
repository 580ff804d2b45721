import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, ik_amd
from ik_amd import workload
model = ik_amd.Model.from_urdf_file(workload.MODELS_DIR + "/cassie_fixed.kin.urdf")
problem = ik_amd.InverseKinematicsProblem(model)
t = problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem)
q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(4), 0, "near")
Q = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
T = ik_amd.task_frames_fk_batch(problem, Q, data).cpu().numpy()
t.target = ik_amd.SE3.from12(T[0, :, 0])
for it in (8, 50, 100):
    p = ik_amd.dls_parameters(max_iterations=it)
    v = ik_amd.never_stop_visitor()
    for _ in range(5): ik_amd.dls(problem, q0[0], data, v, p)
    t0 = time.perf_counter()
    for _ in range(200): q = ik_amd.dls(problem, q0[0], data, v, p)
    print("single-problem ik::dls, %d iterations: %.1f us per call" % (it, (time.perf_counter() - t0) / 200 * 1e6))
