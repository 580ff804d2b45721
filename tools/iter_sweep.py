"""Kernel time of the headline chain kernel against the iteration count and the batch size: the slope is the cost of one
iteration, the intercept the fixed cost of a launch (dispatch, loads, instruction-cache fill, stores).
    python tools/iter_sweep.py [workload-urdf frame]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

urdf, frame = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cassie_fixed", "LeftFootFront")
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, urdf + ".kin.urdf"))
nom = workload.cassie_nominal(model.names) if urdf.startswith("cassie") else workload.UR5_NOMINAL
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem, device=0)


def timed(B, iters, reps=20):
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(B), 0, "uniform")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    p = ik_amd.dls_parameters(max_iterations=iters)
    out = None
    for _ in range(3):
        out = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("kernel", data.kernel)
print("B,iters,ms,us_per_iter_incremental")
prev = None
for it in (1, 2, 5, 10, 25, 50, 100, 200):
    ms = timed(65536, it)
    inc = "" if prev is None else "%.3f" % ((ms - prev[1]) / (it - prev[0]) * 1e3)
    print("65536,%d,%.4f,%s" % (it, ms, inc))
    prev = (it, ms)
for B in (4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    print("%d,50,%.4f," % (B, timed(B, 50, reps=10)))
