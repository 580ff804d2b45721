"""Reads the stamps of the measurement build of the hot chain kernel (tools/build_variant.sh stamp -DIKGPU_HOT_STAMP; run with
IKGPU_LIB=ik_amd/libikgpu_stamp.so): per wave, the iteration loop's duration in shader clocks and in 100 MHz ticks.
    IKGPU_LIB=$PWD/ik_amd/libikgpu_stamp.so python tools/loop_stamps.py [iters] [near|uniform]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem, device=0)
B = 65536
q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, mode)
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
p = ik_amd.dls_parameters(max_iterations=iters)
out = None
for _ in range(5):
    out = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p, out=out)
torch.cuda.synchronize()
it = out[2].cpu().numpy().reshape(-1, 64)
cyc, real = it[:, 0].astype(float), it[:, 1].astype(float)
pre, epi, t_start, t_end = (it[:, k].astype(float) for k in (2, 3, 4, 5))
print("prologue (wave start -> loop) %.2f us mean / %.2f max; epilogue (loop end -> stores done) %.2f us mean / %.2f max; wave starts spread over "
      "%.2f us, ends over %.2f us; first start -> last end %.2f us"
      % (pre.mean() / 100, pre.max() / 100, epi.mean() / 100, epi.max() / 100, (t_start.max() - t_start.min()) / 100,
         (t_end.max() - t_end.min()) / 100, (t_end.max() - t_start.min()) / 100))
print("mode %s, %d iterations: loop cycles per wave mean %.0f (min %.0f max %.0f) -> %.1f cycles = %.1f quads per iteration; "
      "loop time %.2f us (100 MHz ticks) -> clock %.3f GHz; %.3f us per iteration"
      % (mode, iters, cyc.mean(), cyc.min(), cyc.max(), cyc.mean() / iters, cyc.mean() / iters / 4, real.mean() / 100.0,
         cyc.mean() / (real.mean() * 10.0), real.mean() / 100.0 / iters))

ticks = it[:, 1].astype(float) / 100.0
print("loop time per wave (us): min %.2f p1 %.2f median %.2f p99 %.2f max %.2f" % (ticks.min(), np.percentile(ticks, 1), np.median(ticks), np.percentile(ticks, 99), ticks.max()))
wg = np.arange(ticks.size)
print("by XCD (workgroup index mod 8): loop us " + " ".join("%.2f" % ticks[wg % 8 == x].mean() for x in range(8)))
print("by XCD: loop cycles " + " ".join("%.0f" % cyc[wg % 8 == x].mean() for x in range(8)))
print("by XCD: end tick - first start (us) " + " ".join("%.2f" % ((t_end[wg % 8 == x].max() - t_start.min()) / 100) for x in range(8)))
