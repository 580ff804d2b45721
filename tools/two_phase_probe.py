#!/usr/bin/env python3
"""Where a two-phase stop-rule solve (kernels.hpp run_two_phase) spends its time: the default policy on the Cassie leg by K (lock-step
iterations before the compaction) and by resident waves of the second phase, next to the two fixed modes.
    python tools/two_phase_probe.py            the table
    rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/two_phase_probe.py trace [uniform|near] [B]
                                               20 default-policy solves only: the per-kernel durations of the two launches"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import workload  # noqa: E402

model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
nominal = workload.cassie_nominal(model.names)
data = ik_amd.dls_data(problem, device=0)
vis = ik_amd.inverse_kinematics_visitor()
prm = ik_amd.dls_parameters(max_iterations=100)


def inputs(mode, B):
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, mode)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    return Q0, T


def timed(Q0, T, reps=5):
    out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if len(sys.argv) > 1 and sys.argv[1] == "trace":
    mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    Q0, T = inputs(mode, B)
    print(mode, B, "default policy %.3f ms" % timed(Q0, T, reps=20)[0])
    sys.exit(0)

for mode in ("uniform", "near"):
    for B in (262144, 1048576):
        Q0, T = inputs(mode, B)
        ms0, out0 = with_env({"IKGPU_REFILL": "0"}, lambda: timed(Q0, T))
        row = "%-7s B=%7d | lock-step %.3f" % (mode, B, ms0)
        for wpc in ("2", "4", "8"):
            ms, out = with_env({"IKGPU_REFILL": "1", "IKGPU_REFILL_WAVES_PER_CU": wpc}, lambda: timed(Q0, T))
            row += " | refill %sw %.3f" % (wpc, ms)
        print(row)
        for K in ("1", "2", "4", "8", "16"):
            row = "          two-phase K=%-2s" % K
            for wpc in (None, "2", "4", "8"):
                env = {"IKGPU_TWO_PHASE_ITERS": K, "IKGPU_REFILL": "2"}
                if wpc:
                    env["IKGPU_REFILL_WAVES_PER_CU"] = wpc
                ms, out = with_env(env, lambda: timed(Q0, T))
                same = all(torch.equal(x, y) for x, y in zip(out0, out))
                row += " | %s %.3f%s" % ("policy" if wpc is None else wpc + "w", ms, "" if same else " DIFFERENT")
            print(row)
