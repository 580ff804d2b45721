#!/usr/bin/env python3
"""The two-phase stop-rule solve (kernels.hpp run_two_phase) by its two switch parameters -- K (iterations before a wave of the first
phase may leave) and N (it leaves with <= N lanes still iterating) -- next to the two fixed modes, on the Cassie leg or ("tree") the
full body; every result compared bit for bit with the lock-step kernel's.
    python tools/two_phase_probe.py [tree]     the table
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

TREE = "tree" in sys.argv[1:]
if TREE:
    sys.argv.remove("tree")
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie.kin.urdf"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    for i, f in enumerate(["LeftFootFront", "RightFootFront", "pelvis"]):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
else:
    model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
nominal = workload.cassie_nominal(model.names)
data = ik_amd.dls_data(problem, device=0)
print("==", data.kernel)
vis = ik_amd.inverse_kinematics_visitor()
prm = ik_amd.dls_parameters(max_iterations=100)


def inputs(mode, B):
    if TREE:
        q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), seed=0, mode=mode)
    else:
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
        # what the switch does to THIS batch, from the lock-step kernel's own iteration counts: a lane is active in iteration n
        # (1-based) while n <= iters + 1 (its visitor fires in iteration iters + 1) or, unconverged, n <= max_iterations
        ok, its = out0[1].cpu().numpy().astype(bool), out0[2].cpu().numpy().astype(np.int64)
        last = np.where(ok, its + 1, prm.max_iterations).reshape(-1, 64)
        for K, N in ((4, 16), (8, 16), (8, 32), (4, 48)):
            n = np.arange(1, prm.max_iterations + 1)[None, :, None]
            live = (last[:, None, :] > n).sum(axis=2)                     # lanes still iterating after n iterations, per wave
            leave = np.argmax((live <= N) & (n[:, :, 0] >= K), axis=1) + 1   # (the loop also ends at max_iterations: live == 0 there)
            listed = int(sum(int((last[w] > leave[w]).sum()) for w in range(last.shape[0]) if leave[w] < prm.max_iterations))
            print("          K=%d N=%d: waves leave after %d..%d iterations (mean %.1f), %d problems listed (%.1f %%), %d of them unconverged"
                  % (K, N, leave.min(), leave.max(), leave.mean(), listed, 100.0 * listed / B, int((~ok).sum())))
        for K in ("2", "4", "8"):
            row = "          two-phase, leave after K=%-2s with <= N active:" % K
            for act in ("8", "16", "32", "48"):
                env = {"IKGPU_TWO_PHASE_ITERS": K, "IKGPU_TWO_PHASE_ACTIVE": act, "IKGPU_REFILL": "2"}
                ms, out = with_env(env, lambda: timed(Q0, T))
                same = all(torch.equal(x, y) for x, y in zip(out0, out))
                row += " | N=%s %.3f%s" % (act, ms, "" if same else " DIFFERENT")
            print(row)
        for K, N in (("4", "16"), ("8", "16")):
            row = "          K=%s N=%s by resident waves per CU of the second phase:" % (K, N)
            for wpc in ("1", "2", "3", "4", "6"):
                env = {"IKGPU_TWO_PHASE_ITERS": K, "IKGPU_TWO_PHASE_ACTIVE": N, "IKGPU_REFILL": "2", "IKGPU_REFILL_WAVES_PER_CU": wpc}
                row += " | %sw %.3f" % (wpc, with_env(env, lambda: timed(Q0, T))[0])
            print(row)
        ms, out = timed(Q0, T)
        print("          default policy %.3f%s" % (ms, "" if all(torch.equal(x, y) for x, y in zip(out0, out)) else " DIFFERENT"))
