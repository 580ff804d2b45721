#!/usr/bin/env python3
"""Tail latency of ikgpu_dls_solve_batch_host (include/ikgpu.h) on pinned host buffers: p50 / p99 / max over N calls at B = 65536 and at
B = 1 (the reference's own call pattern: ONE problem per ik::dls() call, 50 times a second, ik_ros/src/cassie.cpp:112,148), with the
per-phase trace of the slowest calls (IKGPU_HOST_TRACE: lock / set-up / enqueue / wait).
    python tools/host_entry_tails.py [calls]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE = os.path.join(tempfile.gettempdir(), "ikgpu_host_trace_%d.txt" % os.getpid())
os.environ["IKGPU_HOST_TRACE"] = TRACE
import torch  # noqa: E402
import ik_amd  # noqa: E402
from ik_amd import capi, workload  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem, device=0)
L = capi.lib()
for B, iters in ((65536, 50), (4096, 50), (1, 50), (1, 8)):
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "uniform")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    hq0, htg = Q0.cpu().pin_memory(), T.cpu().pin_memory()
    hq = torch.empty_like(hq0).pin_memory()
    hok, hit = torch.empty(B, dtype=torch.uint8).pin_memory(), torch.empty(B, dtype=torch.int32).pin_memory()
    prm = capi.DlsParams(iters, 1e-2, 1.0, -1.0)

    def call():
        capi.check(L.ikgpu_dls_solve_batch_host(data._h, B, hq0.data_ptr(), htg.data_ptr(), C.byref(prm), hq.data_ptr(), hok.data_ptr(), hit.data_ptr(), capi.SOA))
    for _ in range(5):
        call()
    open(TRACE, "w").close()
    ts = []
    for _ in range(N):
        t = time.perf_counter()
        call()
        ts.append((time.perf_counter() - t) * 1e3)
    a = np.sort(np.array(ts))
    print("B %6d, %2d iterations, %d calls: p50 %.4f ms  p90 %.4f  p99 %.4f  max %.4f  (mean %.4f)" % (B, iters, N, a[N // 2], a[int(N * 0.9)], a[int(N * 0.99)], a[-1], a.mean()), flush=True)
    lines = [ln.split() for ln in open(TRACE) if ln.startswith("B ")]
    if lines:   # (the staged small-batch path writes no trace)
        rec = sorted(lines, key=lambda w: -float(w[5]))[:3]
        for w in rec:
            print("      slowest traced calls: " + " ".join(w))
os.remove(TRACE)
