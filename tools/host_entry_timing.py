#!/usr/bin/env python3
"""ikgpu_dls_solve_batch_host on pinned host buffers, wall clock per call, by batch size and chunk size (IKGPU_HOST_CHUNK).  A row whose
slowest call is more than 5 ms above its median prints that call's phases (IKGPU_HOST_TRACE: lock / set-up / enqueue / wait)."""
import ctypes as C
import os
import sys
import tempfile
import time



def throttled_usec():
    """time this container's cgroup spent throttled by its CPU quota so far (cgroup v2 cpu.stat / v1 cpu.stat), or None"""
    for path, key, scale in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1.0), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1e-3)):
        try:
            for ln in open(path):
                w = ln.split()
                if w and w[0] == key:
                    return float(w[1]) * scale
        except OSError:
            pass
    return None


def run_delay_ms():
    """time the calling thread has spent RUNNABLE but not running (waiting for a CPU), /proc/thread-self/schedstat field 2"""
    try:
        return float(open("/proc/thread-self/schedstat").read().split()[1]) * 1e-6
    except (OSError, IndexError, ValueError):
        return None


TRACE = os.path.join(tempfile.gettempdir(), "ikgpu_host_trace_%d.txt" % os.getpid())
os.environ["IKGPU_HOST_TRACE"] = TRACE

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ik_amd  # noqa: E402
from ik_amd import capi, workload  # noqa: E402

model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, "cassie_fixed.kin.urdf"))
problem = ik_amd.InverseKinematicsProblem(model)
problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem, device=0)
L = capi.lib()
for B in (65536, 262144):
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "uniform")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    ref = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50))
    for layout in ("soa", "aos"):
        hq0 = (Q0 if layout == "soa" else Q0.t().contiguous()).cpu().pin_memory()
        htg = (T if layout == "soa" else T.permute(2, 0, 1).contiguous()).cpu().pin_memory()
        hq = torch.empty_like(hq0).pin_memory()
        hok, hit = torch.empty(B, dtype=torch.uint8).pin_memory(), torch.empty(B, dtype=torch.int32).pin_memory()
        prm = capi.DlsParams(50, 1e-2, 1.0, -1.0)
        for chunk in ("", "8192", "16384", "32768", "65536", str(B)):
            if chunk:
                os.environ["IKGPU_HOST_CHUNK"] = chunk
            else:
                os.environ.pop("IKGPU_HOST_CHUNK", None)

            def call():
                capi.check(L.ikgpu_dls_solve_batch_host(data._h, B, hq0.data_ptr(), htg.data_ptr(), C.byref(prm), hq.data_ptr(), hok.data_ptr(), hit.data_ptr(),
                                                        capi.SOA if layout == "soa" else capi.AOS))
            call(); call()
            open(TRACE, "w").close()
            thr0, rd0 = throttled_usec(), run_delay_ms()
            ts = []
            for _ in range(30):
                t = time.perf_counter()
                call()
                ts.append((time.perf_counter() - t) * 1e3)
            ts.sort()
            ms = ts[len(ts) // 2]
            same = torch.equal(hq if layout == "soa" else hq.t(), ref[0].cpu())
            print("B %d %s chunk %-7s: median %.3f ms (min %.3f, max %.3f) = %.3e solves/s%s" % (B, layout, chunk or "default", ms, ts[0], ts[-1], B / ms * 1e3, "" if same else "  DIFFERENT"), flush=True)
            if ts[-1] > ms + 5.0:
                rec = sorted((ln.split() for ln in open(TRACE) if ln.startswith("B ")), key=lambda w: -float(w[5]))[:1]
                thr1, rd1 = throttled_usec(), run_delay_ms()
                note = ("cgroup CPU throttling during this row %.1f ms" % ((thr1 - thr0) * 1e-3)) if thr0 is not None and thr1 is not None else "no cpu.stat"
                note += ("; the calling thread waited %.1f ms for a CPU" % (rd1 - rd0)) if rd0 is not None and rd1 is not None else ""
                for w in rec:
                    print("      slowest traced call: " + " ".join(w) + "   (" + note + ")", flush=True)
