#!/usr/bin/env python3
"""What ikgpu_problem_create costs: built-in kernels against run-time compiled ones, with an empty code-object cache (cold: the
compiler runs), with the object on disk (a new process), and a second creation in the same process (module cache).
    python tools/creation_timing.py"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = ["cassie_leg", "arm7_tool", "shared_joints", "demo_task_set", "three_feet_frames"]


def child(row):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import torch
    import ik_amd
    from ik_amd import workload
    from test_gpu_generic import CASES

    def make():
        if row in ("cassie_leg", "arm7_tool"):
            name, frame = ("cassie_fixed", "LeftFootFront") if row == "cassie_leg" else ("arm7", "tool")
            model = ik_amd.Model.from_urdf_file(os.path.join(workload.MODELS_DIR, name + ".kin.urdf"))
            problem = ik_amd.InverseKinematicsProblem(model)
            problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
            return ik_amd, problem
        name, ff, specs, edit = CASES[row]
        xml = open(os.path.join(workload.MODELS_DIR, name + ".kin.urdf"), "rb").read()
        if edit:
            xml = edit(xml)
        model = ik_amd.Model.from_urdf_xml(xml, free_flyer=ff)
        problem = ik_amd.InverseKinematicsProblem(model, max(s[4] for s in specs))
        for i, (kind, f, r, t, p, w) in enumerate(specs):
            if kind == "com":
                task = problem.add_centre_of_mass_task(ik_amd.CentreOfMassTask.create(model, r), p)
            elif kind == "align":
                task = problem.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(model, f, ik_amd.AlignAxisType(t), r), p)
            else:
                task = problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t), r), p)
            if w is not None:
                task.weighting()[:] = w
        return ik_amd, problem

    torch.cuda.init()
    torch.zeros(1, device="cuda")
    out = []
    for _ in range(2):
        ik, problem = make()
        t = time.perf_counter()
        data = ik.dls_data(problem, device=0)
        out.append((time.perf_counter() - t) * 1e3)
    print("%s|%.2f|%.2f" % (data.kernel, out[0], out[1]))


if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2])
    sys.exit(0)

for row in ROWS:
    cache = tempfile.mkdtemp(prefix="ikgpu_cache_")
    env = dict(os.environ, IKGPU_CACHE_DIR=cache, IKGPU_DLS_KERNEL="generic" if row not in ("cassie_leg", "arm7_tool") else "")
    if not env["IKGPU_DLS_KERNEL"]:
        env.pop("IKGPU_DLS_KERNEL")
    res = []
    for _ in range(2):   # first: empty cache; second: objects on disk
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", row], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if "|" in l]
        if r.returncode != 0 or not line:
            print(row, "failed:", r.stderr[-400:])
            break
        res.append(line[-1].split("|"))
    else:
        print("%-20s %-50s first creation: compiler %9.1f ms, object on disk %7.1f ms | again in the same process %6.2f ms | objects %d" % (
            row, res[0][0], float(res[0][1]), float(res[1][1]), float(res[1][2]), len(os.listdir(cache))), flush=True)
    shutil.rmtree(cache, ignore_errors=True)
