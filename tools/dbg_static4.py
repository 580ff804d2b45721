import os, sys, json, subprocess
import numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import ik_amd, oracle as O
from ik_amd import workload
from conftest import urdf_path
from test_gpu_parity import _cpp_binary
model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
om = O.OracleModel(model.flat())
nom = workload.cassie_nominal(model.names)
q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(2), seed=11)
q0, qs = q0[1], qs[1]
fid = model.getFrameId("LeftFootFront")
tg = np.zeros((2, 12))
tg[0] = O.fk(om, qs)[1][fid]
tg[1, :9] = np.eye(3).ravel()
tg[1, 9:] = O.evaluate(om, O.make_tasks([(0, 0, 7, 0, None)]), np.zeros((1, 12)), qs)[0]
ot = O.make_tasks([(fid, 0, 2, 0, None), (0, 0, 7, 1, None)])
args = [_cpp_binary(), urdf_path("cassie"), "1", "30", "0.01", "1.0", "1e-8", "1", "LeftFootFront", "2", "0"]
args += ["%.17g" % x for x in tg[0]] + ["%.17g" % x for x in q0] + ["com", "universe", "1"] + ["%.17g" % x for x in tg[1, 9:]]
q1, ok1, it1 = O.dls(om, ot, tg, q0, O.params(30, 0.01, 1.0, 1e-8))
q1x, _, _ = O.dls_batch(om, ot, tg[None], q0[None], O.params(30, 0.01, 1.0, 1e-8), 1, ext="q")
for env in ({}, {"IKGPU_GENERIC_STATIC": "0"}):
    out = json.loads(subprocess.check_output(args, text=True, env=dict(os.environ, **env)))
    d = np.abs(np.array(out["q_first"]) - q1)
    print(env, out["kernel"], "max |q_first - oracle| %.3e" % d.max(), "vs float128 oracle %.3e" % np.abs(np.array(out["q_first"]) - q1x[0]).max(), "it", out["iterations"], it1)
print("oracle vs its float128 self %.3e" % np.abs(q1 - q1x[0]).max())
for iters in (1, 2, 3, 5, 10, 20, 30):
    res = {}
    for env in ({}, {"IKGPU_GENERIC_STATIC": "0"}):
        a2 = list(args); a2[3] = str(iters)
        out = json.loads(subprocess.check_output(a2, text=True, env=dict(os.environ, **env)))
        res[len(env)] = np.array(out["q_first"])
    qq, _, _ = O.dls(om, ot, tg, q0, O.params(iters, 0.01, 1.0, 1e-8))
    print(iters, "static vs oracle %.3e coop vs oracle %.3e static vs coop %.3e" % (np.abs(res[0] - qq).max(), np.abs(res[1] - qq).max(), np.abs(res[0] - res[1]).max()))
print("---- the same problem through the Python mirror, host entry, B = 1")
import torch
problem = ik_amd.InverseKinematicsProblem(model, 1)
problem.add_frame_task("t0", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full), 0)
com = problem.add_centre_of_mass_task(ik_amd.CentreOfMassTask.create(model, "universe"), 1)
for env in ({}, {"IKGPU_GENERIC_STATIC": "0"}):
    for k, v_ in env.items(): os.environ[k] = v_
    data = ik_amd.dls_data(problem, device=0)
    for k in env: os.environ.pop(k)
    for iters in (3, 5, 30):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=0.01, step_length=1.0)
        Q, ok, it = ik_amd.dls_batch(problem, np.ascontiguousarray(q0[:, None]), np.ascontiguousarray(tg[:, :, None]), data, ik_amd.inverse_kinematics_visitor(1e-8), p)
        qq, okq, itq = O.dls(om, ot, tg, q0, O.params(iters, 0.01, 1.0, 1e-8))
        print(data.kernel, iters, "it", it.tolist(), "oracle it", itq, "max |dq| %.3e" % np.abs(Q[:, 0] - qq).max())
print("---- which entries move")
data = ik_amd.dls_data(problem, device=0)
res = {}
for iters in (3, 4, 5):
    p = ik_amd.dls_parameters(max_iterations=iters, damping=0.01, step_length=1.0)
    Q, ok, it = ik_amd.dls_batch(problem, np.ascontiguousarray(q0[:, None]), np.ascontiguousarray(tg[:, :, None]), data, ik_amd.inverse_kinematics_visitor(1e-8), p)
    res[iters] = Q[:, 0]
    print(iters, ok, it)
np.set_printoptions(precision=3, linewidth=200)
print("q(5) - q(3):", res[5] - res[3])
print("q(4) - q(3):", res[4] - res[3])
Qn, _, itn = ik_amd.dls_batch(problem, np.ascontiguousarray(q0[:, None]), np.ascontiguousarray(tg[:, :, None]), data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=4, damping=0.01, step_length=1.0))
print("never-stop 4 iterations - q(3):", Qn[:, 0] - res[3])
