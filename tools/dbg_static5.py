import os, sys
import numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"
from test_gpu_generic import build
specs = [("frame", "LeftFootFront", "universe", 2, 0, None), ("com", None, "universe", None, 1, None)]
ik, O, model, problem, data_s, om, ot, q0, tg = build("cassie", True, specs, 256, seed=11, static=True)
bc = build("cassie", True, specs, 256, seed=11, static=False)
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
for tol in (1e-8, 1e-4):
    p = ik.dls_parameters(max_iterations=30, damping=0.01, step_length=1.0)
    v = ik.inverse_kinematics_visitor(tol)
    Qs, oks, its = ik.dls_batch(problem, Q0, T, data_s, v, p)
    Qc, okc, itc = ik.dls_batch(bc[3], Q0, T, bc[4], v, p)
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(30, 0.01, 1.0, tol))
    print(tol, data_s.kernel, bc[4].kernel)
    print(" iters static", its[:16].tolist()); print(" iters coop  ", itc[:16].tolist()); print(" iters oracle", it_ref[:16].tolist())
    print(" equal static/oracle %.3f coop/oracle %.3f" % ((its.cpu().numpy() == it_ref).mean(), (itc.cpu().numpy() == it_ref).mean()))
print("---- host entry, B = 1 and B = 3")
for Bh in (1, 3, 64, 65):
    p = ik.dls_parameters(max_iterations=30, damping=0.01, step_length=1.0)
    v = ik.inverse_kinematics_visitor(1e-8)
    qh, th = np.ascontiguousarray(q0[:Bh].T), np.ascontiguousarray(tg[:Bh].transpose(1, 2, 0))
    Qs, oks, its = ik.dls_batch(problem, qh, th, data_s, v, p)
    Qc, okc, itc = ik.dls_batch(bc[3], qh, th, bc[4], v, p)
    Qd, okd, itd = ik.dls_batch(problem, torch.from_numpy(qh).cuda(), torch.from_numpy(th).cuda(), data_s, v, p)
    print(Bh, "host static iters", its.tolist()[:4], "host coop", itc.tolist()[:4], "device static", itd[:4].tolist(), "max |dq| host static vs coop %.3e, device static vs coop %.3e" % (np.abs(Qs - Qc).max(), np.abs(Qd.cpu().numpy() - Qc).max()))
