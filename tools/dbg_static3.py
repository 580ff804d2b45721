import os, sys
import numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"
import test_gpu_static as S
for name in ("demo_task_set", "demo_right_foot_pinned"):
    both = S._build(name, 4109, 12)
    ik, O, model, problem, data_s, om, ot, q0, tg = both["static"]
    data_t, problem_t = both["tree"][4], both["tree"][3]
    cons = S.ROUTED[name][3]
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, damping, step, tol in ((50, 1e-1, 0.5, -1.0), (100, 1e-1, 0.5, 1e-4), (200, 1e-1, 1e-1, 1e-4)):
        p = ik.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        v = ik.inverse_kinematics_visitor(tol)
        Qs, oks, its = ik.dls_batch(problem, Q0, T, data_s, v, p)
        Qt, okt, itt = ik.dls_batch(problem_t, Q0, T, data_t, v, p)
        prm = O.params(iters, damping, step, tol)
        if cons:
            oc = O.make_tasks([(model.getFrameId(cons[0]), 0, cons[1], 0, None)])
            q_ref, ok_ref, it_ref = O.dls_batch_constrained(om, ot, oc, tg, q0, prm, os.cpu_count() or 1)
            q_x, _, _ = O.dls_batch_constrained(om, ot, oc, tg, q0, prm, os.cpu_count() or 1, ext="q")
        else:
            q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, prm, os.cpu_count() or 1)
            q_x, _, _ = O.dls_batch(om, ot, tg, q0, prm, os.cpu_count() or 1, ext="q")
        ds = np.abs(Qs.cpu().numpy().T - q_ref).max(axis=1); dt = np.abs(Qt.cpu().numpy().T - q_ref).max(axis=1)
        dx = np.abs(q_ref - q_x).max(axis=1); dsx = np.abs(Qs.cpu().numpy().T - q_x).max(axis=1)
        print(name, (iters, damping, step, tol), "kernels", data_s.kernel, data_t.kernel)
        print("   iters equal: static %.4f tree %.4f | within 1e-6 of the oracle: static %.4f tree %.4f | oracle vs its float128 self %.4f | static vs float128 %.4f | success %.3f mean it %.1f"
              % ((its.cpu().numpy() == it_ref).mean(), (itt.cpu().numpy() == it_ref).mean(), (ds <= 1e-6).mean(), (dt <= 1e-6).mean(), (dx <= 1e-6).mean(), (dsx <= 1e-6).mean(), ok_ref.mean(), it_ref.mean()))
