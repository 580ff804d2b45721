import os, sys
import numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"
from test_gpu_generic import CASES, build
case = sys.argv[1] if len(sys.argv) > 1 else "com_in_foot_frame"
name, ff, specs, edit = CASES[case]
B = 500
ik, O, model, problem, data_s, om, ot, q0, tg = build(name, ff, specs, B, xml_edit=edit, static=True)
bc = build(name, ff, specs, B, xml_edit=edit, static=False)
problem_c, data_c = bc[3], bc[4]
print(data_s.kernel, data_c.kernel)
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
for iters in (1, 2):
    p = ik.dls_parameters(max_iterations=iters, damping=1e-2, step_length=1.0)
    v = ik.never_stop_visitor()
    Qs, _, _ = ik.dls_batch(problem, Q0, T, data_s, v, p)
    Qc, _, _ = ik.dls_batch(problem_c, Q0, T, data_c, v, p)
    print(data_s.kernel, data_c.kernel)
    q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, -1.0))
    ds = np.abs(Qs.cpu().numpy().T - q_ref); dc = np.abs(Qc.cpu().numpy().T - q_ref)
    print("iters", iters, "static vs oracle max %.3e, coop vs oracle max %.3e" % (ds.max(), dc.max()))
    bad = np.argwhere(ds > 1e-6)
    print(" bad entries (problem, q index):", bad[:20].tolist(), "count", len(bad))
    if len(bad):
        b = bad[0][0]
        print(" q0  ", q0[b]); print(" ref ", q_ref[b]); print(" stat", Qs.cpu().numpy().T[b]); print(" coop", Qc.cpu().numpy().T[b])
        print(" lower", model.lowerPositionLimit); print(" upper", model.upperPositionLimit)
