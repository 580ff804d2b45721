import os, sys
import numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"
from test_gpu_generic import build
variants = {
    "pos+com(foot)": [("frame", "LeftFootFront", "universe", 0, 0, None), ("com", None, "LeftFootFront", None, 0, [1.0, 2.0, 0.5])],
    "com(foot) only": [("com", None, "LeftFootFront", None, 0, [1.0, 2.0, 0.5])],
    "com(foot) unit w": [("com", None, "LeftFootFront", None, 0, None)],
    "com(universe) w": [("com", None, "universe", None, 0, [1.0, 2.0, 0.5])],
    "pos only": [("frame", "LeftFootFront", "universe", 0, 0, None)],
    "pos(ref tarsus)": [("frame", "LeftFootFront", "lefttarsus", 0, 0, None)],
    "full(ref tarsus)": [("frame", "LeftFootFront", "lefttarsus", 2, 0, None)],
}
for label, specs in variants.items():
    ik, O, model, problem, data_s, om, ot, q0, tg = build("cassie_fixed", False, specs, 500, static=True)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    p = ik.dls_parameters(max_iterations=1, damping=1e-2, step_length=1.0)
    Qs, _, _ = ik.dls_batch(problem, Q0, T, data_s, ik.never_stop_visitor(), p)
    q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(1, 1e-2, 1.0, -1.0))
    ds = np.abs(Qs.cpu().numpy().T - q_ref)
    print("%-18s %-45s max %.3e  worst entry per column: %s" % (label, data_s.kernel, ds.max(), np.array2string(ds.max(axis=0), precision=1)))
