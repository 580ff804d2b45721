"""Where do the GPU and the CPU oracle differ by more than the bar, and is the oracle itself well determined there?
For every problem of a bench workload (B = 65536, 50 fixed iterations): |q_gpu - q_oracle|, the oracle's own sensitivity to
q0 + 1e-13, whether the result sits on a joint limit, and for the worst lanes the independent numpy twin's answer.
    python tools/parity_probe.py [cassie_leg|ur5_clamp|ur10_clamp|cassie_full_body] [B]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import ik_amd  # noqa: E402
import bench  # noqa: E402
import oracle as O  # noqa: E402
from ik_amd import workload  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cassie_leg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
w = bench.WORKLOADS[name]
model, xml = bench.load_model(ik_amd, workload, w)
problem = ik_amd.InverseKinematicsProblem(model)
for i, f in enumerate(w["frames"]):
    problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
data = ik_amd.dls_data(problem, device=0)
q0, qs = bench.make_inputs(name, model, np.arange(B))
Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
T = ik_amd.task_frames_fk_batch(problem, QS, data)
Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50))
q_gpu = Q.cpu().numpy().T
tg = T.permute(2, 0, 1).contiguous().cpu().numpy()
om = O.OracleModel(model.flat())
tasks = O.make_tasks([(model.getFrameId(f), 0, 2, 0, None) for f in w["frames"]])
prm = O.params(50, 1e-2, 1.0, -1.0)
cores = os.cpu_count() or 1
q_ref, _, _ = O.dls_batch(om, tasks, tg, q0, prm, cores)
sens = np.zeros(B)
for dq, dt in ((1e-13, 0.0), (0.0, 1e-13), (-1e-13, -1e-13)):
    tgp = tg.copy()
    tgp[:, :, 9:] += dt
    q_pert, _, _ = O.dls_batch(om, tasks, tgp, q0 + dq, prm, cores)
    sens = np.maximum(sens, np.abs(q_pert - q_ref).max(axis=1))
d = np.abs(q_gpu - q_ref).max(axis=1)
lo, hi = model.lowerPositionLimit, model.upperPositionLimit
on_limit = ((np.abs(q_ref - lo) < 1e-12) | (np.abs(q_ref - hi) < 1e-12)).any(axis=1)
err = np.array([np.abs(O.evaluate(om, tasks, tg[b], q_ref[b])[0]).max() for b in range(min(B, 65536))])
conv = err < 1e-8
print("workload %s kernel %s B %d" % (name, data.kernel, B))
print("max d all %.3e | converged(|e|<1e-8) %d: max d %.3e | not converged %d: max d %.3e" %
      (d.max(), conv.sum(), d[conv].max() if conv.any() else 0, (~conv).sum(), d[~conv].max() if (~conv).any() else 0))
for thr in (1e-9, 1e-8, 1e-7, 1e-6):
    st = sens <= thr
    print("stable(sens<=%.0e): %d lanes, max d %.3e ; unstable %d, max d %.3e" % (thr, st.sum(), d[st].max() if st.any() else 0, (~st).sum(), d[~st].max() if (~st).any() else 0))
print("lanes with d > 1e-6: %d ; of them on a limit %d ; with sens > 1e-7: %d" % ((d > 1e-6).sum(), (on_limit & (d > 1e-6)).sum(), ((sens > 1e-7) & (d > 1e-6)).sum()))
print("ratio d/sens on lanes with d>1e-9: median %.2f max %.2f" % (np.median((d / np.maximum(sens, 1e-300))[d > 1e-9]) if (d > 1e-9).any() else 0, (d / np.maximum(sens, 1e-300))[d > 1e-9].max() if (d > 1e-9).any() else 0))
worst = np.argsort(-d)[:8]
rows = []
try:
    import twin as TW
    tm = TW.load_urdf(open(os.path.join(workload.MODELS_DIR, w["urdf"] + ".kin.urdf")).read(), free_flyer=w["free_flyer"]) if hasattr(TW, "load_urdf") else None
except Exception as e:  # the twin's loader API differs: skip
    tm = None
    print("twin not used:", e)
for b in worst:
    rows.append(dict(lane=int(b), d=float(d[b]), sens=float(sens[b]), on_limit=bool(on_limit[b]), err=float(err[b]) if b < err.size else None))
    print(rows[-1])
json.dump(dict(workload=name, B=B, d=d.tolist()[:0], worst=rows), open(os.path.join(ROOT, "gpurun_out", "parity_probe_%s.json" % name), "w"))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_probe_%s.npz" % name), d=d, sens=sens, on_limit=on_limit, err=err)
