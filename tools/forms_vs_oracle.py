"""Both forms of the generic DLS kernel against the oracle on cases of tests/test_gpu_generic.py, at the parameter sets of its
parity test: tells a numerically chaotic case (both forms part from the oracle alike) from a kernel fault.
    python tools/forms_vs_oracle.py <case> ..."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"tests"), os.path.join(ROOT,"oracle")]
import torch
from test_gpu_generic import CASES, build
for case in sys.argv[1:]:
    name, ff, specs, edit = CASES[case]
    for form in ("", "lane"):
        os.environ["IKGPU_GENERIC_KERNEL"]=form
        ik_amd, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, 500, xml_edit=edit)
        Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda(); T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
        for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4), (40, 1e-2, 1.0, 1e-6)):
            p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
            Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
            q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
            d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
            print(case, form or "coop", data.kernel, iters, step, "max %.3g frac<=1e-6 %.3f" % (d.max(), (d<=1e-6).mean()), "flags", np.array_equal(ok.cpu().numpy(), ok_ref))
