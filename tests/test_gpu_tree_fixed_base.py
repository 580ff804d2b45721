"""Fixed-base models on the tree kernel (device/tree_solver.hpp, TreeParams::fixed_base): two disjoint chain tasks, or one chain
task with alignment / posture rows.  The kernel solves its arrow system and drops the base block (dq_base = 0, base pose = the
world), which leaves each chain its own 7x7 system -- the dq of the reference's dense dual solve.  Through the C ABI against the
CPU oracle, 1e-6 rad, and against the generic kernel on the same problem."""
import os

import numpy as np
import pytest

from test_gpu_generic import build

pytestmark = pytest.mark.gpu
TOL = 1e-6

CASES = {
    "two_feet": ("cassie_fixed", [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None)],
                 "dls_tree<NJ=7,chains=2,fixed_base>"),
    "two_feet_types_weights_priorities": ("cassie_fixed", [("frame", "LeftFootFront", "universe", 2, 0, None),
                                                           ("frame", "RightFootBack", "universe", 0, 1, [2.0, 1.0, 0.5])],
                                          "dls_tree<NJ=7,chains=2,fixed_base>"),
    "leg_with_alignment_and_posture": ("cassie_fixed", [("frame", "LeftFootFront", "universe", 0, 0, None),
                                                        ("align", "LeftFootFront", "universe", 2, 0, None),
                                                        ("posture", 16, None, None, 1, ([0.1] * 16, [1.0] * 16))],
                                       "dls_tree<NJ=7,chains=1,align_axis,posture,fixed_base>"),
    "arm_position_with_posture": ("ur5", [("posture", 4, None, None, 0, ([1.0] * 4, [1.0] * 4)), ("frame", "tool0", "universe", 0, 0, None)],
                                  "dls_tree<NJ=6,chains=1,posture,fixed_base>"),
}


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("case", sorted(CASES))
def test_fixed_base_tree_kernel_matches_oracle(torch_cuda, case, monkeypatch):
    torch = torch_cuda
    monkeypatch.delenv("IKGPU_DLS_KERNEL", raising=False)
    name, specs, kernel = CASES[case]
    B = 600
    ik_amd, O, model, problem, data, om, ot, q0, tg = build(name, False, specs, B, seed=17)
    assert data.kernel == kernel
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    chaotic = any(s[0] == "align" for s in specs)
    for iters, damping, step, tol in ((0, 1e-2, 1.0, 1e-4), (1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4), (40, 1e-2, 1.0, 1e-6)):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
        d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
        if chaotic and step == 1.0 and iters > 3:
            assert (d <= TOL).mean() > 0.7, (case, iters, (d <= TOL).mean())
            continue
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (case, iters)
        assert d.max() <= TOL, (case, iters, d.max())
    Qa, oka, ita = ik_amd.dls_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data,
                                    ik_amd.inverse_kinematics_visitor(tol), p, layout="aos")
    assert torch.equal(Q.T.contiguous(), Qa) and torch.equal(ok, oka) and torch.equal(it, ita)
    # the generic kernel on the same problem
    p = ik_amd.dls_parameters(max_iterations=200, damping=1e-1, step_length=1e-1)
    Qt, okt, itt = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(1e-4), p)
    monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")
    gen = ik_amd.dls_data(problem, device=0)
    assert gen.kernel.startswith("dls_generic<")
    Qg, okg, itg = ik_amd.dls_batch(problem, Q0, T, gen, ik_amd.inverse_kinematics_visitor(1e-4), p)
    assert torch.equal(okt, okg) and torch.equal(itt, itg) and (Qt - Qg).abs().max().item() < 1e-8
