"""ik::dls with ik::FrameConstraint entries (reference ik/ik/frame.hpp:325-449; ik/ik/dls.cpp:26-34,43-53) on the MI355X,
through the C ABI (ikgpu_problem_create_constrained + ikgpu_dls_solve_batch), against the CPU oracle.  Tolerance 1e-6 rad."""
import os

import numpy as np
import pytest

from conftest import urdf_path
from test_gpu_generic import assert_within_bar_or_oracle_unstable, build

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


CASES = {
    # the demo's commented-out intent (reference ik_ros/src/cassie.cpp:49-51,74-75): keep the right foot where it is
    "demo_right_foot_pinned": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                                ("align", "LeftFootFront", "universe", 1, 0, None)], [("RightFootFront", 0, "universe")]),
    # ... and with the demo's posture regulariser on all sixteen joints next to it (posture + constraint build of the tree kernel)
    "demo_right_foot_pinned_with_posture": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                                             ("align", "LeftFootFront", "universe", 1, 0, None),
                                                             ("posture", 16, None, None, 0, ([0.3 + 0.04 * k for k in range(16)], [1.0] * 16))],
                                            [("RightFootFront", 0, "universe")]),
    # every line of the reference demo switched on (ik_ros/src/cassie.cpp:45-81): foot, pelvis, alignment, posture regulariser, centre of
    # mass, and the right foot pinned -- the generic kernel (centre-of-mass rows), posture rows eliminated, 13 x 13 system
    "demo_everything_on": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                            ("align", "LeftFootFront", "universe", 1, 0, None),
                                            ("posture", 16, None, None, 0, ([0.3 + 0.04 * k for k in range(16)], [1.0] * 16)),
                                            ("com", None, "universe", None, 0, None)],
                           [("RightFootFront", 0, "universe")]),
    "pelvis_with_both_feet_locked": ("cassie", True, [("frame", "pelvis", "universe", 2, 0, None)],
                                     [("RightFootFront", 2, "universe"), ("LeftFootFront", 0, "RightFootFront")]),
    "arm_keeps_tool_orientation": ("ur5", False, [("frame", "tool0", "universe", 0, 0, None)], [("tool0", 1, "universe")]),
    # a shape that has a register-resident kernel without the constraint
    "leg_with_relative_orientation": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None)],
                                      [("RightFootFront", 1, "LeftFootBack")]),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_constrained_dls_matches_oracle(torch_cuda, case, monkeypatch):
    torch = torch_cuda
    monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)   # the cooperative LDS-resident form first, the per-lane form at the end
    name, ff, specs, cspecs = CASES[case]
    B = 300
    ik_amd, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=21)
    for i, (f, t, r) in enumerate(cspecs):
        problem.add_frame_constraint("c%d" % i, ik_amd.FrameConstraint.create(model, f, ik_amd.KinematicType(t), r))
    assert problem.c_size() == sum(6 if t == 2 else 3 for _, t, _ in cspecs)
    data = ik_amd.dls_data(problem, device=0)
    # one constraint with the universe as reference on the foot of the leg that carries no task (the pinned stance foot) runs on
    # the tree kernel's constraint build; every other shape on the generic kernel
    on_tree = case.startswith("demo_right_foot_pinned")
    assert ("posture" in data.kernel) == case.endswith("with_posture")
    assert data.kernel.startswith("dls_tree<NJ=7,chains=1" if on_tree else "dls_generic<") and "constraint_rows=%d" % problem.c_size() in data.kernel
    oc = O.make_tasks([(model.getFrameId(f), model.getFrameId(r), t, 0, None) for f, t, r in cspecs])
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (5, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4), (40, 1e-1, 0.5, 1e-7)):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
        q_ref, ok_ref, it_ref = O.dls_batch_constrained(om, ot, oc, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (case, iters)
        prm_o = O.params(iters, damping, step, tol)
        assert_within_bar_or_oracle_unstable(lambda t_, q_: O.dls_batch_constrained(om, ot, oc, t_, q_, prm_o, os.cpu_count() or 1), Q.cpu().numpy().T, q_ref, tg, q0, (case, iters))
    if on_tree:                                                  # same problem, same parameters, the generic kernel
        monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")
        data_g = ik_amd.dls_data(problem, device=0)
        assert data_g.kernel.startswith("dls_generic<")
        Ql, okl, itl = ik_amd.dls_batch(problem, Q0, T, data_g, ik_amd.inverse_kinematics_visitor(tol), p)
        monkeypatch.delenv("IKGPU_DLS_KERNEL")
    else:
        monkeypatch.setenv("IKGPU_GENERIC_KERNEL", "lane")      # same problem, same parameters, the per-lane program
        Ql, okl, itl = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
        monkeypatch.delenv("IKGPU_GENERIC_KERNEL")
    assert torch.equal(ok, okl) and torch.equal(it, itl) and (Q - Ql).abs().max().item() < 1e-8
    assert np.abs(Ql.cpu().numpy().T - q_ref).max() <= TOL
    # the constraint changes the answer, and holds to first order: after ONE small step the constrained coordinates of the
    # frame relative to its reference have moved by O(step^2) only
    p = ik_amd.dls_parameters(max_iterations=1, damping=1e-2, step_length=0.01)
    Q1, _, _ = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), p)
    q1 = Q1.cpu().numpy().T
    f, t, r = cspecs[0]
    fid, rid = model.getFrameId(f), model.getFrameId(r)

    def rel(q):
        oMf = O.fk(om, q)[1]
        Mr, Mf = np.eye(4), np.eye(4)
        Mr[:3, :3], Mr[:3, 3] = oMf[rid][:9].reshape(3, 3), oMf[rid][9:]
        Mf[:3, :3], Mf[:3, 3] = oMf[fid][:9].reshape(3, 3), oMf[fid][9:]
        return np.linalg.inv(Mr) @ Mf

    for b in range(0, B, 37):
        d = np.linalg.inv(rel(q0[b])) @ rel(q1[b])       # motion of the frame relative to the reference, in the frame
        moved = np.abs(q1[b] - q0[b]).max()
        lin, ang = np.abs(d[:3, 3]).max(), np.abs(d[:3, :3] - np.eye(3)).max()
        inside = np.all((q1[b] > model.lowerPositionLimit + 1e-9) & (q1[b] < model.upperPositionLimit - 1e-9))
        if inside and moved > 1e-5:
            if t in (0, 2):
                assert lin < 20 * moved * moved + 1e-12, (case, b, lin, moved)
            if t in (1, 2):
                assert ang < 20 * moved * moved + 1e-12, (case, b, ang, moved)


def test_single_problem_dls_with_a_constraint_added_later(torch_cuda):
    """A constraint added after the data object was created is picked up at the next call, as a task is."""
    import ik_amd
    import oracle as O
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    pelvis = problem.add_frame_task("pelvis", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem)
    om = O.OracleModel(model.flat())
    q0 = np.zeros(model.nq)
    q0[2], q0[6] = 1.0, 1.0
    q0[7:] = np.clip(0.1 * np.sin(np.arange(model.nq - 7)), model.lowerPositionLimit[7:], model.upperPositionLimit[7:])
    pelvis.target.translation[:] = [0.03, -0.02, 0.95]
    ot = O.make_tasks([(model.getFrameId("pelvis"), 0, 2, 0, None)])
    tg = pelvis.target.to12()[None]
    p = ik_amd.dls_parameters(max_iterations=25, damping=1e-2, step_length=0.8)
    qa = ik_amd.dls(problem, q0, data, ik_amd.inverse_kinematics_visitor(1e-10), p)
    qa_ref, oka, ita = O.dls(om, ot, tg, q0, O.params(25, 1e-2, 0.8, 1e-10))
    assert np.abs(qa - qa_ref).max() <= TOL and data.success == oka
    for name in ("LeftFootFront", "RightFootFront"):
        problem.add_frame_constraint(name, ik_amd.FrameConstraint.create(model, name, ik_amd.KinematicType.Full))
    oc = O.make_tasks([(model.getFrameId(n), 0, 2, 0, None) for n in ("LeftFootFront", "RightFootFront")])
    qb = ik_amd.dls(problem, q0, data, ik_amd.inverse_kinematics_visitor(1e-10), p)
    qb_ref, okb, itb = O.dls_constrained(om, ot, oc, tg, q0, O.params(25, 1e-2, 0.8, 1e-10))
    assert "constraint_rows=12" in data.kernel
    assert np.abs(qb - qb_ref).max() <= TOL and data.success == okb and data.iterations == itb
    assert np.abs(qb - qa).max() > 1e-3
    # the feet drift by the second-order terms of 25 steps only, far less than without the constraints, while the pelvis moves
    f0, f1, fa = O.fk(om, q0)[1], O.fk(om, qb)[1], O.fk(om, qa)[1]
    for n in ("LeftFootFront", "RightFootFront"):
        held = np.abs(f1[model.getFrameId(n)][9:] - f0[model.getFrameId(n)][9:]).max()
        free = np.abs(fa[model.getFrameId(n)][9:] - f0[model.getFrameId(n)][9:]).max()
        assert held < 2e-2 and held < 0.5 * free
    assert np.abs(f1[model.getFrameId("pelvis")][9:] - f0[model.getFrameId("pelvis")][9:]).max() > 1e-2


def test_cpp_api_program_with_a_constraint(torch_cuda):
    """ik::FrameConstraint through the C++ mirror (tests/cpp/test_dls_api.cpp's `constraint` option)."""
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from test_gpu_parity import _cpp_binary
    model = ik_amd.Model.from_urdf_file(urdf_path("ur5"))
    om = O.OracleModel(model.flat())
    rng = np.random.default_rng(31)
    q0 = np.array([0.1, -1.4, 1.5, 0.1, 1.4, 0.05]) + rng.uniform(-0.1, 0.1, 6)
    qs = q0 + rng.uniform(-0.15, 0.15, 6)
    fid = model.getFrameId("tool0")
    tg = O.fk(om, qs)[1][[fid]]
    ot = O.make_tasks([(fid, 0, 0, 0, None)])
    oc = O.make_tasks([(fid, 0, 1, 0, None)])
    args = [_cpp_binary(), urdf_path("ur5"), "0", "30", "0.01", "1.0", "1e-10", "1", "tool0", "0", "0"]
    args += ["%.17g" % x for x in tg[0]] + ["%.17g" % x for x in q0] + ["constraint", "tool0", "1", "universe"]
    out = json.loads(subprocess.check_output(args, text=True))
    assert "constraint_rows=3" in out["kernel"]
    q1, ok1, it1 = O.dls_constrained(om, ot, oc, tg, q0, O.params(30, 0.01, 1.0, 1e-10))
    q2, ok2, it2 = O.dls_constrained(om, ot, oc, tg, q1, O.params(30, 0.01, 1.0, 1e-10))
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL and np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2
