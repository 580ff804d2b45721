"""ik::pik (reference ik/ik/pik.cpp:31-103) on the MI355X through the C ABI (ikgpu_pik_solve_batch) against the CPU oracle
(oracle/ik_oracle.c: iko_pik).  Tolerance: 1e-6 rad on q, as BASELINE.json's north_star states for the solvers; measured
differences are ~1e-12.  Cases whose final projector hinges on a noise-level rank decision (see tests/test_lane_emulation.py,
PIK_CASES) are compared with da = 0, where the result does not depend on it."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import urdf_path
from test_gpu_generic import build

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


PIK_CASES = {
    "ur5_pos_then_ori": ("ur5", False, [("frame", "tool0", "universe", 0, 0, None), ("frame", "tool0", "universe", 1, 1, None)], None, True),
    "ur5_full_then_elbow": ("ur5", False, [("frame", "tool0", "universe", 2, 0, None), ("frame", "forearm_link", "universe", 0, 1, None)], None, False),
    "fixed_two_feet": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None),
                                               ("frame", "RightFootFront", "universe", 0, 1, [1.0, 2.0, 0.5])], None, True),
    "demo_three_levels": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                           ("align", "LeftFootFront", "universe", 1, 1, None),
                                           ("posture", 16, None, None, 2, ([0.5] * 16, [1.0] * 16))], None, False),
    "feet_then_pelvis": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                          ("frame", "pelvis", "universe", 2, 1, None)], None, False),
    "single_level_leg": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None)], None, True),
    # the bench workload cassie_demo_pik: the demo's task set over two levels -- the shape the tree kernel takes (PikRow)
    "demo_two_levels": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                         ("align", "LeftFootFront", "universe", 1, 1, None)], None, True),
    "demo_two_levels_full_foot": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 2, 0, [1.0, 2.0, 0.5, 1.0, 1.5, 3.0]),
                                                   ("align", "LeftFootFront", "universe", 2, 1, None)], None, False),   # (six foot rows + the alignment row use up the chain's seven directions: the last projector is rounding noise there)
}


def _hiprtc():
    return any(os.path.exists(p) for p in ("/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7"))


def _pik_data(ik_amd, problem, lam, da=None):
    data = ik_amd.pik_data(problem, device=0)
    data.lambda_ = list(lam)
    if da is not None:
        data.da = np.asarray(da, float)
    return data


@pytest.mark.parametrize("case", sorted(PIK_CASES))
def test_pik_kernel_matches_oracle(torch_cuda, case):
    torch = torch_cuda
    name, ff, specs, edit, projector_determined = PIK_CASES[case]
    B = 300  # not a multiple of 64
    ik_amd, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=4, xml_edit=edit)
    levels = problem.max_priority_level() + 1
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, step, tol, lam, da in ((1, 1.0, -1.0, [1.0] * levels, None),
                                      (4, 1.0, -1.0, [0.1] * levels, None),
                                      (30, 0.5, 1e-8, [0.05, 0.1, 0.2][:levels], None),
                                      (6, 1.0, -1.0, [0.1] * levels, 0.01 * np.cos(np.arange(model.nv)))):
        if da is not None and not projector_determined:
            continue
        data = _pik_data(ik_amd, problem, lam, da)
        if levels == 1 and da is None:   # ik::pik with one level and no secondary step IS the DLS iteration (include/ikgpu.h)
            assert data.kernel == ik_amd.plan(problem) and data.kernel.startswith("dls_chain<")
        elif case.startswith("demo_two_levels") and da is None:   # two levels in the tree kernel's shape (device/tree_solver.hpp PikRow)
            assert data.kernel.startswith("dls_tree<NJ=7,chains=1") and data.kernel.endswith(",pik_levels=2>"), data.kernel
        elif case == "demo_three_levels":     # 26 rows: beyond what the compiled lane program takes -- the cooperative interpreter
            assert data.kernel.startswith("pik_generic<") and not data.kernel.endswith(",static>"), data.kernel
        else:                                 # every other split: the lane program compiled for the problem (device/pik_solver.hpp static_pik)
            assert data.kernel.startswith("pik_generic<") and (data.kernel.endswith(",static>") or not _hiprtc()), data.kernel
        p = ik_amd.pik_parameters(max_iterations=iters, step_length=step)
        visitor = ik_amd.inverse_kinematics_visitor(tol)
        Q, ok, it = ik_amd.pik_batch(problem, Q0, T, data, visitor, p)
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, None if da is None else list(da)), os.cpu_count() or 1)
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (case, iters)
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL, (case, iters)
    # array-of-structures input gives the same bits; so does the host-pointer entry point
    Qa, oka, ita = ik_amd.pik_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data, visitor, p, layout="aos")
    assert torch.equal(Q.T.contiguous(), Qa) and torch.equal(ok, oka) and torch.equal(it, ita)
    Qh, okh, ith = ik_amd.pik_batch(problem, q0, tg, data, visitor, p, layout="aos")
    assert np.array_equal(Qh, Qa.cpu().numpy()) and np.array_equal(okh, oka.cpu().numpy()) and np.array_equal(ith, ita.cpu().numpy())


@pytest.mark.parametrize("case", ["ur5_pos_then_ori", "fixed_two_feet", "feet_then_pelvis"])
def test_cooperative_and_per_lane_pik_kernels_agree(torch_cuda, case, monkeypatch):
    """ik::pik has two device forms: cooperative and LDS-resident (device/pik_coop.hpp, the default when the workspace fits and
    every lambda > 0) and per-lane with the workspace in HBM (device/pik_solver.hpp; IKGPU_GENERIC_KERNEL=lane, lambda = 0)."""
    torch = torch_cuda
    name, ff, specs, edit, projector_determined = PIK_CASES[case]
    B = 1003
    ik_amd, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=7, xml_edit=edit)
    levels = problem.max_priority_level() + 1
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, step, tol, lam in ((1, 1.0, -1.0, [1.0] * levels), (5, 1.0, -1.0, [0.1] * levels), (40, 0.5, 1e-8, [0.05, 0.2][:levels])):
        data = _pik_data(ik_amd, problem, lam)
        p, v = ik_amd.pik_parameters(max_iterations=iters, step_length=step), ik_amd.inverse_kinematics_visitor(tol)
        monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)
        monkeypatch.setenv("IKGPU_PIK_STATIC", "0")                   # (the interpreter forms: the compiled lane program is compared below)
        Qc, okc, itc = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        assert not data.kernel.endswith(",static>")
        Qc2, _, _ = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        assert torch.equal(Qc, Qc2)                                   # run-to-run bit-identical
        monkeypatch.setenv("IKGPU_GENERIC_KERNEL", "lane")
        Ql, okl, itl = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        assert torch.equal(okc, okl) and torch.equal(itc, itl), (case, iters)
        d = (Qc - Ql).abs().amax(dim=0)
        assert (d <= 1e-7).double().mean().item() >= 0.995, (case, iters, d.max().item())
    # lambda = 0 is the per-lane program's (the cooperative one factors Jbar Jbar^T + lambda^2 I): same entry point, finite result
    monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)
    monkeypatch.delenv("IKGPU_PIK_STATIC", raising=False)
    data = _pik_data(ik_amd, problem, [0.0] * levels)
    Qz, _, _ = ik_amd.pik_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.pik_parameters(max_iterations=2, step_length=0.5))
    assert torch.isfinite(Qz).double().mean().item() > 0.99


@pytest.mark.parametrize("name,ff,specs,kernel", [
    ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None)], "dls_chain<"),
    ("ur5", False, [("frame", "tool0", "universe", 2, 0, None)], "dls_chain<"),
    ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                      ("frame", "pelvis", "universe", 2, 0, None)], "dls_tree<"),
    ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                      ("align", "LeftFootFront", "universe", 1, 0, None)], "dls_tree<"),
])
def test_pik_with_one_level_is_dls_with_damping_lambda(torch_cuda, monkeypatch, name, ff, specs, kernel):
    """One priority level, no secondary step: pik's step is -J^T (J J^T + lambda^2 I)^-1 e, i.e. ik::dls with damping = lambda,
    and ikgpu_pik_solve_batch runs it on the problem's register-resident DLS kernel.  Checked against the oracle's ik::pik
    (an SVD-based damped pseudo-inverse, oracle/ik_oracle.c) and against the generic PIK kernel on the same handle."""
    torch = torch_cuda
    monkeypatch.delenv("IKGPU_PIK_KERNEL", raising=False)
    B = 515
    ik_amd, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=6)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, step, tol, lam in ((1, 1.0, -1.0, 1.0), (25, 1.0, 1e-10, 1e-2), (50, 0.5, -1.0, 0.1)):
        data = _pik_data(ik_amd, problem, [lam])
        assert data.kernel == ik_amd.plan(problem) and data.kernel.startswith(kernel)
        v, p = ik_amd.inverse_kinematics_visitor(tol), ik_amd.pik_parameters(max_iterations=iters, step_length=step)
        Qp, okp, itp = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, [lam], None), os.cpu_count() or 1)
        d = np.abs(Qp.cpu().numpy().T - q_ref).max(axis=1)
        if any(s[0] == "align" for s in specs) and step == 1.0 and iters > 1:
            # an alignment row with a direction the foot cannot reach, at full step and light damping: the iteration is chaotic
            # (oracle and twin part ways the same way, tests/test_gpu_generic.py CHAOTIC_AT_FULL_STEP) -- statistical parity
            assert (d <= TOL).mean() >= 0.97 and np.median(d) < 1e-10, (name, iters, d.max())
        else:
            assert np.array_equal(okp.cpu().numpy(), ok_ref) and np.array_equal(itp.cpu().numpy(), it_ref), (name, iters)
            assert d.max() <= TOL, (name, iters)
        Qd, okd, itd = ik_amd.dls_batch(problem, Q0, T, data, v, ik_amd.dls_parameters(max_iterations=iters, damping=lam, step_length=step))
        assert torch.equal(Qp, Qd) and torch.equal(okp, okd) and torch.equal(itp, itd)      # the very same launch
    monkeypatch.setenv("IKGPU_PIK_KERNEL", "generic")
    assert data.kernel.startswith("pik_generic<")
    Qg, okg, itg = ik_amd.pik_batch(problem, Q0, T, data, v, p)
    monkeypatch.delenv("IKGPU_PIK_KERNEL")
    assert torch.equal(okp, okg) and torch.equal(itp, itg)
    assert (Qp - Qg).abs().max().item() < 1e-7
    # a secondary step, or a constraint in the problem (which ik::pik does not read), keeps the call on the PIK kernel
    data.da = 0.01 * np.ones(model.nv)
    assert data.kernel.startswith("pik_generic<")
    data.da = np.zeros(model.nv)
    problem.add_frame_constraint("hold", ik_amd.FrameConstraint.create(model, specs[0][1], ik_amd.KinematicType.Orientation))
    data2 = _pik_data(ik_amd, problem, [lam])
    assert data2.kernel.startswith("pik_generic<")
    Qc, okc, itc = ik_amd.pik_batch(problem, Q0, T, data2, v, p)
    assert torch.equal(okc, okp) and (Qc - Qp).abs().max().item() < 1e-7


def test_demo_problem_through_pik_as_the_demo_declares_it(torch_cuda):
    """The demo's PIK branch (reference ik_ros/src/cassie.cpp:43,72-81,115-121): the problem is declared with max priority
    level 1 but every task sits on level 0, so pik_data carries two lambdas and the second level has no rows -- a no-op in
    the reference's loop (ik/ik/pik.cpp:47).  It is therefore the DLS iteration and runs on the tree kernel."""
    torch = torch_cuda
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    fl = problem.add_frame_task("fl", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Position, "pelvis"))
    pelvis = problem.add_frame_task("pelvis", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
    align = problem.add_align_axis_task("align", ik_amd.AlignAxisTask.create(model, "LeftFootFront", ik_amd.AlignAxisType.AxisY))
    data = ik_amd.pik_data(problem, device=0)
    assert data.lambda_ == [1.0, 1.0]
    data.lambda_ = [0.1, 1.0]
    assert data.kernel.startswith("dls_tree<") and "align_axis" in data.kernel
    B = 777
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names),
                                         np.arange(B), seed=3)
    om = O.OracleModel(model.flat())
    fids = [model.getFrameId("LeftFootFront"), model.getFrameId("pelvis")]
    ot = O.make_tasks([(fids[0], fids[1], 0, 0, None), (fids[1], 0, 2, 0, None), (fids[0], 0, 4, 0, None)])
    tg = np.zeros((B, 3, 12))
    for b in range(B):
        oMf = O.fk(om, qs[b])[1]
        Rp, pp, Rf, pf = oMf[fids[1]][:9].reshape(3, 3), oMf[fids[1]][9:], oMf[fids[0]][:9].reshape(3, 3), oMf[fids[0]][9:]
        tg[b, 0, :9], tg[b, 0, 9:] = (Rp.T @ Rf).ravel(), Rp.T @ (pf - pp)
        tg[b, 1] = oMf[fids[1]]
        tg[b, 2, :9], tg[b, 2, 9:] = np.eye(3).ravel(), Rf[:, 1]         # the direction the foot's Y axis has at q*
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    # the demo's own parameters (cassie.cpp:103-106 for DLS; its PIK branch keeps the defaults): small steps, many of them
    for iters, step, tol in ((1, 1.0, -1.0), (200, 0.1, 1e-4), (30, 0.5, -1.0)):
        Q, ok, it = ik_amd.pik_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol),
                                     ik_amd.pik_parameters(max_iterations=iters, step_length=step))
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, [0.1, 1.0]), os.cpu_count() or 1)
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), iters
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL, iters
    # one problem through ik::pik(), targets read from the task objects
    fl.target = ik_amd.SE3.from12(tg[5, 0])
    pelvis.target = ik_amd.SE3.from12(tg[5, 1])
    align.target[:] = tg[5, 2, 9:]
    q1 = ik_amd.pik(problem, q0[5], data, ik_amd.inverse_kinematics_visitor(1e-4), ik_amd.pik_parameters(max_iterations=200, step_length=0.1))
    q1_ref, ok1_ref, it1_ref = O.pik(om, ot, tg[5], q0[5], O.pik_params(200, 0.1, 1e-4, [0.1, 1.0]))
    assert np.abs(q1 - q1_ref).max() <= TOL and data.success == ok1_ref and data.iterations == it1_ref


def test_single_problem_pik_with_edited_lambda_and_da(torch_cuda):
    """ik::pik() as a caller of the reference would use it: pik_data owns lambda and da and both are read at every call."""
    import ik_amd
    import oracle as O
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    left = problem.add_frame_task("left", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    right = problem.add_frame_task("right", ik_amd.FrameTask.create(model, "RightFootFront", ik_amd.KinematicType.Position), 1)
    data = ik_amd.pik_data(problem)
    assert data.lambda_ == [1.0, 1.0] and not data.da.any()
    om = O.OracleModel(model.flat())
    rng = np.random.default_rng(8)
    q = q_ref = np.clip(rng.uniform(-0.2, 0.2, model.nq), model.lowerPositionLimit, model.upperPositionLimit)
    qs = np.clip(q + rng.uniform(-0.15, 0.15, model.nq), model.lowerPositionLimit, model.upperPositionLimit)
    oMf = O.fk(om, qs)[1]
    left.target = ik_amd.SE3.from12(oMf[model.getFrameId("LeftFootFront")])
    right.target = ik_amd.SE3.from12(oMf[model.getFrameId("RightFootFront")])
    ot = O.make_tasks([(model.getFrameId("LeftFootFront"), 0, 2, 0, None), (model.getFrameId("RightFootFront"), 0, 0, 1, None)])
    tg = np.stack([left.target.to12(), right.target.to12()])
    p = ik_amd.pik_parameters(max_iterations=20, step_length=0.7)
    for lam, da in (([1.0, 1.0], None), ([0.05, 0.3], None), ([0.05, 0.3], 0.02 * np.sin(np.arange(model.nv)))):
        data.lambda_ = lam
        data.da = np.zeros(model.nv) if da is None else da
        q = ik_amd.pik(problem, q, data, ik_amd.inverse_kinematics_visitor(1e-9), p)
        q_ref, ok_ref, it_ref = O.pik(om, ot, tg, q_ref, O.pik_params(20, 0.7, 1e-9, lam, None if da is None else list(da)))
        assert data.success == ok_ref and data.iterations == it_ref
        assert np.abs(q - q_ref).max() <= TOL


def test_pik_argument_errors(torch_cuda):
    import ik_amd
    from ik_amd import capi
    model = ik_amd.Model.from_urdf_file(urdf_path("ur5"))
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    problem.add_frame_task("a", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Position))
    problem.add_frame_task("b", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Orientation), 1)
    data = ik_amd.pik_data(problem)
    q0, tg = np.zeros((1, model.nq)), np.zeros((1, 2, 12))
    tg[:, :, [0, 4, 8]] = 1.0
    data.lambda_ = [1.0]                       # one level short
    with pytest.raises(capi.IkgpuError) as ei:
        ik_amd.pik_batch(problem, q0, tg, data, layout="aos")
    assert ei.value.code == capi.ERR_INVALID and "priority levels" in ei.value.message
    data.lambda_ = [1.0, -0.5]
    with pytest.raises(capi.IkgpuError):
        ik_amd.pik_batch(problem, q0, tg, data, layout="aos")
    data.lambda_ = [1.0, 1.0]
    Q, ok, it = ik_amd.pik_batch(problem, np.zeros((0, model.nq)), np.zeros((0, 2, 12)), data, layout="aos")   # empty batch
    assert Q.shape == (0, model.nq)
    p = ik_amd.pik_parameters(max_iterations=0)
    Q, ok, it = ik_amd.pik_batch(problem, q0 + 0.3, tg, data, p=p, layout="aos")
    assert np.array_equal(Q, q0 + 0.3) and ok[0] == 0 and it[0] == 0


def test_cpp_api_program_with_pik(torch_cuda):
    """ik::pik through the C++ mirror (tests/cpp/test_dls_api.cpp's `pik` option)."""
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from test_gpu_parity import _cpp_binary
    model = ik_amd.Model.from_urdf_file(urdf_path("ur5"))
    om = O.OracleModel(model.flat())
    rng = np.random.default_rng(12)
    q0 = np.array([0.1, -1.4, 1.5, 0.1, 1.4, 0.05]) + rng.uniform(-0.1, 0.1, 6)
    qs = q0 + rng.uniform(-0.15, 0.15, 6)
    fid = model.getFrameId("tool0")
    tg = np.stack([O.fk(om, qs)[1][fid]] * 2)
    ot = O.make_tasks([(fid, 0, 0, 0, None), (fid, 0, 1, 1, None)])
    args = [_cpp_binary(), urdf_path("ur5"), "0", "30", "0.01", "1.0", "1e-10", "2"]
    for typ, prio in ((0, 0), (1, 1)):
        args += ["tool0", str(typ), str(prio)] + ["%.17g" % x for x in tg[0]]
    args += ["%.17g" % x for x in q0] + ["pik", "0.05", "0.1"]
    out = json.loads(subprocess.check_output(args, text=True))
    q1, ok1, it1 = O.pik(om, ot, tg, q0, O.pik_params(30, 1.0, 1e-10, [0.05, 0.1]))
    q2, ok2, it2 = O.pik(om, ot, tg, q1, O.pik_params(30, 1.0, 1e-10, [0.05, 0.1]))
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL and np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2
    assert out["kernel"].startswith("pik_generic<")
    # one level: the C++ mirror's pik_data::kernel() names the DLS kernel the call ran on
    args = [_cpp_binary(), urdf_path("ur5"), "0", "30", "0.01", "1.0", "1e-10", "1", "tool0", "2", "0"]
    args += ["%.17g" % x for x in tg[0]] + ["%.17g" % x for x in q0] + ["pik", "0.05"]
    out = json.loads(subprocess.check_output(args, text=True))
    ot1 = O.make_tasks([(fid, 0, 2, 0, None)])
    q1, ok1, it1 = O.pik(om, ot1, tg[:1], q0, O.pik_params(30, 1.0, 1e-10, [0.05]))
    q2, ok2, it2 = O.pik(om, ot1, tg[:1], q1, O.pik_params(30, 1.0, 1e-10, [0.05]))
    assert out["kernel"].startswith("dls_chain<NJ=6")
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL and np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2


@pytest.mark.parametrize("case", ["ur5_pos_then_ori", "ur5_full_then_elbow", "fixed_two_feet", "feet_then_pelvis"])
def test_compiled_pik_lane_program_against_the_interpreter_and_the_oracle(torch_cuda, case, monkeypatch):
    """ik::pik for any level split on a lane program compiled for the problem (rtc.cpp, device/pik_solver.hpp static_pik: projector
    in factored form, each level's damped pseudo-inverse as a dual Cholesky solve, pivoted Gram-Schmidt with the reference's rank
    rule) -- what a problem with several priority levels runs on by default.  B = 8192 (not a multiple of 64 + a tail), against the
    oracle's Jacobi-SVD / COD ik::pik and the cooperative interpreter; lambda = 0 on a level falls back to the interpreter."""
    torch = torch_cuda
    if not _hiprtc():
        pytest.skip("libhiprtc is not installed")
    name, ff, specs, edit, projector_determined = PIK_CASES[case]
    B = 8192 + 37
    ik_amd, O, model, problem, _, om, ot, q0, tg = build(name, ff, specs, 512, seed=12, xml_edit=edit)
    rep = -(-B // 512)
    rng = np.random.default_rng(3)
    q0 = (np.tile(q0, (rep, 1))[:B] + rng.uniform(-0.02, 0.02, (B, model.nq)) * (np.arange(B) >= 512)[:, None])
    if ff:
        q0[:, 3:7] /= np.linalg.norm(q0[:, 3:7], axis=1, keepdims=True)
    q0 = np.clip(q0, model.lowerPositionLimit, model.upperPositionLimit)
    tg = np.tile(tg, (rep, 1, 1))[:B]
    levels = problem.max_priority_level() + 1
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)
    for iters, step, tol, lam, da in ((1, 1.0, -1.0, [1.0] * levels, None), (30, 0.5, 1e-8, [0.05, 0.1][:levels], None),
                                      (50, 1.0, -1.0, [0.1] * levels, None), (6, 1.0, -1.0, [0.1] * levels, 0.01 * np.cos(np.arange(model.nv)))):
        if da is not None and not projector_determined:
            continue
        data = _pik_data(ik_amd, problem, lam, da)
        assert data.kernel.startswith("pik_generic<") and data.kernel.endswith(",static>"), data.kernel
        p, v = ik_amd.pik_parameters(max_iterations=iters, step_length=step), ik_amd.inverse_kinematics_visitor(tol)
        Qs, oks, its = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, None if da is None else list(da)), os.cpu_count() or 1)
        same = its.cpu().numpy() == it_ref
        assert same.mean() > 0.999 and np.array_equal(oks.cpu().numpy()[same], ok_ref[same]), (case, iters, same.mean())
        d = np.abs(Qs.cpu().numpy().T - q_ref).max(axis=1)
        # (50 full steps at lambda = 0.1 on far targets: the iteration is chaotic on a few problems, as for ik::dls -- tests/test_gpu_generic.py)
        assert (d[same] <= TOL).mean() >= (0.99 if iters == 50 else 1.0), (case, iters, d[same].max(), (d[same] <= TOL).mean())
        monkeypatch.setenv("IKGPU_PIK_STATIC", "0")
        Qc, okc, itc = ik_amd.pik_batch(problem, Q0, T, data, v, p)
        monkeypatch.delenv("IKGPU_PIK_STATIC")
        agree = (its == itc)
        assert agree.double().mean().item() > 0.999
        dc = (Qs - Qc).abs().amax(dim=0)[agree]
        assert (dc <= 1e-7).double().mean().item() >= (0.99 if iters == 50 else 0.999), (case, iters, dc.max().item())
    # AoS gives the same bits; a second call the same bits (no atomics, no races)
    Qa, oka, ita = ik_amd.pik_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data, v, p, layout="aos")
    assert torch.equal(Qs.T.contiguous(), Qa) and torch.equal(oks, oka) and torch.equal(its, ita)
    # lambda = 0 on a level: the one-sided-Jacobi interpreter (the compiled program factors Jbar Jbar^T + lambda^2 I)
    data0 = _pik_data(ik_amd, problem, [0.0] + [0.1] * (levels - 1))
    assert not data0.kernel.endswith(",static>")
