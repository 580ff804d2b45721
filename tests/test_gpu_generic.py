"""GPU parity of the generic fallback kernel (ik_amd/csrc/device/generic_solver.hpp) through the C ABI:
shapes the register-resident specialisations do not take -- fixed-base multi-task problems, tasks that share
joints, reference frames that move with q, prismatic joints, AlignAxisTask rows (the demo's own task set,
reference ik_ros/src/cassie.cpp:45-81) -- against the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def T4(m12):
    M = np.eye(4)
    M[:3, :3] = np.asarray(m12[:9]).reshape(3, 3)
    M[:3, 3] = m12[9:]
    return M


def build(name, ff, specs, B, seed=0, xml_edit=None, static=True, device=True):
    """specs: (kind 'frame'|'align', frame, reference, type-or-axis, priority, weights).
    static: False keeps a generic problem off the run-time specialised lane program (IKGPU_GENERIC_STATIC=0 when the handle is
    created): the cooperative / per-lane memory-resident forms then run."""
    import ik_amd
    import oracle as O
    from ik_amd import workload
    seed += int(os.environ.get("IKGPU_TEST_SEED_OFFSET", "0"))   # other draws of every workload built here, against the same assertions
    xml = open(urdf_path(name), "rb").read()
    if xml_edit:
        xml = xml_edit(xml)
    model = ik_amd.Model.from_urdf_xml(xml, free_flyer=ff)
    problem = ik_amd.InverseKinematicsProblem(model, max(s[4] for s in specs))
    for i, (kind, f, r, t, p, w) in enumerate(specs):
        if kind == "com":       # r = reference frame
            task = problem.add_centre_of_mass_task(ik_amd.CentreOfMassTask.create(model, r), p)
            if w is not None:
                task.weighting()[:] = w
            continue
        if kind == "posture":   # f = nj, w = (weights, mask)
            task = problem.add_posture_task("t%d" % i, ik_amd.PostureTask.create(model, f), p)
            task.weighting()[:], task.mask[:] = w
            continue
        if kind == "align":
            task = problem.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(model, f, ik_amd.AlignAxisType(t), r), p)
        else:
            task = problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t), r), p)
        if w is not None:
            task.weighting()[:] = w
    prev = os.environ.get("IKGPU_GENERIC_STATIC")
    if not static:
        os.environ["IKGPU_GENERIC_STATIC"] = "0"
    try:
        data = ik_amd.dls_data(problem, device=0) if device else None   # (device=False: the problem and the oracle's view of it only)
    finally:
        if not static:
            if prev is None:
                del os.environ["IKGPU_GENERIC_STATIC"]
            else:
                os.environ["IKGPU_GENERIC_STATIC"] = prev
    if not static and device:
        assert not data.kernel.endswith(",static>"), data.kernel
    om = O.OracleModel(model.flat())
    ordered = problem.ordered_tasks()
    ospec = []
    for t, prio in ordered:
        if isinstance(t, ik_amd.PostureTask):   # nj rows: (tangent column, q index, IKGPU_POSTURE_ROW, priority, [weight, mask])
            ospec += [(model.nv - t.nj + k, model.nq - t.nj + k, 6, prio, [t.weighting()[k], t.mask[k]]) for k in range(t.nj)]
            continue
        w = None if np.all(t.weighting() == 1) else list(t.weighting())
        if isinstance(t, ik_amd.CentreOfMassTask):
            ospec.append((0, t._ref_id, 7, prio, w))
            continue
        ospec.append((t._frame_id, t._ref_id, 3 + int(t.axis) if isinstance(t, ik_amd.AlignAxisTask) else int(t.type), prio, w))
    rng = np.random.default_rng(seed)
    if ff:
        q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                             workload.cassie_nominal(model.names), np.arange(B), seed=seed)
    else:
        lo, hi = np.maximum(model.lowerPositionLimit, -2.5), np.minimum(model.upperPositionLimit, 2.5)
        q0 = np.clip(0.5 * (lo + hi) + rng.uniform(-0.2, 0.2, (B, model.nq)), lo, hi)
        qs = np.clip(q0 + rng.uniform(-0.15, 0.15, (B, model.nq)), lo, hi)
    tg = np.zeros((B, len(ospec), 12))
    for b in range(B):
        _, oMf = O.fk(om, qs[b])
        for i, (fid, rid, typ, _, _) in enumerate(ospec):
            if typ == 7:   # centre of mass at q*, seen from the reference frame
                tg[b, i, :9] = np.eye(3).ravel()
                tg[b, i, 9:] = O.evaluate(om, O.make_tasks([(0, rid, 7, 0, None)]), np.zeros((1, 12)), qs[b])[0]
            elif typ == 6:
                tg[b, i, 9] = qs[b, rid]
            elif typ >= 3:
                tg[b, i, :9] = np.eye(3).ravel()
                tg[b, i, 9:] = rng.normal(size=3)
            else:
                rel = np.linalg.inv(T4(oMf[rid])) @ T4(oMf[fid])
                tg[b, i] = np.concatenate([rel[:3, :3].ravel(), rel[:3, 3]])
    return ik_amd, O, model, problem, data, om, O.make_tasks(ospec), q0, tg


def _prismatic_elbow(xml):
    return xml.replace(b'<joint name="elbow_joint" type="revolute">', b'<joint name="elbow_joint" type="prismatic">', 1) \
              .replace(b'lower="-3.14159265359" upper="3.14159265359"', b'lower="-0.2" upper="0.2"', 1)


def _tail(n):
    """A chain of n extra revolute joints off the pelvis (nv = 22 + n): the tangent-space sizes the Cassie / UR5 fixtures do not reach."""
    def edit(xml):
        extra, parent = [], "pelvis"
        for k in range(n):
            link = "tail%d" % k
            extra.append('<link name="%s"><inertial><mass value="0.%d"/><origin rpy="0 0 0" xyz="0.01 0 0.02"/></inertial></link>' % (link, k % 9 + 1))
            extra.append('<joint name="tail-joint%d" type="revolute"><parent link="%s"/><child link="%s"/><origin rpy="0 0 0" xyz="0.05 0 %s"/>'
                         '<axis xyz="%s"/><limit lower="-1.5" upper="1.5"/></joint>' % (k, parent, link, "0.03" if k % 2 else "-0.02", ("0 0 1", "0 1 0", "1 0 0")[k % 3]))
            parent = link
        return xml.replace(b"</robot>", ("\n".join(extra) + "\n</robot>").encode())
    return edit


CASES = {
    # the demo's own task set (reference ik_ros/src/cassie.cpp:45-81) -- since the tree kernel learnt base-relative references
    # and an alignment row it runs there (dls_tree<7,1,base_task,base_reference,align_axis>); kept here for its stage kernels ...
    "demo_task_set": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                       ("align", "LeftFootFront", "universe", 1, 0, None)], None),
    # ... and the same with the alignment direction given in the pelvis frame, which stays on the generic kernel
    "demo_with_direction_in_pelvis_frame": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None),
                                                             ("frame", "pelvis", "universe", 2, 0, None),
                                                             ("align", "LeftFootFront", "pelvis", 1, 0, None)], None),
    "fixed_two_feet_priorities": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None),
                                                          ("frame", "RightFootBack", "universe", 0, 1, [2.0, 1.0, 0.5])], None),
    "shared_joints": ("ur5", False, [("frame", "tool0", "universe", 0, 0, None), ("frame", "forearm_link", "universe", 1, 0, None)], None),
    "moving_reference_prismatic": ("ur5", False, [("frame", "tool0", "upper_arm_link", 2, 0, None)], _prismatic_elbow),
    "three_feet_frames": ("cassie", True, [("frame", "LeftFootFront", "universe", 0, 0, None), ("frame", "LeftFootBack", "universe", 0, 0, None),
                                           ("frame", "RightFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 1, 1, None)], None),
    # M = 21 > 15: the cooperative kernel's LDS Gram matrix / Cholesky (the register forms cover M <= 15)
    "feet_frames_beyond_the_register_solve": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "LeftFootBack", "universe", 0, 0, None),
                                                               ("frame", "RightFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 2, 0, None)], None),
    # the boundaries of the register solves: M = 16 (first size with two rows per lane), M = 31 (last), M = 32 (the LDS form)
    "rows_16": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "LeftFootBack", "universe", 0, 0, None),
                                 ("frame", "RightFootFront", "universe", 2, 0, None), ("align", "RightFootFront", "universe", 1, 0, None)], None),
    "rows_31": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "LeftFootBack", "universe", 2, 0, None),
                                 ("frame", "RightFootFront", "universe", 2, 0, None), ("frame", "RightFootBack", "universe", 2, 0, None),
                                 ("frame", "pelvis", "universe", 2, 0, None), ("align", "RightFootFront", "universe", 1, 0, None)], None),
    "rows_32": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "LeftFootBack", "universe", 2, 0, None),
                                 ("frame", "RightFootFront", "universe", 2, 0, None), ("frame", "RightFootBack", "universe", 2, 0, None),
                                 ("frame", "pelvis", "universe", 2, 0, None), ("align", "RightFootFront", "universe", 1, 0, None),
                                 ("align", "LeftFootFront", "universe", 0, 0, None)], None),
    # tangent-space sizes beyond the fixtures': nv = 30 (the register Gram matrix's widest build), nv = 36 (its LDS form)
    "nv_30": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "tail7", "universe", 0, 0, None),
                               ("frame", "pelvis", "universe", 2, 0, None)], _tail(8)),
    "nv_36": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "tail13", "universe", 0, 0, None),
                               ("frame", "pelvis", "universe", 2, 0, None)], _tail(14)),
    # ik::PostureTask (reference ik/ik/posture.hpp:17-85) regularising two pose tasks, with weights and a mask with holes
    "posture_regulariser": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                             ("posture", 16, None, None, 1, ([0.1 + 0.05 * k for k in range(16)],
                                                                             [0.0 if k in (3, 9) else 1.0 for k in range(16)]))], None),
    # ik::CentreOfMassTask (reference ik/ik/centre_of_mass.hpp:14-62): the demo's commented-out balance task under two foot
    # poses; seen from a moving frame with weights; alone on the arm
    "com_under_feet": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                        ("com", None, "universe", None, 1, None)], None),
    "com_in_foot_frame": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 0, 0, None),
                                                  ("com", None, "LeftFootFront", None, 0, [1.0, 2.0, 0.5])], None),
    "com_of_the_arm": ("ur5", False, [("com", None, "universe", None, 0, None)], None),
    "posture_first_level": ("ur5", False, [("posture", 4, None, None, 0, ([1.0] * 4, [1.0] * 4)),
                                           ("frame", "tool0", "universe", 0, 0, None)], None),
    # every line of the reference demo switched on but the pinned foot (ik_ros/src/cassie.cpp:45-81; M = 29): the problem whose run-time
    # compilation ABORTED the host process in round 3 (gpurun_out/abort.log).  Its lane program now comes from the compile worker
    # (rtc.cpp); here it goes through the gfx950 backend on the GPU box and against the oracle
    "demo_everything_on_unconstrained": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                                          ("align", "LeftFootFront", "universe", 1, 0, None),
                                                          ("posture", 16, None, None, 0, ([0.3 + 0.04 * k for k in range(16)], [1.0] * 16)),
                                                          ("com", None, "universe", None, 0, None)], None),
}


# (the demo task set with a random, in general unreachable, direction for its alignment row never settles at full step
# either: at 40 full steps one problem in 500 sits a hair above the bar, 1.09e-6)
# (rows_16: a pose and a position task on one foot plus an alignment row with a random direction -- at 40 full steps 77 % of the
# problems agree to the bar, both forms of the kernel part from the oracle alike: tools/forms_vs_oracle.py)
CHAOTIC_AT_FULL_STEP = {"com_in_foot_frame", "demo_task_set", "demo_with_direction_in_pelvis_frame", "rows_16", "demo_everything_on_unconstrained"}


# Problems that have since found a register-resident kernel (the tree kernel with posture rows or on a fixed base,
# tests/test_gpu_tree_posture.py, tests/test_gpu_tree_fixed_base.py): here they are kept on the generic kernel by
# IKGPU_DLS_KERNEL=generic
FORCED_GENERIC = {"posture_regulariser", "fixed_two_feet_priorities", "posture_first_level"}


def assert_within_bar_or_oracle_unstable(solve, q_dev, q_ref, tg, q0, label):
    """|q_dev - q_ref| <= 1e-6 rad on every lane -- or, for at most one lane in a hundred, a lane the ORACLE cannot answer to 1e-7 either:
    its own result moves by more than that under 1e-13 perturbations of its inputs (the exclusion rule of tests/test_gpu_full_size.py,
    decided by the oracle alone: three draws along (1, ..., 1), then sixteen with random signs).  A lane parked on a joint limit for 200
    small steps amplifies rounding; with the suite's own seeds no case needs this, with IKGPU_TEST_SEED_OFFSET one lane in 500 does
    (both forms of the kernel then part from the oracle by the same 1e-4 ... 6e-2 while agreeing with it to 1e-12 step by step).
    solve(targets, q0) runs the oracle with the case's parameters."""
    from test_gpu_full_size import oracle_sensitivity_more_draws
    d = np.abs(q_dev - q_ref).max(axis=1)
    bad = np.flatnonzero(d > TOL)
    if not bad.size:
        return
    assert bad.size <= max(1, q0.shape[0] // 100), (label, bad.size, d.max())
    sens = np.zeros(bad.size)
    for dq, dt in ((1e-13, 0.0), (0.0, 1e-13), (-1e-13, -1e-13)):
        tgp = tg[bad].copy()
        tgp[:, :, 9:] += dt
        sens = np.maximum(sens, np.abs(solve(tgp, q0[bad] + dq)[0] - q_ref[bad]).max(axis=1))
    sens = np.maximum(sens, oracle_sensitivity_more_draws(lambda t_, q_, _e: solve(t_, q_), tg, q0, q_ref, bad))
    assert (sens > 1e-7).all(), (label, bad, d[bad], sens)


@pytest.mark.parametrize("static", [True, False], ids=["static", "coop"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_generic_kernel_matches_oracle(torch_cuda, case, static, monkeypatch):
    """Every generic problem on BOTH forms a handle can get: the lane program specialised for the problem at run time ("...,static>",
    rtc.cpp) and the cooperative LDS-resident kernel (IKGPU_GENERIC_STATIC=0)."""
    torch = torch_cuda
    name, ff, specs, edit = CASES[case]
    B = 500  # not a multiple of 64
    if case in FORCED_GENERIC:
        monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")
    ik_amd, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, B, xml_edit=edit, static=static)
    if case == "demo_task_set" and not static:
        pytest.skip("the demo task set runs on the tree kernel either way")
    assert data.kernel.startswith("dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis>" if case == "demo_task_set" else "dls_generic<")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    e, J = ik_amd.evaluate_batch(problem, Q0, T, data)
    e, J = e.cpu().numpy().T, J.permute(2, 0, 1).cpu().numpy()
    for b in range(0, B, 7):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-10 and np.abs(J[b] - Jo).max() < 1e-10
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4), (40, 1e-2, 1.0, 1e-6)):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
        if case in CHAOTIC_AT_FULL_STEP and step == 1.0 and iters > 3:
            # full undamped-ish steps on a task whose Jacobian leaves out the motion of its own reference frame (the reference's
            # choice, ik/ik/centre_of_mass.hpp:41-45) bounce between joint limits: rounding differences grow like in the
            # far-target UR5 case; the oracle and the twin part ways the same way (15 % of such problems
            # beyond 1e-6, up to 1.2 rad, between those two CPU programs).  Most problems still agree to the bar; the
            # step-wise configurations above and the small-step one pin the arithmetic.
            d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
            assert (d <= TOL).mean() > 0.7, (case, iters, (d <= TOL).mean())
            continue
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (case, iters)
        prm_o = O.params(iters, damping, step, tol)
        assert_within_bar_or_oracle_unstable(lambda t_, q_: O.dls_batch(om, ot, t_, q_, prm_o, os.cpu_count() or 1), Q.cpu().numpy().T, q_ref, tg, q0, (case, iters))
    # AoS gives the same bits
    Qa, oka, ita = ik_amd.dls_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data,
                                    ik_amd.inverse_kinematics_visitor(tol), p, layout="aos")
    assert torch.equal(Q.T.contiguous(), Qa) and torch.equal(ok, oka) and torch.equal(it, ita)


@pytest.mark.parametrize("case", ["demo_with_direction_in_pelvis_frame", "fixed_two_feet_priorities", "shared_joints", "moving_reference_prismatic",
                                  "three_feet_frames", "posture_regulariser", "posture_first_level", "com_under_feet", "com_of_the_arm"])
def test_cooperative_and_per_lane_generic_kernels_agree(torch_cuda, case, monkeypatch):
    """The generic DLS kernel has two forms: the cooperative LDS-resident one (device/coop_solver.hpp, the default when
    the problem fits) and the per-lane memory-resident one (device/generic_solver.hpp; IKGPU_GENERIC_KERNEL=lane)."""
    torch = torch_cuda
    name, ff, specs, edit = CASES[case]
    B = 1003   # not a multiple of 4 (problems per workgroup) nor of 64
    if case in FORCED_GENERIC:
        monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")
    ik_amd, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, B, seed=3, xml_edit=edit, static=False)
    assert data.kernel.startswith("dls_generic<") and not data.kernel.endswith(",static>")
    data_static = build(name, ff, specs, B, seed=3, xml_edit=edit, static=True)[4]
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4)):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        v = ik_amd.inverse_kinematics_visitor(tol)
        monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)
        Qc, okc, itc = ik_amd.dls_batch(problem, Q0, T, data, v, p)
        if data_static.kernel.endswith(",static>"):     # the run-time specialised lane program: same flags, same q to 1e-9
            Qs, oks, its = ik_amd.dls_batch(problem, Q0, T, data_static, v, p)
            assert torch.equal(okc, oks) and torch.equal(itc, its), (case, iters, "static")
            assert (Qc - Qs).abs().max().item() < 1e-9, (case, iters, "static")
        Qc2, _, _ = ik_amd.dls_batch(problem, Q0, T, data, v, p)
        assert torch.equal(Qc, Qc2)                                  # run-to-run bit-identical (no lane races)
        monkeypatch.setenv("IKGPU_GENERIC_KERNEL", "lane")
        Ql, okl, itl = ik_amd.dls_batch(problem, Q0, T, data, v, p)
        assert torch.equal(okc, okl) and torch.equal(itc, itl), (case, iters)
        assert (Qc - Ql).abs().max().item() < 1e-9, (case, iters)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
        # against the oracle: all but the odd ill-conditioned problem of this larger, differently seeded batch (where the two
        # kernels still agree with each other to 1e-9, above)
        d = np.abs(Qc.cpu().numpy().T - q_ref).max(axis=1)
        assert (d <= TOL).mean() >= 0.995, (case, iters, (d <= TOL).mean(), d.max())


@pytest.mark.parametrize("case,kernel", [("demo_task_set", "dls_tree<"), ("demo_with_direction_in_pelvis_frame", "dls_generic<"),
                                         ("fixed_two_feet_priorities", "dls_tree<NJ=7,chains=2,fixed_base>")])
def test_full_size_properties_of_the_demo_and_generic_kernels(torch_cuda, case, kernel):
    """Full batch (65536), size-independent properties instead of an oracle run: the evaluated error of the solution
    vanishes where the solve converged, solving again from the solution moves nothing, two runs give the same bits, base
    quaternions stay unit, joints stay inside their limits, and the stop rule's bookkeeping is consistent."""
    torch = torch_cuda
    name, ff, specs, edit = CASES[case]
    B = 65536
    import ik_amd
    from ik_amd import workload
    _, _, model, problem, data, _, _, q_small, tg_small = build(name, ff, specs, 8)
    assert data.kernel.startswith(kernel)
    idx = np.arange(B)
    if ff:
        q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), idx, seed=2)
    else:
        lo, hi = np.maximum(model.lowerPositionLimit, -2.5), np.minimum(model.upperPositionLimit, 2.5)
        q0, qs = workload.chain_workload(lo, hi, 0.5 * (lo + hi), idx, seed=2, mode="near")
    Q0, QS = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda(), torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
    # reachable targets: every task's error vanishes at q*; build them on the device from the stage kernels
    F = ik_amd.task_frames_fk_batch(problem, QS, data)                     # world placement of each task's frame at q*
    T = F.clone()
    ordered = problem.ordered_tasks()
    for i, (t, _) in enumerate(ordered):
        if isinstance(t, ik_amd.AlignAxisTask):                            # the direction the axis has at q*, in the reference frame
            j = next(k for k, (u, _) in enumerate(ordered) if isinstance(u, ik_amd.FrameTask) and u.frame == t.reference_frame) \
                if t.reference_frame != "universe" else None
            axis_w = F[i, :9].reshape(3, 3, B)[:, int(t.axis), :]
            T[i, 9:] = axis_w if j is None else torch.einsum("kib,kb->ib", F[j, :9].reshape(3, 3, B), axis_w)
            T[i, :9] = torch.eye(3, dtype=torch.float64, device="cuda").reshape(9, 1)
        elif t.reference_frame != "universe":                              # pose w.r.t. the reference frame at q*
            j = next(k for k, (u, _) in enumerate(ordered) if isinstance(u, ik_amd.FrameTask) and u.frame == t.reference_frame)
            Rr, pr = F[j, :9].reshape(3, 3, B), F[j, 9:]
            T[i, :9] = torch.einsum("kib,kjb->ijb", Rr, F[i, :9].reshape(3, 3, B)).reshape(9, B)
            T[i, 9:] = torch.einsum("kib,kb->ib", Rr, F[i, 9:] - pr)
    e_star, _ = ik_amd.evaluate_batch(problem, QS, T, data, jacobian=False)
    assert e_star.abs().max().item() < 1e-12                               # the targets are what they are meant to be
    v, p = ik_amd.inverse_kinematics_visitor(1e-16), ik_amd.dls_parameters(max_iterations=120, damping=1e-2, step_length=0.5)
    Q1, ok1, it1 = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    Q2, ok2, it2 = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    assert torch.equal(Q1, Q2) and torch.equal(ok1, ok2) and torch.equal(it1, it2)
    e1, _ = ik_amd.evaluate_batch(problem, Q1, T, data, jacobian=False)
    # (the alignment row's error 1 - cos has a vanishing gradient at its solution: the demo sets converge sublinearly,
    # the oracle reaches 1e-6 .. 3e-6 on them after the same 120 iterations)
    conv_tol, still_tol = (1e-4, 1e-3) if case.startswith("demo") else (1e-7, 1e-6)
    conv = e1.abs().amax(dim=0) < conv_tol
    assert conv.double().mean().item() > 0.9
    # success <=> the priority-0 error met the tolerance at the iteration reported; a solve that ran out reports max_iterations
    assert bool(((ok1 == 1) == (it1 < 120)).all())
    Q3, _, _ = ik_amd.dls_batch(problem, Q1, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=3, damping=1e-2, step_length=0.5))
    assert (Q3 - Q1)[:, conv].abs().max().item() < still_tol
    s = 7 if ff else 0
    if ff:
        assert ((Q1[3:7] ** 2).sum(0).sqrt() - 1).abs().max().item() < 1e-12
    lo_t = torch.from_numpy(model.lowerPositionLimit).cuda()[s:, None]
    hi_t = torch.from_numpy(model.upperPositionLimit).cuda()[s:, None]
    assert bool(((Q1[s:] >= lo_t) & (Q1[s:] <= hi_t)).all())


def test_generic_kernel_task_frames_fk(torch_cuda):
    torch = torch_cuda
    name, ff, specs, edit = CASES["shared_joints"]
    ik_amd, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, 100)
    got = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(q0.T)).cuda(), data).permute(2, 0, 1).cpu().numpy()
    want = O.fk_batch(om, q0, [model.getFrameId("tool0"), model.getFrameId("forearm_link")])
    assert np.abs(got - want).max() < 1e-13


def test_single_problem_demo_loop(torch_cuda):
    """The demo's loop (reference ik_ros/src/cassie.cpp:92-112): same tasks, same parameters (200 iterations,
    damping 0.1, step 0.1), warm-started from the previous solution, through the reference-shaped dls()."""
    import ik_amd
    import oracle as O
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    fl = problem.add_frame_task("fl", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Position, "pelvis"))
    pelvis = problem.add_frame_task("pelvis", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
    align = problem.add_align_axis_task("align", ik_amd.AlignAxisTask.create(model, "LeftFootFront", ik_amd.AlignAxisType.AxisY))
    align.target = np.array([1.0, 0.0, 0.0])
    data = ik_amd.dls_data(problem)
    om = O.OracleModel(model.flat())
    ot = O.make_tasks([(model.getFrameId("LeftFootFront"), model.getFrameId("pelvis"), 0, 0, None),
                       (model.getFrameId("pelvis"), 0, 2, 0, None), (model.getFrameId("LeftFootFront"), 0, 4, 0, None)])
    p = ik_amd.dls_parameters(max_iterations=200, damping=1e-1, step_length=1e-1)
    q = np.zeros(model.nq)
    q[6] = 1.0                               # cassie.cpp:66-68
    q_ref = q.copy()
    for tick in range(3):
        fl.target.translation[:] = [0.0, 0.1, -0.6 + 0.2 * np.sin(0.5 * tick)]   # cassie.cpp:95-96
        tg = np.stack([fl.target.to12(), pelvis.target.to12(), np.concatenate([np.eye(3).ravel(), align.target])])
        q = ik_amd.dls(problem, q, data, ik_amd.inverse_kinematics_visitor(), p)
        q_ref, ok_ref, it_ref = O.dls(om, ot, tg, q_ref, O.params(200, 1e-1, 1e-1, 1e-4))
        assert data.success == ok_ref and data.iterations == it_ref
        assert np.abs(q - q_ref).max() <= TOL


def test_single_problem_with_posture_task(torch_cuda):
    """dls() with a PostureTask whose `target` / `mask` / weighting() are edited between solves, as a caller of the
    reference would (ik/ik/posture.hpp:75-82)."""
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    foot = problem.add_frame_task("fl", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    posture = problem.add_posture_task("posture", ik_amd.PostureTask.create(model, 16), 1)
    assert problem.get_posture_task("posture") is posture and problem.e_size(1) == 16 and problem.target_slots() == 17
    data = ik_amd.dls_data(problem)
    assert data.rows == 22 and data.kernel == "dls_tree<NJ=7,chains=1,posture>"   # rows on the chain joints and on the nine others
    om = O.OracleModel(model.flat())
    nom = workload.cassie_nominal(model.names)
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(1), seed=5)
    _, oMf = O.fk(om, qs[0])
    foot.target = ik_amd.SE3.from12(oMf[model.getFrameId("LeftFootFront")])
    p = ik_amd.dls_parameters(max_iterations=30, damping=1e-2, step_length=0.5)
    q = q_ref = q0[0]
    for round_, (w, hole) in enumerate(((0.05, None), (0.5, 4))):
        posture.target[:] = nom + 0.01 * round_
        posture.weighting()[:] = w
        posture.mask[:] = 1.0
        if hole is not None:
            posture.mask[hole] = 0.0
        ot = O.make_tasks([(model.getFrameId("LeftFootFront"), 0, 2, 0, None)] +
                          [(model.nv - 16 + k, model.nq - 16 + k, 6, 1, [w, posture.mask[k]]) for k in range(16)])
        tg = np.zeros((17, 12))
        tg[0] = foot.target.to12()
        tg[1:, 9] = posture.target
        q = ik_amd.dls(problem, q, data, ik_amd.inverse_kinematics_visitor(1e-8), p)
        q_ref, ok_ref, it_ref = O.dls(om, ot, tg, q_ref, O.params(30, 1e-2, 0.5, 1e-8))
        assert data.success == ok_ref and data.iterations == it_ref
        assert np.abs(q - q_ref).max() <= TOL


def test_cpp_api_program_with_posture_task(torch_cuda):
    """ik::PostureTask through the C++ mirror (tests/cpp/test_dls_api.cpp's `posture` option)."""
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from ik_amd import workload
    from test_gpu_parity import _cpp_binary
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    om = O.OracleModel(model.flat())
    nom = workload.cassie_nominal(model.names)
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(3), seed=9)
    q0, qs = q0[2], qs[2]
    fid = model.getFrameId("RightFootFront")
    tg = np.zeros((17, 12))
    tg[0] = O.fk(om, qs)[1][fid]
    tg[1:, 9] = nom
    ot = O.make_tasks([(fid, 0, 2, 0, None)] + [(model.nv - 16 + k, model.nq - 16 + k, 6, 1, [0.25, 1.0]) for k in range(16)])
    args = [_cpp_binary(), urdf_path("cassie"), "1", "30", "0.01", "1.0", "1e-6", "1", "RightFootFront", "2", "0"]
    args += ["%.17g" % x for x in tg[0]] + ["%.17g" % x for x in q0]
    args += ["posture", "16", "1", "0.25"] + ["%.17g" % x for x in nom]
    out = json.loads(subprocess.check_output(args, text=True))
    assert out["kernel"] == "dls_tree<NJ=7,chains=1,posture>"
    q1, ok1, it1 = O.dls(om, ot, tg, q0, O.params(30, 0.01, 1.0, 1e-6))
    q2, ok2, it2 = O.dls(om, ot, tg, q1, O.params(30, 0.01, 1.0, 1e-6))
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL
    assert np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2


def test_cpp_api_program_with_centre_of_mass_task(torch_cuda):
    """ik::CentreOfMassTask through the C++ mirror (tests/cpp/test_dls_api.cpp's `com` option): a foot pose at priority 0,
    the centre of mass over a target point at priority 1."""
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from ik_amd import workload
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
    out = json.loads(subprocess.check_output(args, text=True))
    assert out["kernel"].startswith("dls_generic<M=9")
    q1, ok1, it1 = O.dls(om, ot, tg, q0, O.params(30, 0.01, 1.0, 1e-8))
    q2, ok2, it2 = O.dls(om, ot, tg, q1, O.params(30, 0.01, 1.0, 1e-8))
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL and np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2


def test_cpp_demo_program_matches_oracle(torch_cuda):
    """tests/cpp/test_demo_loop.cpp is the reference's demo (ik_ros/src/cassie.cpp) on the C++ mirror."""
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_demo_loop")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ik_amd", "csrc", "host"), "-o", exe, src,
                               "-L" + os.path.join(ROOT, "ik_amd"), "-likgpu", "-Wl,-rpath," + os.path.join(ROOT, "ik_amd")])
    out = json.loads(subprocess.check_output([exe, urdf_path("cassie"), "4"], text=True))
    assert out["kernel"] == "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis>"   # the demo has a register-resident kernel
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    om = O.OracleModel(model.flat())
    lf, pe = model.getFrameId("LeftFootFront"), model.getFrameId("pelvis")
    ot = O.make_tasks([(lf, pe, 0, 0, None), (pe, 0, 2, 0, None), (lf, 0, 4, 0, None)])
    q = np.zeros(model.nq)
    q[6] = 1.0
    I9 = np.eye(3).ravel()
    for k, tick in enumerate(out["ticks"]):
        tg = np.stack([np.concatenate([I9, [0.0, 0.1, -0.6 + 0.2 * np.sin(0.5 * k)]]), np.concatenate([I9, np.zeros(3)]),
                       np.concatenate([I9, [1.0, 0.0, 0.0]])])
        q, ok, it = O.dls(om, ot, tg, q, O.params(200, 1e-1, 1e-1, 1e-4))
        assert tick["success"] == int(ok) and tick["iterations"] == it
        assert np.abs(np.array(tick["q"]) - q).max() <= TOL
