"""The generic lane program specialised for ONE problem at run time (rtc.cpp, device/generic_solver.hpp with IKD_STATIC_TABLES): the
model tree and the task table are compile-time constants, the workspace is a register array, every loop is unrolled.  It is what a
generic problem runs on by default ("dls_generic<...,static>") and what a Tree-kind problem with few rows is routed to
(capi.cpp tree_prefers_static).  Here: the routing, parity with the oracle, and agreement with the kernels it replaces."""
import os

import numpy as np
import pytest

from conftest import urdf_path
from test_gpu_generic import CASES, build

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


DEMO = [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None), ("align", "LeftFootFront", "universe", 1, 0, None)]
ROUTED = {
    # name: (model, free-flyer, task specs, constraint (frame, type) or None)
    "demo_task_set": ("cassie", True, DEMO, None),
    "demo_right_foot_pinned": ("cassie", True, DEMO, ("RightFootFront", 0)),
    "fixed_two_feet_positions": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 0, 0, None), ("frame", "RightFootFront", "universe", 0, 0, None)], None),
    # posture regulariser on all 16 joints: M = 26, the static program eliminates the posture rows and solves a 10 x 10 system
    "demo_with_posture": ("cassie", True, DEMO + [("posture", 16, None, None, 1, ([0.05] * 16, [1.0] * 16))], None),
    "pelvis_and_foot": ("cassie", True, [("frame", "pelvis", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, [1.0, 2.0, 0.5, 1.0, 1.0, 3.0])], None),
}


def _build(name, B, rows, seed=0):
    """(data routed by the product's default, data on the tree kernel, ...): the same problem twice."""
    import ik_amd
    model_name, ff, specs, cons = ROUTED[name]
    out = {}
    for label, val in (("static", str(rows)), ("tree", "0")):
        os.environ["IKGPU_TREE_STATIC_ROWS"] = val
        os.environ["IKGPU_TREE_STATIC_CONSTRAINED"] = "1"   # (the product leaves constrained tree problems on the tree kernel: it is faster there)
        try:
            ik, O, model, problem, data, om, ot, q0, tg = build(model_name, ff, specs, B, seed=seed)
            if cons:
                problem.add_frame_constraint("c", ik_amd.FrameConstraint.create(model, cons[0], ik_amd.KinematicType(cons[1])))
                data = ik_amd.dls_data(problem, device=0)
        finally:
            os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"
            os.environ.pop("IKGPU_TREE_STATIC_CONSTRAINED", None)
        out[label] = (ik, O, model, problem, data, om, ot, q0, tg)
    return out


@pytest.mark.parametrize("name", sorted(ROUTED))
def test_small_tree_problems_run_on_the_static_program_and_agree_with_the_tree_kernel_and_the_oracle(torch_cuda, name):
    torch = torch_cuda
    B = 4096 + 13
    both = _build(name, B, 12)
    ik, O, model, problem, data_s, om, ot, q0, tg = both["static"]
    data_t = both["tree"][4]
    if not data_s.kernel.endswith(",static>"):
        pytest.skip("hipRTC unavailable: %s" % data_s.kernel)
    assert data_s.kernel.startswith("dls_generic<") and data_t.kernel.startswith("dls_tree<"), (data_s.kernel, data_t.kernel)
    cons = ROUTED[name][3]
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    # (full steps at damping 1e-2 are chaotic on these task sets beyond a few iterations -- tests/test_gpu_generic.py
    # CHAOTIC_AT_FULL_STEP -- so the long runs take the demo's own damping and a half step)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (50, 1e-1, 0.5, -1.0), (100, 1e-1, 0.5, 1e-4)):
        p = ik.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        v = ik.inverse_kinematics_visitor(tol)
        Qs, oks, its = ik.dls_batch(problem, Q0, T, data_s, v, p)
        Qt, okt, itt = ik.dls_batch(both["tree"][3], Q0, T, data_t, v, p)
        prm = O.params(iters, damping, step, tol)
        cores = os.cpu_count() or 1
        if cons:
            oc = O.make_tasks([(model.getFrameId(cons[0]), 0, cons[1], 0, None)])

            def solve(ext):
                return O.dls_batch_constrained(om, ot, oc, tg, q0, prm, cores, ext=ext)
        else:
            def solve(ext):
                return O.dls_batch(om, ot, tg, q0, prm, cores, ext=ext)
        q_ref, ok_ref, it_ref = solve(None)
        q_x, _, _ = solve("q")                                   # the same oracle in _Float128 arithmetic
        same = its.cpu().numpy() == it_ref                     # (a stop decision within rounding of the tolerance may flip: rare)
        assert same.mean() > 0.999 and np.array_equal(oks.cpu().numpy()[same], ok_ref[same]), (name, iters)
        # These task sets have problems whose long runs amplify rounding (the double oracle itself is within 1e-6 of its own
        # _Float128 run on 99.7 % of them at 100 damped half steps): the device is held to being as close to the _Float128
        # trajectory as the double oracle is -- and to the bar against the oracle on the rest.
        d_gpu_x = np.abs(Qs.cpu().numpy().T - q_x).max(axis=1)
        d_ref_x = np.abs(q_ref - q_x).max(axis=1)
        frac_gpu, frac_ref = (d_gpu_x <= TOL).mean(), (d_ref_x <= TOL).mean()
        assert frac_gpu >= frac_ref - 0.002 and frac_gpu >= 0.99, (name, iters, frac_gpu, frac_ref)
        d = np.abs(Qs.cpu().numpy().T - q_ref).max(axis=1)
        # (the eliminated-posture program solves the system by another algebraic route -- Woodbury on D + J_f^T J_f -- than the oracle's
        # dense dual solve: same answer to cond x eps, ~2e-12 rad per step at damping 1e-2, as the tree kernel's primal arrow solve)
        median_bar = 1e-10 if name == "demo_with_posture" else 1e-12
        assert np.median(d) < median_bar and (d[(d_ref_x <= 1e-9) & same] <= TOL).mean() >= 0.9995, (name, iters, np.median(d))
        agree = (its == itt)
        assert agree.double().mean().item() > 0.999
        dt = (Qs - Qt).abs().max(dim=0).values[agree]
        # (two kernels that each differ from the oracle by their own rounding: the pair agrees a little less often than either does with it)
        assert (dt <= 1e-8).double().mean().item() >= frac_ref - (0.01 if name == "demo_with_posture" else 0.004), (name, iters, dt.max().item())
    print("%s: %s replaces %s" % (name, data_s.kernel, data_t.kernel))


@pytest.mark.parametrize("case", ["shared_joints", "moving_reference_prismatic", "com_of_the_arm", "demo_with_direction_in_pelvis_frame"])
def test_static_generic_program_full_size_properties(torch_cuda, case):
    """B = 65536 on the static program: results independent of the batch composition (a problem solved alone or in the batch gives
    the same bits), run-to-run identical, within the joint limits."""
    torch = torch_cuda
    name, ff, specs, edit = CASES[case]
    ik, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, 512, seed=5, xml_edit=edit)
    if not data.kernel.endswith(",static>"):
        pytest.skip("hipRTC unavailable: %s" % data.kernel)
    rep = 65536 // 512
    Q0 = torch.from_numpy(np.ascontiguousarray(np.tile(q0, (rep, 1)).T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(np.tile(tg, (rep, 1, 1)).transpose(1, 2, 0))).cuda()
    p = ik.dls_parameters(max_iterations=50, damping=1e-1, step_length=0.5)
    Q, ok, it = ik.dls_batch(problem, Q0, T, data, ik.never_stop_visitor(), p)
    Q2, _, _ = ik.dls_batch(problem, Q0, T, data, ik.never_stop_visitor(), p)
    assert torch.equal(Q, Q2)
    assert torch.equal(Q[:, :512], Q[:, 512 * 77:512 * 78])           # every copy of a problem gives the same bits
    Qs, _, _ = ik.dls_batch(problem, Q0[:, :512].contiguous(), T[:, :, :512].contiguous(), data, ik.never_stop_visitor(), p)
    assert torch.equal(Q[:, :512], Qs)
    lo, hi = torch.from_numpy(model.lowerPositionLimit).cuda(), torch.from_numpy(model.upperPositionLimit).cuda()
    assert (Q >= lo[:, None] - 1e-15).all() and (Q <= hi[:, None] + 1e-15).all()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("case", ["shared_joints", "demo_task_set", "com_of_the_arm"])
def test_static_program_lane_refill_is_bit_identical_to_lock_step(torch_cuda, monkeypatch, case, layout):
    """The stop-rule mode of a static lane program on a batch larger than the machine (generic_solver.hpp GenericRefill): a lane that
    is done stores its result and takes the next problem.  Same bits as the lock-step program -- q, success, iterations -- at sizes
    around the wave and machine boundaries, forced on at small sizes too (IKGPU_REFILL=1: tail lanes, waves that start empty)."""
    torch = torch_cuda
    name, ff, specs, edit = CASES[case]
    monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")   # (the demo task set plans onto the tree kernel otherwise)
    ik, O, model, problem, data, om, ot, q0, tg = build(name, ff, specs, 1024, seed=9, xml_edit=edit)
    monkeypatch.delenv("IKGPU_DLS_KERNEL")
    if not data.kernel.endswith(",static>"):
        pytest.skip("hipRTC unavailable: %s" % data.kernel)
    p = ik.dls_parameters(max_iterations=24, damping=1e-1, step_length=0.5)
    vis = ik.inverse_kinematics_visitor(1e-3)
    for B in (1, 63, 1000, 65536 + 64 * 3 + 5, 3 * 65536 + 17):
        rep = -(-B // 1024)
        q = np.tile(q0, (rep, 1))[:B]
        t = np.tile(tg, (rep, 1, 1))[:B]
        rng = np.random.default_rng(B)
        q = q + rng.uniform(-0.05, 0.05, q.shape) * (np.arange(B) % 7 != 0)[:, None]   # a spread of iteration counts, some lanes at their seed
        if layout == "soa":
            Q0 = torch.from_numpy(np.ascontiguousarray(q.T)).cuda()
            T = torch.from_numpy(np.ascontiguousarray(t.transpose(1, 2, 0))).cuda()
        else:
            Q0 = torch.from_numpy(np.ascontiguousarray(q)).cuda()
            T = torch.from_numpy(np.ascontiguousarray(t)).cuda()
        res = {}
        # "2": the two-phase solve (kernels.hpp run_two_phase: the lock-step program until a wave's stragglers are few, the refill twin
        # on the problems left open), also with the hand-over at once and late; None: the default policy for this batch size
        for mode, env in (("0", {}), ("1", {}), ("2", {}), ("2 at once", {"IKGPU_TWO_PHASE_ITERS": "1", "IKGPU_TWO_PHASE_ACTIVE": "63"}),
                          ("2 late", {"IKGPU_TWO_PHASE_ITERS": "12", "IKGPU_TWO_PHASE_ACTIVE": "2"}), (None, {})):
            if mode is not None:
                os.environ["IKGPU_REFILL"] = mode[0]
            os.environ.update(env)
            try:
                out = ik.dls_batch(problem, Q0, T, data, vis, p, layout=layout)
                torch.cuda.synchronize()
            finally:
                for k in ["IKGPU_REFILL"] + list(env):
                    os.environ.pop(k, None)
            res[mode] = [x.clone() for x in out]
        for mode in res:
            for a, b in zip(res["0"], res[mode]):
                assert torch.equal(a, b), (case, layout, B, mode)
        it = res["1"][2]
        assert int(it.max()) <= 24 and int(it.min()) >= 0
        if B >= 1000:
            assert it.unique().numel() > 2, "the workload should spread the iteration counts"
