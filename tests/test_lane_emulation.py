"""The exact per-lane program of the gfx950 chain kernels (ik_amd/csrc/device/*.hpp), compiled for the
host by this test and run lane after lane, against the C oracle.  This is how the lane program and the
host-side problem analysis are checked in the GPU-less build container; the -m gpu tests then check
the same program on the device through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, urdf_path

import oracle as O


@pytest.fixture(scope="module")
def emu(native_built):
    src = os.path.join(ROOT, "tests", "lane_emu", "lane_emu.cpp")
    out = os.path.join(ROOT, "tests", "lane_emu", "liblane_emu.so")
    deps = [src] + [os.path.join(ROOT, "ik_amd", "csrc", f) for f in
                    ("model.cpp", "problem.cpp", "model.hpp", "problem.hpp", "device/lane_math.hpp",
                     "device/chain_solver.hpp", "device/chain_kernel_body.hpp", "device/chain_hot.hpp", "device/tree_solver.hpp",
                     "device/tree_kernel_body.hpp", "device/lane_math.hpp", "device/generic_solver.hpp", "device/pik_solver.hpp", "device/coop_solver.hpp", "device/pik_coop.hpp", "generic_tables.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ik_amd", "csrc"), "-o", out, src,
                               os.path.join(ROOT, "ik_amd", "csrc", "model.cpp"), os.path.join(ROOT, "ik_amd", "csrc", "problem.cpp")])
    L = C.CDLL(out)
    L.lane_emu_last_error.restype = C.c_char_p
    return L


def run(L, urdf, task, mode, q0, tg, prm, nv, M, layout=1, root=0, ntasks=1):
    B, nq = (q0.shape[0], q0.shape[1]) if layout == 1 else (q0.shape[1], q0.shape[0])
    qo = np.empty_like(q0)
    ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
    e, J, oMf = np.empty((B, M)), np.empty((B, M, nv)), np.empty((B, ntasks, 12))
    p = lambda a: C.c_void_p(a.ctypes.data)
    tasks = task if ntasks > 1 else C.byref(task)
    rc = L.lane_emu_run(urdf, C.c_size_t(len(urdf)), root, tasks, ntasks, mode, C.c_int64(B), p(q0), p(tg),
                        C.byref(prm) if prm is not None else None, p(qo), p(ok), p(it), p(e), p(J), p(oMf), layout)
    assert rc == 0, L.lane_emu_last_error()
    return qo, ok, it, e, J, oMf


def setup(name, frame, ktype=2, weights=None, B=256, mode="near", narrow=None):
    import ik_amd
    from ik_amd import capi, workload
    urdf = open(urdf_path(name), "rb").read()
    model = ik_amd.Model.from_urdf_xml(urdf)
    om = O.OracleModel(model.flat())
    fid = model.getFrameId(frame)
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else workload.cassie_nominal(model.names)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, mode, narrow)
    tg = O.fk_batch(om, qs, [fid])
    w = list(weights) + [1.0] * (6 - len(weights)) if weights is not None else [1.0] * 6
    task = capi.Task(fid, 0, ktype, 0, (C.c_double * 6)(*w))
    ot = O.make_tasks([(fid, 0, ktype, 0, weights)])
    return urdf, model, om, task, ot, q0, qs, tg


@pytest.mark.parametrize("name,frame", [("cassie_fixed", "LeftFootFront"), ("ur5", "tool0"), ("cassie_fixed", "RightFootBack")])
def test_lane_program_stagewise(emu, name, frame):
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup(name, frame, B=200, mode="uniform", narrow=2.0 if name == "ur5" else None)
    *_, oMf = run(emu, urdf, task, 2, qs, tg, None, model.nv, 6)
    assert np.abs(oMf - tg).max() < 1e-14
    _, _, _, e, J, _ = run(emu, urdf, task, 1, q0, tg, None, model.nv, 6)
    for b in range(q0.shape[0]):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-11 and np.abs(J[b] - Jo).max() < 1e-11


@pytest.mark.parametrize("name,frame", [("cassie_fixed", "LeftFootFront"), ("ur5", "tool0")])
@pytest.mark.parametrize("iters,tol", [(50, -1.0), (100, 1e-4), (1, -1.0), (0, 1e-4)])
def test_lane_program_full_loop(emu, name, frame, iters, tol):
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup(name, frame, B=256, mode="near")
    q0[:, -1] += 9.0  # an entry outside the support (Cassie) / inside it (UR5), beyond its limit
    prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
    qo, ok, it, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6)
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
    assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
    assert np.abs(qo - q_ref).max() < 1e-9
    # SoA gives the same bits
    qo2, ok2, it2, *_ = run(emu, urdf, task, 0, np.ascontiguousarray(q0.T), np.ascontiguousarray(tg.transpose(1, 2, 0)),
                            prm, model.nv, 6, layout=0)
    assert np.array_equal(qo2.T, qo) and np.array_equal(ok2, ok) and np.array_equal(it2, it)


def test_device_sincos_accuracy(emu):
    """device/lane_math.hpp: dsincos (0), dsincos_fast (1: reduction by 2 pi, then) dsincos_bounded<D> (2, 3: the fdlibm kernels
    at x / 2^D and D angle doublings, valid on |x| <= 2^D pi/4 and a little beyond)."""
    p = lambda a: C.c_void_p(a.ctypes.data)
    for D, lim in ((2, np.pi + 0.05), (3, 2 * np.pi + 0.05), (0, 50.0), (1, 50.0), (1, 3000.0)):   # 1: dsincos_fast, any angle
        x = np.concatenate([np.linspace(-lim, lim, 400001), [0.0, -0.0, 1e-300, np.pi / 4, -np.pi / 2, min(lim, np.pi)]])
        s, c = np.empty_like(x), np.empty_like(x)
        emu.lane_emu_sincos(D, C.c_int64(x.size), p(x), p(s), p(c))
        assert np.abs(s - np.sin(x)).max() < 2e-15 and np.abs(c - np.cos(x)).max() < 2e-15, D
        assert np.abs(s * s + c * c - 1.0).max() < 5e-15
    # dsincos_hot (the headline loop): the reduction keeps one word of 2 pi, so its error grows with the angle, 3.9e-17 |x| -- less
    # than half an ulp of x; within +-2 pi (every joint limit of the fixture models) it meets the same 2e-15
    for lim, bar in ((2 * np.pi + 0.05, 2e-15), (50.0, 2e-15 + 50.0 * 4e-17), (3000.0, 2e-15 + 3000.0 * 4e-17)):
        x = np.concatenate([np.linspace(-lim, lim, 400001), [0.0, -0.0, 1e-300, np.pi / 4, -np.pi / 2, np.pi]])
        s, c = np.empty_like(x), np.empty_like(x)
        emu.lane_emu_sincos(4, C.c_int64(x.size), p(x), p(s), p(c))
        assert np.abs(s - np.sin(x)).max() < bar and np.abs(c - np.cos(x)).max() < bar, lim
    s0, c0 = np.empty(1), np.empty(1)
    emu.lane_emu_sincos(2, C.c_int64(1), p(np.zeros(1)), p(s0), p(c0))
    assert s0[0] == 0.0 and c0[0] == 1.0


def test_log6_front_ends_agree_up_to_a_rotation_by_pi(emu):
    """log6_and_jlog6_hot (one reciprocal, branch-free acos: the headline loop and the tree kernels) against log6_and_jlog6_inv (the
    general builds, which follow the oracle's formulas) over the whole range of the rotation angle -- and AT theta = pi to rounding, where
    the trace is <= -1 and (1 + cos theta) / 2 rounds to an exact zero: the hot front end then returned 1 / theta = 0 (beta, Jlog3's
    diagonal and the translation of log6 wrong; a DLS step off by a radian, one lane-step in 3e7 of tests/test_gpu_full_size.py's
    step-synchronised runs, seed 9).  Reference formulas: SURVEY.md App. A.3 (pinocchio log6 / Jlog6; ik/ik/frame.hpp:50-61,162-166)."""
    rng = np.random.default_rng(5)
    n = 20000
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    theta = np.concatenate([rng.uniform(0.0, np.pi, n - 6000), np.pi - 10.0 ** rng.uniform(-16, -2, 3000), np.full(3000, np.pi)])
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -axis[:, 2], axis[:, 1], axis[:, 2], -axis[:, 0], -axis[:, 1], axis[:, 0]
    R = np.eye(3)[None] + np.sin(theta)[:, None, None] * K + (1.0 - np.cos(theta))[:, None, None] * (K @ K)
    R[-1500:, 0, 0] -= 4.5e-16           # the last rotations by pi: the trace pushed to and below -1 (x clamps to -1 exactly)
    pe = rng.uniform(-0.5, 0.5, (n, 3))
    out = [np.empty((n, 24)), np.empty((n, 24))]
    p = lambda a: C.c_void_p(a.ctypes.data)
    Rc, pc = np.ascontiguousarray(R.reshape(n, 9)), np.ascontiguousarray(pe)
    for which in (0, 1):
        emu.lane_emu_log6(which, C.c_int64(n), p(Rc), p(pc), p(out[which]))
    assert np.isfinite(out[0]).all() and np.isfinite(out[1]).all()
    d = np.abs(out[1] - out[0]).max(axis=1)
    # both evaluate sin(theta) from 1 + cos(theta), whose cancellation near pi is the formulas' own: eps / (pi - theta) on the terms it
    # feeds (all O(pi - theta) themselves, so the ABSOLUTE difference stays at rounding level)
    assert d.max() < 1e-12, (d.max(), theta[np.argmax(d)], np.trace(R[np.argmax(d)]))
    assert (np.trace(R[-1500:], axis1=1, axis2=2) <= -1.0).sum() > 500      # the regime the bug lived in was exercised
    assert np.abs(np.linalg.norm(out[1][-3000:, 3:6], axis=1) - np.pi).max() < 1e-7


@pytest.mark.parametrize("name,frame", [("cassie_fixed", "LeftFootFront"), ("ur5", "tool0")])
@pytest.mark.parametrize("iters,tol", [(50, -1.0), (100, 1e-4), (1, -1.0)])
def test_lane_program_device_general_build(emu, monkeypatch, name, frame, iters, tol):
    """The device's general chain build (SMASK = 0: compile-time "skip nothing", sin / cos by dsincos_fast -- reduction by
    2 pi and angle doubling), also from configurations more than a turn away from the joint limits."""
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup(name, frame, B=192, mode="near")
    q0[::7, 2] += 7.0          # the first evaluation is at q0 as given (the reference does not clamp it)
    prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
    monkeypatch.setenv("LANE_EMU_TRIG", "0")
    qo, ok, it, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6)
    monkeypatch.delenv("LANE_EMU_TRIG")
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
    assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
    assert np.abs(qo - q_ref).max() < 1e-9


@pytest.mark.parametrize("name,frame", [("cassie_fixed", "LeftFootFront"), ("cassie_fixed", "RightFootFront"), ("ur5", "tool0"), ("ur10", "tool0")])
@pytest.mark.parametrize("mode,iters,tol", [("near", 50, -1.0), ("near", 100, 1e-4), ("near", 1, -1.0), ("near", 0, 1e-4),
                                            ("uniform", 1, -1.0), ("uniform", 3, -1.0)])
def test_lane_program_structure_specialised_build(emu, monkeypatch, name, frame, mode, iters, tol):
    """The structure-specialised chain program (device/chain_hot.hpp; the headline kernel of kernels_hot.hip): literal zeros and
    ones for the structural placement entries, the compact table, the one-reciprocal log6 front end with the branch-free acos.
    "uniform" targets are far from the start: rotation errors up to pi (every acos range, the theta -> pi formula of log3)."""
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup(name, frame, B=512, mode=mode, narrow=2.0 if name.startswith("ur") and mode == "uniform" else None)
    q0[::7, 2] += 7.0          # the first evaluation is at q0 as given (the reference does not clamp it)
    prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
    monkeypatch.setenv("LANE_EMU_HOT", "1")
    qo, ok, it, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6)
    qo2, ok2, it2, *_ = run(emu, urdf, task, 0, np.ascontiguousarray(q0.T), np.ascontiguousarray(tg.transpose(1, 2, 0)), prm, model.nv, 6, layout=0)
    monkeypatch.delenv("LANE_EMU_HOT")
    qg, okg, itg, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6)          # the general chain program
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
    assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
    bar = 1e-9 if mode == "near" else (1e-9 if iters == 1 else 1e-6)
    assert np.abs(qo - q_ref).max() < bar
    assert np.abs(qo - qg).max() < bar
    assert np.array_equal(qo2.T, qo) and np.array_equal(ok2, ok) and np.array_equal(it2, it)


@pytest.mark.parametrize("ktype,weights", [(0, None), (1, None), (2, [1.0, 2.0, 0.5, 1.5, 1.0, 3.0]), (1, [2.0, 1.0, 0.25])])
def test_lane_program_types_and_weights(emu, ktype, weights):
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup("cassie_fixed", "LeftFootFront", ktype, weights, B=128)
    M = 6 if ktype == 2 else 3
    _, _, _, e, J, _ = run(emu, urdf, task, 1, q0, tg, None, model.nv, M)
    eo, Jo = O.evaluate(om, ot, tg[5], q0[5])
    assert np.abs(e[5] - eo).max() < 1e-12 and np.abs(J[5] - Jo).max() < 1e-12
    for iters in (1, 25):
        prm = capi.DlsParams(iters, 1e-2, 1.0, -1.0)
        qo, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, M)
        q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, -1.0))
        assert np.abs(qo - q_ref).max() < 1e-8


def test_lane_program_general_axis_and_far_targets_stepwise(emu):
    """A joint about a non-aligned axis (folded into the placements on the host) and UR5 far targets,
    where only step-wise parity is meaningful (the iteration is chaotic there)."""
    from ik_amd import capi
    import ik_amd
    xml = open(urdf_path("ur5"), "rb").read().replace(b'<axis xyz="0 1 0"/>', b'<axis xyz="0.6 0.64 0.48"/>', 1)
    model = ik_amd.Model.from_urdf_xml(xml)
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("tool0")
    rng = np.random.default_rng(0)
    q0 = rng.uniform(-2, 2, (200, 6))
    tg = O.fk_batch(om, rng.uniform(-2, 2, (200, 6)), [fid])
    task = capi.Task(fid, 0, 2, 0, (C.c_double * 6)(*[1.0] * 6))
    ot = O.make_tasks([(fid, 0, 2, 0, None)])
    for iters, bar in ((1, 1e-9), (3, 1e-6)):
        qo, *_ = run(emu, xml, task, 0, q0, tg, capi.DlsParams(iters, 1e-2, 1.0, -1.0), 6, 6)
        q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, -1.0))
        assert np.abs(qo - q_ref).max() < bar


# ---------------------------------------------------------------------------------------------------
# free-flyer tree program (shape F: Cassie full body, LeftFootFront + RightFootFront + pelvis)
# ---------------------------------------------------------------------------------------------------
def setup_full_body(B, types=(2, 2, 2), weights=(None, None, None), prios=(0, 0, 0), frames=("LeftFootFront", "RightFootFront", "pelvis")):
    import ik_amd
    from ik_amd import capi, workload
    urdf = open(urdf_path("cassie"), "rb").read()
    model = ik_amd.Model.from_urdf_xml(urdf, free_flyer=True)
    om = O.OracleModel(model.flat())
    fids = [model.getFrameId(f) for f in frames]
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                         workload.cassie_nominal(model.names), np.arange(B), seed=0,
                                         integrate=lambda q, v: O.integrate(om, q, v))
    tg = O.fk_batch(om, qs, fids)
    tasks = (capi.Task * len(fids))()
    spec = []
    for i, f in enumerate(fids):
        w = weights[i]
        ww = list(w) + [1.0] * (6 - len(w)) if w is not None else [1.0] * 6
        tasks[i] = capi.Task(f, 0, types[i], prios[i], (C.c_double * 6)(*ww))
        spec.append((f, 0, types[i], prios[i], w))
    return urdf, model, om, tasks, O.make_tasks(spec), q0, qs, tg, fids


def test_tree_program_stagewise(emu):
    urdf, model, om, tasks, ot, q0, qs, tg, fids = setup_full_body(48)
    *_, oMf = run(emu, urdf, tasks, 2, qs, tg, None, model.nv, 18, root=1, ntasks=3)
    assert np.abs(oMf - tg).max() < 1e-14
    _, _, _, e, J, _ = run(emu, urdf, tasks, 1, q0, tg, None, model.nv, 18, root=1, ntasks=3)
    for b in range(q0.shape[0]):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-12 and np.abs(J[b] - Jo).max() < 1e-12


@pytest.mark.parametrize("iters,tol", [(1, -1.0), (3, -1.0), (50, -1.0), (100, 1e-4), (0, 1e-4)])
def test_tree_program_full_loop(emu, iters, tol):
    """The arrow-structured primal solve (7x7 Cholesky per chain + 6x6 Schur complement on the base) returns
    the dq of the reference's dense 18x18 dual solve; the base moves by SE(3) integration (App. A.5)."""
    from ik_amd import capi
    urdf, model, om, tasks, ot, q0, qs, tg, fids = setup_full_body(64)
    q0[:, 13] += 4.0  # LeftAchillesSpring: outside every support, beyond its limit -> clamped iff a step is taken
    prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
    qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, 18, root=1, ntasks=3)
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
    assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
    assert np.abs(qo - q_ref).max() < 1e-9
    assert np.abs(np.linalg.norm(qo[:, 3:7], axis=1) - 1).max() < 1e-12


@pytest.mark.parametrize("iters,tol", [(1, -1.0), (50, -1.0), (100, 1e-4)])
def test_tree_program_device_general_build(emu, monkeypatch, iters, tol):
    """The device's general tree build (SPEC = 0: compile-time "skip nothing", sin / cos by dsincos_fast), also from
    configurations several turns away from the joint limits."""
    from ik_amd import capi
    urdf, model, om, tasks, ot, q0, qs, tg, fids = setup_full_body(40)
    q0[::5, 9] += 11.0         # LeftHipPitch far outside its limits at the first evaluation (q0 is not clamped before it)
    prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
    monkeypatch.setenv("LANE_EMU_TRIG", "0")
    qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, 18, root=1, ntasks=3)
    monkeypatch.delenv("LANE_EMU_TRIG")
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
    assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
    assert np.abs(qo - q_ref).max() < 1e-9


def test_tree_program_types_weights_priorities(emu):
    """Position / Orientation tasks are Full tasks with zero-weight rows in the primal form; the stop test
    reads priority-0 rows only (reference ik/ik/visitor.hpp:19)."""
    from ik_amd import capi
    urdf, model, om, tasks, ot, q0, qs, tg, fids = setup_full_body(
        32, types=(0, 2, 1), weights=([2.0, 1.0, 0.5], [1, 1, 1, 0.3, 0.3, 0.3], None), prios=(0, 0, 1))
    M = 3 + 6 + 3
    _, _, _, e, J, _ = run(emu, urdf, tasks, 1, q0, tg, None, model.nv, M, root=1, ntasks=3)
    eo, Jo = O.evaluate(om, ot, tg[3], q0[3])
    assert np.abs(e[3] - eo).max() < 1e-12 and np.abs(J[3] - Jo).max() < 1e-12
    for iters, tol in ((1, -1.0), (30, 1e-4)):
        prm = capi.DlsParams(iters, 1e-2, 1.0, tol)
        qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=1, ntasks=3)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref) and np.abs(qo - q_ref).max() < 1e-8


def test_tree_program_single_chain_plus_base_task(emu):
    from ik_amd import capi
    urdf, model, om, tasks, ot, q0, qs, tg, fids = setup_full_body(32, types=(2, 2), weights=(None, None), prios=(0, 0),
                                                                     frames=("RightFootBack", "pelvis"))
    prm = capi.DlsParams(20, 1e-2, 1.0, -1.0)
    qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, 12, root=1, ntasks=2)
    q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(20, 1e-2, 1.0, -1.0))
    assert np.abs(qo - q_ref).max() < 1e-9


# ---------------------------------------------------------------------------------------------------
# generic fallback program: any tree, any task list (incl. AlignAxisTask rows, moving reference frames,
# prismatic joints), workspace in memory
# ---------------------------------------------------------------------------------------------------
def _generic_case(name, ff, specs, B, seed=0, xml_edit=None):
    """specs: list of (frame, reference, type, priority, weights)."""
    import ik_amd
    from ik_amd import capi, workload
    urdf = open(urdf_path(name), "rb").read()
    if xml_edit:
        urdf = xml_edit(urdf)
    model = ik_amd.Model.from_urdf_xml(urdf, free_flyer=ff)
    om = O.OracleModel(model.flat())
    rng = np.random.default_rng(seed)
    lo, hi = np.maximum(model.lowerPositionLimit, -2.5), np.minimum(model.upperPositionLimit, 2.5)
    if ff:
        nom = workload.cassie_nominal(model.names)
        q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(B), seed=seed)
    else:
        mid = 0.5 * (lo + hi)
        q0 = np.clip(mid + rng.uniform(-0.2, 0.2, (B, model.nq)), lo, hi)
        qs = np.clip(q0 + rng.uniform(-0.15, 0.15, (B, model.nq)), lo, hi)
    specs = expand_posture(specs, model.nq, model.nv)
    tasks = (capi.Task * len(specs))()
    ospec, fids, rids = [], [], []
    for i, (f, r, t, p, w) in enumerate(specs):
        fid, rid = (f, r) if t == 6 else ((0, model.getFrameId(r)) if t == 7 else (model.getFrameId(f), model.getFrameId(r)))
        ww = list(w) + [1.0] * (6 - len(w)) if w is not None else [1.0] * 6
        tasks[i] = capi.Task(fid, rid, t, p, (C.c_double * 6)(*ww))
        ospec.append((fid, rid, t, p, w))
        fids.append(fid)
        rids.append(rid)
    # targets: pose of the frame at q* expressed in the reference frame at q* (reachable); direction for align rows
    tg = np.zeros((B, len(specs), 12))
    for b in range(B):
        _, oMf = O.fk(om, qs[b])
        for i, (f, r, t, p, w) in enumerate(specs):
            if t == 7:   # centre of mass at q*, seen from the reference frame: the error of the task with a zero target
                tg[b, i, 9:] = O.evaluate(om, O.make_tasks([(0, rids[i], 7, 0, None)]), np.zeros((1, 12)), qs[b])[0]
            elif t == 6:
                tg[b, i, 9] = qs[b, rids[i]]
            elif t >= 3:
                tg[b, i, 9:] = rng.normal(size=3)
                tg[b, i, :9] = np.eye(3).ravel()
            else:
                Mr, Mf = T4(oMf[rids[i]]), T4(oMf[fids[i]])
                rel = np.linalg.inv(Mr) @ Mf
                tg[b, i] = np.concatenate([rel[:3, :3].ravel(), rel[:3, 3]])
    M = sum(6 if t == 2 else (1 if 3 <= t <= 6 else 3) for _, _, t, _, _ in specs)
    return urdf, model, om, tasks, O.make_tasks(ospec), q0, tg, M


def expand_posture(specs, nq, nv):
    """("@posture", nj, 6, priority, (weights[nj], mask[nj])) -> nj IKGPU_POSTURE_ROW specs (tangent column, q index)."""
    out = []
    for f, r, t, p, w in specs:
        if f == "@posture":
            out += [(nv - r + k, nq - r + k, 6, p, [w[0][k], w[1][k]]) for k in range(r)]
        else:
            out.append((f, r, t, p, w))
    return out


def T4(m12):
    M = np.eye(4)
    M[:3, :3] = np.asarray(m12[:9]).reshape(3, 3)
    M[:3, 3] = m12[9:]
    return M


def _prismatic_elbow(xml):
    return xml.replace(b'<joint name="elbow_joint" type="revolute">', b'<joint name="elbow_joint" type="prismatic">', 1) \
              .replace(b'lower="-3.14159265359" upper="3.14159265359"', b'lower="-0.2" upper="0.2"', 1)


GENERIC_CASES = {
    "leg_forced": ("cassie_fixed", False, [("LeftFootFront", "universe", 2, 0, None)], 2, None),
    "full_body_forced": ("cassie", True, [("LeftFootFront", "universe", 2, 0, None), ("RightFootFront", "universe", 2, 0, None),
                                          ("pelvis", "universe", 2, 0, None)], 3, None),
    # the demo's own task set (reference ik_ros/src/cassie.cpp:45-81): foot position w.r.t. the moving pelvis,
    # pelvis pose in the world, foot Y axis aligned with a direction
    "demo": ("cassie", True, [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None),
                              ("LeftFootFront", "universe", 4, 0, None)], 1, None),
    "fixed_two_feet_prio": ("cassie_fixed", False, [("LeftFootFront", "universe", 2, 0, None),
                                                    ("RightFootBack", "universe", 0, 1, [2.0, 1.0, 0.5])], 0, None),
    "shared_joints": ("ur5", False, [("tool0", "universe", 0, 0, None), ("forearm_link", "universe", 1, 0, None)], 0, None),
    "moving_reference_prismatic": ("ur5", False, [("tool0", "upper_arm_link", 2, 0, None)], 0, _prismatic_elbow),
    # ik::PostureTask (reference ik/ik/posture.hpp:17-85) as a regulariser at priority 1, weights and a mask with a hole
    "posture_regulariser": ("cassie", True, [("LeftFootFront", "universe", 2, 0, None), ("pelvis", "universe", 2, 0, None),
                                             ("@posture", 16, 6, 1, ([0.1 + 0.05 * k for k in range(16)],
                                                                     [0.0 if k in (3, 9) else 1.0 for k in range(16)]))], 1, None),
    # ik::CentreOfMassTask (reference ik/ik/centre_of_mass.hpp:14-62): the demo's commented-out balance task
    # (ik_ros/src/cassie.cpp:58,79,101) at priority 1 under the two foot poses; and seen from a moving frame, weighted
    "com_under_feet": ("cassie", True, [("LeftFootFront", "universe", 2, 0, None), ("RightFootFront", "universe", 2, 0, None),
                                        ("@com", "universe", 7, 1, None)], 1, None),
    "com_in_foot_frame": ("cassie_fixed", False, [("LeftFootFront", "universe", 0, 0, None),
                                                  ("@com", "LeftFootFront", 7, 0, [1.0, 2.0, 0.5])], 0, None),
    "com_of_the_arm": ("ur5", False, [("@com", "universe", 7, 0, None)], 0, None),
    "posture_only_arm": ("ur5", False, [("@posture", 4, 6, 0, ([1.0] * 4, [1.0] * 4)), ("tool0", "universe", 0, 0, None)], 0, None),
}


@pytest.mark.parametrize("case", sorted(GENERIC_CASES))
def test_generic_program_matches_oracle(emu, case):
    from ik_amd import capi
    name, ff, specs, root, edit = GENERIC_CASES[case]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, ff, specs, B, xml_edit=edit)
    nt = len(tasks)
    tk = tasks if nt > 1 else tasks[0]
    _, _, _, e, J, _ = run(emu, urdf, tk, 1, q0, tg, None, model.nv, M, root=root | 2 if "forced" in case else root, ntasks=nt)
    for b in range(B):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-11 and np.abs(J[b] - Jo).max() < 1e-11
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (40, 1e-1, 0.5, 1e-6)):
        prm = capi.DlsParams(iters, damping, step, tol)
        qo, ok, it, *_ = run(emu, urdf, tk, 0, q0, tg, prm, model.nv, M, root=root | 2 if "forced" in case else root, ntasks=nt)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())


def test_generic_and_specialised_programs_agree(emu):
    """The same problem through the register-resident chain program and through the generic one."""
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup("cassie_fixed", "LeftFootFront", B=64)
    prm = capi.DlsParams(50, 1e-2, 1.0, -1.0)
    qa, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6)
    qb, *_ = run(emu, urdf, task, 0, q0, tg, prm, model.nv, 6, root=2)
    assert np.abs(qa - qb).max() < 1e-9


# ---------------------------------------------------------------------------------------------------
# ik::pik -- prioritised IK on the generic lane program (device/pik_solver.hpp vs oracle iko_pik)
# ---------------------------------------------------------------------------------------------------
def run_pik(L, urdf, tasks, q0, tg, prm, root=0):
    B = q0.shape[0]
    qo = np.empty_like(q0)
    ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.lane_emu_pik(urdf, C.c_size_t(len(urdf)), root, tasks, len(tasks), C.c_int64(B), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1)
    assert rc == 0, L.lane_emu_last_error()
    return qo, ok, it


def pik_prm(iters, step, tol, lam, da=None):
    from ik_amd import capi
    prm = capi.PikParams()
    prm.max_iterations, prm.step_length, prm.stop_sq_tol, prm.num_levels = iters, step, tol, len(lam)
    for i, l in enumerate(lam):
        prm.lam[i] = l
    if da is not None:
        prm._da = (C.c_double * len(da))(*da)
        prm.da = C.cast(prm._da, C.POINTER(C.c_double))
    return prm


# Last field: is the projector left by the LAST level well determined?  When the projected Jacobian of a level is exactly
# rank deficient, its surplus singular values are rounding noise of a few eps * cond(previous levels), the same size as the
# rank threshold the reference inherits from Eigen (eps * min(rows, cols)): whether a noise direction is removed from P is
# then decided by rounding, in the reference as much as here.  dq before `P da` does not depend on it (a noise direction
# carries a factor sigma / (lambda^2 + sigma^2) ~ 1e-13), so those cases are compared with da = 0 only.
PIK_CASES = {
    # position first, orientation in its null space (6-DoF arm: level 1 gets the 3 remaining dof)
    "ur5_pos_then_ori": ("ur5", False, [("tool0", "universe", 0, 0, None), ("tool0", "universe", 1, 1, None)], 0, None, True),
    # a full pose exhausts the arm: level 1 sees a numerically-zero projected Jacobian
    "ur5_full_then_elbow": ("ur5", False, [("tool0", "universe", 2, 0, None), ("forearm_link", "universe", 0, 1, None)], 0, None, False),
    "fixed_two_feet": ("cassie_fixed", False, [("LeftFootFront", "universe", 2, 0, None),
                                               ("RightFootFront", "universe", 0, 1, [1.0, 2.0, 0.5])], 0, None, True),
    # the demo's tasks split over two levels, then a posture regulariser at the third
    "demo_three_levels": ("cassie", True, [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None),
                                           ("LeftFootFront", "universe", 4, 1, None),
                                           ("@posture", 16, 6, 2, ([0.5] * 16, [1.0] * 16))], 1, None, False),
    "feet_then_pelvis": ("cassie", True, [("LeftFootFront", "universe", 2, 0, None), ("RightFootFront", "universe", 2, 0, None),
                                          ("pelvis", "universe", 2, 1, None)], 1, None, False),
    "single_level_prismatic": ("ur5", False, [("tool0", "upper_arm_link", 2, 0, None)], 0, _prismatic_elbow, True),
}


@pytest.mark.parametrize("case", sorted(PIK_CASES))
def test_pik_program_matches_oracle(emu, case):
    name, ff, specs, root, edit, projector_determined = PIK_CASES[case]
    B = 16
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, ff, specs, B, seed=3, xml_edit=edit)
    levels = max(t.priority for t in tasks) + 1
    for iters, step, tol, lam, da in ((1, 1.0, -1.0, [1.0] * levels, None),
                                      (4, 1.0, -1.0, [0.1] * levels, None),
                                      (30, 0.5, 1e-8, [0.05, 0.1, 0.2][:levels], None),
                                      (6, 1.0, -1.0, [0.1] * levels, list(0.01 * np.cos(np.arange(model.nv))))):
        if da is not None and not projector_determined:
            continue
        qo, ok, it = run_pik(emu, urdf, tasks, q0, tg, pik_prm(iters, step, tol, lam, da), root=root)
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, da))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())


def test_pik_with_one_level_and_small_lambda_is_dls(emu):
    """With one priority level pik's step is J^T (J J^T + lambda^2 I)^-1 e -- ik::dls with damping = lambda."""
    from ik_amd import capi
    urdf, model, om, task, ot, q0, qs, tg = setup("cassie_fixed", "LeftFootFront", B=32)
    tasks = (capi.Task * 1)(task)
    qa, oka, ita = run_pik(emu, urdf, tasks, q0, tg, pik_prm(25, 1.0, 1e-10, [1e-2]))
    qb, okb, itb, *_ = run(emu, urdf, task, 0, q0, tg, capi.DlsParams(25, 1e-2, 1.0, 1e-10), model.nv, 6)
    assert np.array_equal(oka, okb) and np.array_equal(ita, itb)
    assert np.abs(qa - qb).max() < 1e-9


# ---------------------------------------------------------------------------------------------------
# ik::dls with ik::FrameConstraint rows: the step is projected into the null space of the constraint Jacobian
# ---------------------------------------------------------------------------------------------------
CONSTRAINT_CASES = {
    # the demo's commented-out intent (reference ik_ros/src/cassie.cpp:49-51,74-75): keep the right foot where it is
    "demo_right_foot_pinned": ("cassie", True, [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None)],
                               [("RightFootFront", "universe", 0)]),
    "pelvis_with_both_feet_locked": ("cassie", True, [("pelvis", "universe", 2, 0, None)],
                                     [("RightFootFront", "universe", 2), ("LeftFootFront", "RightFootFront", 0)]),
    "arm_keeps_tool_orientation": ("ur5", False, [("tool0", "universe", 0, 0, None)], [("tool0", "universe", 1)]),
    "relative_orientation_between_feet": ("cassie_fixed", False, [("LeftFootFront", "universe", 2, 0, None)],
                                          [("RightFootFront", "LeftFootBack", 1)]),
}


@pytest.mark.parametrize("foot_type", [0, 2])
@pytest.mark.parametrize("lam", [(0.1, 0.1), (1e-2, 0.5), (1.0, 1.0)])
def test_two_level_pik_on_the_tree_program(emu, monkeypatch, foot_type, lam):
    """ik::pik with two priority levels in the shape the tree kernel takes (device/tree_solver.hpp PikRow): level 0 = the foot
    task (Position w.r.t. the pelvis, or Full) and the pelvis pose, level 1 = the AlignAxisTask row; against the oracle's ik::pik
    (SVD damped pseudo-inverse + COD projector, reference ik/ik/pik.cpp:31-103)."""
    from ik_amd import capi
    specs = [("LeftFootFront", "pelvis", foot_type, 0, None), ("pelvis", "universe", 2, 0, None), ("LeftFootFront", "universe", 4, 1, None)]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, specs, B, seed=17)
    p = lambda a: C.c_void_p(a.ctypes.data)
    for iters, step, tol in ((1, 1.0, -1.0), (5, 1.0, -1.0), (50, 0.5, 1e-7), (100, 1.0, 1e-4)):
        qo = np.empty_like(q0)
        ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        prm = capi.DlsParams(iters, lam[0], step, tol)
        e, J, oMf = np.empty((B, M)), np.empty((B, M, model.nv)), np.empty((B, len(tasks), 12))
        monkeypatch.setenv("LANE_EMU_TREE_PIK_LAMBDA1", repr(lam[1]))
        rc = emu.lane_emu_run(urdf, C.c_size_t(len(urdf)), 1, tasks, len(tasks), 0, C.c_int64(B), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it),
                              p(e), p(J), p(oMf), 1)
        monkeypatch.delenv("LANE_EMU_TREE_PIK_LAMBDA1")
        assert rc == 0, emu.lane_emu_last_error()
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, list(lam)))
        # (full steps towards random alignment directions make a few lanes chaotic: the oracle's own answer moves by 0.7 rad
        # under a 1e-13 perturbation of q0 there -- those lanes have no answer to compare)
        q_pert, _, _ = O.pik_batch(om, ot, tg, q0 + 1e-13, O.pik_params(iters, step, tol, list(lam)))
        stable = np.abs(q_pert - q_ref).max(axis=1) < 1e-9
        assert stable.sum() >= B - 4
        assert np.array_equal(ok[stable], ok_ref[stable]) and np.array_equal(it[stable], it_ref[stable]), (foot_type, lam, iters)
        assert np.abs(qo - q_ref)[stable].max() < 1e-8, (foot_type, lam, iters, np.abs(qo - q_ref)[stable].max())
    # and it is not the one-level answer (the alignment row in level 0)
    q_dls, _, _ = O.dls_batch(om, O.make_tasks([(t.frame, t.reference, t.type, 0, None) for t in tasks]), tg, q0, O.params(100, lam[0], 1.0, 1e-4))
    assert np.abs(q_dls - q_ref)[stable].max() > 1e-6


@pytest.mark.parametrize("ctype", [0, 1, 2])
@pytest.mark.parametrize("with_align,with_posture", [(False, False), (True, False), (True, True)])
def test_constraint_build_of_the_tree_program(emu, ctype, with_align, with_posture):
    """One FrameConstraint (Position / Orientation / Full, reference = the universe) on the foot of the leg that carries no task
    -- the pinned stance foot -- next to the demo's tasks: the tree kernel's constraint build (device/tree_solver.hpp
    constraint_project: world rows of the constraint Jacobian on base + chain columns, Gram-Schmidt applied twice, dq -= V^T V dq)
    against the oracle's dense N = I - pinv(Jc) Jc (reference ik/ik/dls.cpp:26-34,43-53)."""
    from ik_amd import capi
    specs = [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None)]
    if with_align:
        specs.append(("LeftFootFront", "universe", 4, 0, None))        # AlignAxisTask, frame Y axis
    if with_posture:   # the demo's PostureTask on all sixteen joints: the constrained chain's joints carry no task, so their posture
                       # rows are 1 x 1 systems whose steps the projection then acts on
        specs.append(("@posture", 16, 6, 0, ([0.3 + 0.04 * k for k in range(16)], [1.0] * 16)))
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, specs, B, seed=13)
    cons = (capi.Task * 1)(capi.Task(model.getFrameId("RightFootFront"), 0, ctype, 0, (C.c_double * 6)(*[1.0] * 6)))
    oc = O.make_tasks([(model.getFrameId("RightFootFront"), 0, ctype, 0, None)])
    buf = C.create_string_buffer(200)
    rc = capi.lib().ikgpu_problem_plan_constrained(ik_amd_model_handle(model), tasks, len(tasks), cons, 1, buf, len(buf))
    assert rc == 0 and buf.value.decode().startswith("dls_tree<NJ=7,chains=1") and "constraint_rows=%d" % (6 if ctype == 2 else 3) in buf.value.decode(), buf.value
    assert ("posture" in buf.value.decode()) == with_posture
    p = lambda a: C.c_void_p(a.ctypes.data)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (5, 1e-2, 1.0, -1.0), (40, 1e-1, 0.5, 1e-7), (100, 1e-1, 1e-1, 1e-4)):
        qo = np.empty_like(q0)
        ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        prm = capi.DlsParams(iters, damping, step, tol)
        rc = emu.lane_emu_dls_constrained(urdf, C.c_size_t(len(urdf)), 1 | 4, tasks, len(tasks), cons, 1, C.c_int64(B),
                                          p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1)
        assert rc == 0, emu.lane_emu_last_error()
        q_ref, ok_ref, it_ref = O.dls_batch_constrained(om, ot, oc, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (ctype, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (ctype, iters, np.abs(qo - q_ref).max())


def ik_amd_model_handle(model):
    return model._h


@pytest.mark.parametrize("case", sorted(CONSTRAINT_CASES))
def test_constrained_dls_program_matches_oracle(emu, case):
    from ik_amd import capi
    name, ff, specs, cspecs = CONSTRAINT_CASES[case]
    B = 16
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, ff, specs, B, seed=9)
    cons = (capi.Task * len(cspecs))()
    for i, (f, r, t) in enumerate(cspecs):
        cons[i] = capi.Task(model.getFrameId(f), model.getFrameId(r), t, 0, (C.c_double * 6)(*[1.0] * 6))
    oc = O.make_tasks([(model.getFrameId(f), model.getFrameId(r), t, 0, None) for f, r, t in cspecs])
    p = lambda a: C.c_void_p(a.ctypes.data)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (5, 1e-2, 1.0, -1.0), (40, 1e-1, 0.5, 1e-7)):
        qo = np.empty_like(q0)
        ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        prm = capi.DlsParams(iters, damping, step, tol)
        rc = emu.lane_emu_dls_constrained(urdf, C.c_size_t(len(urdf)), 1 if ff else 0, tasks, len(tasks), cons, len(cons), C.c_int64(B),
                                          p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1)
        assert rc == 0, emu.lane_emu_last_error()
        q_ref, ok_ref, it_ref = O.dls_batch_constrained(om, ot, oc, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())
        # the cooperative program (device/coop_solver.hpp: pivoted Gram-Schmidt instead of Jacobi for the projector)
        qc, okc, itc = np.empty_like(q0), np.zeros(B, np.uint8), np.zeros(B, np.int32)
        rc = emu.lane_emu_dls_coop_constrained(urdf, C.c_size_t(len(urdf)), 1 if ff else 0, tasks, len(tasks), cons, len(cons), C.c_int64(B),
                                               p(q0), p(tg), C.byref(prm), p(qc), p(okc), p(itc), 1)
        assert rc == 0, emu.lane_emu_last_error()
        assert np.array_equal(okc, ok_ref) and np.array_equal(itc, it_ref), (case, iters)
        assert np.abs(qc - q_ref).max() < 1e-8, (case, iters, np.abs(qc - q_ref).max())
    # and the constraint bites: without it the same problem goes elsewhere
    q_free, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
    assert np.abs(q_free - q_ref).max() > 1e-3


# ---------------------------------------------------------------------------------------------------
# the cooperative (16 lanes per problem, LDS-resident) form of the generic DLS program, device/coop_solver.hpp
# ---------------------------------------------------------------------------------------------------
COOP_CASES = sorted(GENERIC_CASES)


@pytest.mark.parametrize("case", COOP_CASES)
def test_cooperative_program_matches_oracle(emu, case):
    from ik_amd import capi
    name, ff, specs, root, edit = GENERIC_CASES[case]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, ff, specs, B, xml_edit=edit)
    p = lambda a: C.c_void_p(a.ctypes.data)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (40, 1e-1, 0.5, 1e-6)):
        qo = np.empty_like(q0)
        ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        prm = capi.DlsParams(iters, damping, step, tol)
        rc = emu.lane_emu_dls_coop(urdf, C.c_size_t(len(urdf)), root & 1, tasks, len(tasks), C.c_int64(B), p(q0), p(tg), C.byref(prm),
                                   p(qo), p(ok), p(it), 1)
        if rc == 2:
            pytest.skip("four workspaces of this problem do not fit 64 KB of LDS: the per-lane program runs it")
        assert rc == 0, emu.lane_emu_last_error()
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())
    # structure-of-arrays inputs give the same bits
    qs, oks, its = np.empty((q0.shape[1], B)), np.zeros(B, np.uint8), np.zeros(B, np.int32)
    q0s, tgs = np.ascontiguousarray(q0.T), np.ascontiguousarray(tg.transpose(1, 2, 0))
    rc = emu.lane_emu_dls_coop(urdf, C.c_size_t(len(urdf)), root & 1, tasks, len(tasks), C.c_int64(B), p(q0s), p(tgs), C.byref(prm),
                               p(qs), p(oks), p(its), 0)
    assert rc == 0 and np.array_equal(qs.T, qo) and np.array_equal(oks, ok) and np.array_equal(its, it)


def test_cooperative_program_is_refused_when_four_workspaces_do_not_fit_the_lds_budget(emu):
    """Four workspaces + the tables must fit the CU's 160 KB of LDS (round 1: 64 KB).  M = 28 (two feet, pelvis, sixteen posture rows:
    2 k doubles per problem) fits now and matches the oracle; ten Full tasks (M = 60: 5 k doubles per problem) do not."""
    from ik_amd import capi
    name, ff, specs, root, edit = GENERIC_CASES["posture_regulariser"]
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, ff, specs, 4)
    p = lambda a: C.c_void_p(a.ctypes.data)
    prm = capi.DlsParams(5, 1e-2, 1.0, -1.0)
    qo, ok, it = np.empty_like(q0), np.zeros(4, np.uint8), np.zeros(4, np.int32)
    assert emu.lane_emu_dls_coop(urdf, C.c_size_t(len(urdf)), 1, tasks, len(tasks), C.c_int64(4), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1) == 0
    q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(5, 1e-2, 1.0, -1.0))
    assert np.abs(qo - q_ref).max() < 1e-8
    frames = ["LeftFootFront", "LeftFootBack", "RightFootFront", "RightFootBack", "pelvis"] * 2
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, [(f, "universe", 2, 0, None) for f in frames], 4)
    assert M == 60
    qo, ok, it = np.empty_like(q0), np.zeros(4, np.uint8), np.zeros(4, np.int32)
    assert emu.lane_emu_dls_coop(urdf, C.c_size_t(len(urdf)), 1, tasks, len(tasks), C.c_int64(4), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1) == 2
