"""URDF "continuous" joints: Pinocchio's JointModelRevoluteUnbounded* -- configuration (cos, sin), nq = 2, nv = 1, position limits
-1.01 / +1.01 on both entries, integrate = rotate the pair and renormalise to first order (include/ikgpu.h).  None of the
reference's URDFs has one (so nothing here can be checked against reference data); the fixture is the UR5 with two of its joints
declared continuous.  Loader against the twin's loader, oracle against the twin, the device's generic lane programs (CPU
emulation) and the generic kernels on the GPU against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, urdf_path

import oracle as O
import twin as T

CONT = ["shoulder_pan_joint", "wrist_3_joint"]


def continuous_ur5():
    xml = open(urdf_path("ur5")).read()
    for n in CONT:
        xml, k = re.subn(r'(<joint name="%s" type=")revolute(")' % n, r'\1continuous\2', xml)
        assert k == 1
    return xml


def to_config(model_flat, angles):
    """angles [B, nv] -> q [B, nq]: (cos, sin) for a continuous joint, the angle otherwise."""
    B = angles.shape[0]
    q = np.zeros((B, model_flat["nq"]))
    for j in range(1, len(model_flat["jtype"])):
        iq, iv = model_flat["idx_q"][j], model_flat["idx_v"][j]
        if model_flat["jtype"][j] == 4:
            q[:, iq], q[:, iq + 1] = np.cos(angles[:, iv]), np.sin(angles[:, iv])
        else:
            q[:, iq] = angles[:, iv]
    return q


@pytest.fixture(scope="module")
def case(native_built):
    import ik_amd
    xml = continuous_ur5()
    model = ik_amd.Model.from_urdf_xml(xml)
    flat = model.flat()
    rng = np.random.default_rng(5)
    B = 48
    a0 = np.array([0.3, -1.4, 1.5, 0.1, 1.4, -0.6]) + rng.uniform(-0.2, 0.2, (B, 6))
    a1 = a0 + rng.uniform(-0.15, 0.15, (B, 6))
    q0, qs = to_config(flat, a0), to_config(flat, a1)
    om = O.OracleModel(flat)
    fid = model.getFrameId("tool0")
    tg = O.fk_batch(om, qs, [fid])
    return dict(xml=xml, model=model, flat=flat, om=om, fid=fid, q0=q0, qs=qs, tg=tg, a0=a0)


def test_loader_builds_the_cos_sin_joint_and_matches_the_twin(case):
    model, flat = case["model"], case["flat"]
    assert (model.nq, model.nv) == (8, 6)
    assert list(flat["jtype"]) == [0, 4, 1, 1, 1, 1, 4] and list(flat["idx_q"]) == [0, 0, 2, 3, 4, 5, 6] and list(flat["idx_v"]) == [0, 0, 1, 2, 3, 4, 5]
    assert np.array_equal(model.lowerPositionLimit[[0, 1, 6, 7]], [-1.01] * 4) and np.array_equal(model.upperPositionLimit[[0, 1, 6, 7]], [1.01] * 4)
    tm = T.load_urdf(case["xml"])
    tf = O.flat_from_twin(tm)
    for key in ("jtype", "parent", "idx_q", "idx_v", "frame_parent"):
        assert np.array_equal(flat[key], tf[key]), key
    for key in ("placement", "axis", "lower", "upper", "frame_placement"):
        assert np.array_equal(flat[key], tf[key]), key
    # the same kinematics as the revolute model at the same angles
    import ik_amd
    rev = ik_amd.Model.from_urdf_file(urdf_path("ur5"))
    want = O.fk_batch(O.OracleModel(rev.flat()), case["a0"], [rev.getFrameId("tool0")])
    got = O.fk_batch(case["om"], case["q0"], [case["fid"]])
    assert np.abs(got - want).max() < 1e-15


def test_oracle_matches_the_twin(case):
    tm = T.load_urdf(case["xml"])
    om, fid = case["om"], case["fid"]
    for b in range(6):
        q0, tgt = case["q0"][b], case["tg"][b, 0]
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = tgt[:9].reshape(3, 3), tgt[9:]
        task = T.FrameTask(tm, "tool0", T.FULL, "universe", target=M)
        q_t, ok_t, it_t = T.dls(tm, [task], q0, max_iterations=25, damping=1e-2, step_length=1.0, stop_sq_tol=1e-10)
        q_o, ok_o, it_o = O.dls(om, O.make_tasks([(fid, 0, 2, 0, None)]), case["tg"][b], q0, O.params(25, 1e-2, 1.0, 1e-10))
        assert ok_t == ok_o and it_t == it_o
        assert np.abs(q_t - q_o).max() < 1e-10
        # the pair stays on the unit circle (first-order renormalisation) and the solve converges
        assert abs(np.hypot(q_o[0], q_o[1]) - 1.0) < 1e-12 and abs(np.hypot(q_o[6], q_o[7]) - 1.0) < 1e-12
        assert ok_o
    # integrate: rotation of the pair by v, (3 - |.|^2) / 2 renormalisation; a revolute entry is q + v
    q = case["q0"][0].copy()
    q[0], q[1] = 1.2 * q[0], 1.2 * q[1]                       # deliberately off the circle
    v = np.array([0.3, -0.1, 0.2, 0.0, 0.1, -0.4])
    out = O.integrate(om, q, v)
    c, s = np.cos(0.3) * q[0] - np.sin(0.3) * q[1], np.sin(0.3) * q[0] + np.cos(0.3) * q[1]
    k = (3.0 - (c * c + s * s)) / 2.0
    assert np.allclose(out[:2], [c * k, s * k], rtol=0, atol=1e-15) and np.allclose(out[2:6], q[2:6] + v[1:5], rtol=0, atol=1e-15)
    assert np.allclose(out, T.integrate(tm, q, v), rtol=0, atol=1e-15)


def test_generic_lane_programs_match_the_oracle(case, native_built):
    """The per-lane generic program and the cooperative one (CPU emulation of the device code)."""
    import subprocess
    from ik_amd import capi
    src = os.path.join(ROOT, "tests", "lane_emu", "lane_emu.cpp")
    out = os.path.join(ROOT, "tests", "lane_emu", "liblane_emu.so")
    deps = [src] + [os.path.join(ROOT, "ik_amd", "csrc", f) for f in ("model.cpp", "problem.cpp", "device/generic_solver.hpp", "device/coop_solver.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ik_amd", "csrc"), "-o", out, src,
                               os.path.join(ROOT, "ik_amd", "csrc", "model.cpp"), os.path.join(ROOT, "ik_amd", "csrc", "problem.cpp")])
    L = C.CDLL(out)
    L.lane_emu_last_error.restype = C.c_char_p
    model, om, fid, q0, tg = case["model"], case["om"], case["fid"], case["q0"], case["tg"]
    B = q0.shape[0]
    urdf = case["xml"].encode()
    task = capi.Task(fid, 0, 2, 0, (C.c_double * 6)(*[1.0] * 6))
    p = lambda a: C.c_void_p(a.ctypes.data)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (6, 1e-2, 1.0, -1.0), (60, 1e-1, 0.5, 1e-8)):
        prm = capi.DlsParams(iters, damping, step, tol)
        q_ref, ok_ref, it_ref = O.dls_batch(om, O.make_tasks([(fid, 0, 2, 0, None)]), tg, q0, O.params(iters, damping, step, tol))
        for coop in (False, True):
            qo = np.empty_like(q0)
            ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
            if coop:   # device/coop_solver.hpp
                rc = L.lane_emu_dls_coop(urdf, C.c_size_t(len(urdf)), 0, C.byref(task), 1, C.c_int64(B), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1)
            else:      # device/generic_solver.hpp
                e, J, oMf = np.empty((B, 6)), np.empty((B, 6, model.nv)), np.empty((B, 1, 12))
                rc = L.lane_emu_run(urdf, C.c_size_t(len(urdf)), 0, C.byref(task), 1, 0, C.c_int64(B), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it),
                                    p(e), p(J), p(oMf), 1)
            assert rc == 0, L.lane_emu_last_error()
            assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (iters, coop)
            assert np.abs(qo - q_ref).max() < 1e-9, (iters, coop, np.abs(qo - q_ref).max())


@pytest.mark.gpu
def test_generic_kernels_match_the_oracle_on_the_gpu(case, monkeypatch):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ik_amd
    model, om, fid, q0, tg = case["model"], case["om"], case["fid"], case["q0"], case["tg"]
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel.startswith("dls_generic<")          # a model with continuous joints runs on the generic kernel
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    Tg = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for form in ("coop", "lane"):
        if form == "lane":
            monkeypatch.setenv("IKGPU_GENERIC_KERNEL", "lane")
        for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (6, 1e-2, 1.0, -1.0), (60, 1e-1, 0.5, 1e-8)):
            Q, ok, it = ik_amd.dls_batch(problem, Q0, Tg, data, ik_amd.inverse_kinematics_visitor(tol),
                                         ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step))
            q_ref, ok_ref, it_ref = O.dls_batch(om, O.make_tasks([(fid, 0, 2, 0, None)]), tg, q0, O.params(iters, damping, step, tol))
            assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (form, iters)
            assert np.abs(Q.cpu().numpy().T - q_ref).max() <= 1e-6, (form, iters)
    monkeypatch.delenv("IKGPU_GENERIC_KERNEL", raising=False)
