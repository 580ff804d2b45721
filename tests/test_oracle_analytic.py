"""Oracle-free analytic checks of the CPU oracle (oracle/ik_oracle.c): the reference pins no numbers
(every TEST in reference ik/test/*.cpp is commented out), so the restated Pinocchio semantics
 are anchored on identities and on scipy."""
import numpy as np
import pytest
import scipy.linalg as sl

from conftest import urdf_path

import oracle as O
import twin as T


def hat6(nu):
    H = np.zeros((4, 4))
    H[:3, :3] = T.skew(nu[3:])
    H[:3, 3] = nu[:3]
    return H


def to4(M12):
    M = np.eye(4)
    M[:3, :3] = np.asarray(M12[:9]).reshape(3, 3)
    M[:3, 3] = M12[9:]
    return M


@pytest.fixture(scope="module")
def models(native_built):
    out = {}
    for name, ff in (("cassie_fixed", False), ("cassie", True), ("ur5", False)):
        tm = T.load_urdf(urdf_path(name), ff)
        out[name] = (tm, O.OracleModel(O.flat_from_twin(tm)))
    return out


def test_exp6_matches_expm_and_log6_inverts(native_built):
    rng = np.random.default_rng(0)
    for scale in (1e-10, 1e-6, 1e-4, 1e-2, 1.0, 3.0):
        for _ in range(20):
            w = rng.normal(size=3)
            nu = np.concatenate([rng.normal(size=3), w / np.linalg.norm(w) * scale])
            M = O.exp6(nu)
            assert np.abs(to4(M) - sl.expm(hat6(nu))).max() < 5e-15 * max(1.0, np.abs(nu).max())
            assert np.abs(O.log6(M) - nu).max() < 1e-9 * max(1.0, scale)
            if 1e-3 < scale < 3.0:
                L = sl.logm(to4(M)).real
                assert np.abs(np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]]) - O.log6(M)).max() < 1e-11


def test_log3_near_pi_branch(native_built):
    rng = np.random.default_rng(1)
    for _ in range(50):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = np.pi - rng.uniform(0, 9e-3)
        nu = np.concatenate([np.zeros(3), ax * ang])
        got = O.log6(O.exp6(nu))[3:]
        assert np.abs(got - ax * ang).max() < 1e-9


def test_Jlog6_is_the_right_jacobian_of_log6(native_built):
    rng = np.random.default_rng(2)
    for _ in range(30):
        nu = rng.normal(size=6)
        if np.linalg.norm(nu[3:]) > 2.8:
            continue
        M4 = to4(O.exp6(nu))
        J = O.Jlog6(O.exp6(nu))
        Jfd = np.zeros((6, 6))
        h = 1e-6
        for i in range(6):
            d = np.zeros(6)
            d[i] = h
            plus = T.log6(M4 @ T.exp6(d))
            minus = T.log6(M4 @ T.exp6(-d))
            Jfd[:, i] = (plus - minus) / (2 * h)
        assert np.abs(J - Jfd).max() < 5e-8


def test_lie_maps_small_angle_taylor_is_continuous(native_built):
    ax = np.array([0.3, -0.5, 0.8])
    ax /= np.linalg.norm(ax)
    p = np.array([0.4, -0.2, 0.7])
    below, above = T.TAYLOR_PREC3 * (1 - 1e-6), T.TAYLOR_PREC3 * (1 + 1e-6)
    for f in (O.log6, lambda M: O.Jlog6(M).ravel()):
        a = f(np.concatenate([T.exp3(ax * below).ravel(), p]))
        b = f(np.concatenate([T.exp3(ax * above).ravel(), p]))
        assert np.abs(a - b).max() < 1e-8


def test_fk_known_answers(models):
    tm, om = models["ur5"]
    _, oMf = O.fk(om, np.zeros(6))
    assert np.allclose(oMf[om.frame_id("tool0")][9:], [0.81725, 0.19145, -0.005491], atol=1e-12)
    tm, om = models["cassie_fixed"]
    q = np.array([0.0045, 0, 0.4973, -1.1997, 0, 1.4267, 0, -1.5968, -0.0045, 0, 0.4973, -1.1997, 0, 1.4267, 0, -1.5968])
    _, oMf = O.fk(om, q)
    L, R = oMf[om.frame_id("LeftFootFront")][9:], oMf[om.frame_id("RightFootFront")][9:]
    assert np.allclose(L * [1, -1, 1], R, atol=1e-12)  # mirror symmetry of the two legs


def test_model_shapes(models):
    shapes = {"cassie": (18, 23, 22, 47), "cassie_fixed": (17, 16, 16, 43), "ur5": (7, 6, 6, 23)}
    for name, (nj, nq, nv, nf) in shapes.items():
        tm, om = models[name]
        assert (om.njoints, om.nq, om.nv, om.nframes) == (nj, nq, nv, nf)
    tm, _ = models["cassie"]
    assert tm.names[:3] == ["universe", "root_joint", "LeftHipRoll"] and tm.idx_q[2] == 7
    assert models["ur5"][0].names[1:] == ["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint",
                                          "wrist_2_joint", "wrist_3_joint"]


@pytest.mark.parametrize("name,frames", [("cassie_fixed", ["LeftFootFront"]), ("ur5", ["tool0"]),
                                         ("cassie", ["LeftFootFront", "RightFootFront", "pelvis"])])
def test_task_jacobian_is_the_derivative_of_the_error(models, name, frames):
    """J = -Jlog6(tMf) J_local (reference ik/ik/frame.hpp:152-182) is exact for a universe reference."""
    tm, om = models[name]
    rng = np.random.default_rng(3)
    q = T.neutral(tm)
    nj0 = 7 if name == "cassie" else 0
    lo, hi = np.maximum(tm.lower[nj0:], -2.5), np.minimum(tm.upper[nj0:], 2.5)
    q[nj0:] = rng.uniform(lo, hi)
    qs = q.copy()
    qs[nj0:] = np.clip(q[nj0:] + rng.uniform(-0.3, 0.3, lo.size), lo, hi)
    tasks = O.make_tasks([(om.frame_id(f), 0, 2, 0, None) for f in frames])
    tg = O.fk(om, qs)[1][[om.frame_id(f) for f in frames]]
    e, J = O.evaluate(om, tasks, tg, q)
    Jfd = np.zeros_like(J)
    h = 1e-6
    for i in range(om.nv):
        d = np.zeros(om.nv)
        d[i] = h
        ep = O.evaluate(om, tasks, tg, O.integrate(om, q, d))[0]
        em = O.evaluate(om, tasks, tg, O.integrate(om, q, -d))[0]
        Jfd[:, i] = (ep - em) / (2 * h)
    assert np.abs(J - Jfd).max() < 1e-7


def test_freeflyer_integrate_keeps_unit_quaternion_and_composes(models):
    tm, om = models["cassie"]
    rng = np.random.default_rng(4)
    q = T.neutral(tm)
    q[:3] = rng.normal(size=3)
    quat = rng.normal(size=4)
    q[3:7] = quat / np.linalg.norm(quat)
    v = np.zeros(om.nv)
    v[:6] = rng.normal(size=6) * 0.3
    q1 = O.integrate(om, q, v)
    assert abs(np.linalg.norm(q1[3:7]) - 1) < 1e-14
    M0 = T.se3(T.quat_to_matrix(*q[3:7]), q[:3])
    M1 = T.se3(T.quat_to_matrix(*q1[3:7]), q1[:3])
    assert np.abs(M1 - M0 @ sl.expm(hat6(v[:6]))).max() < 1e-13
    assert np.abs(O.integrate(om, q1, -v) - q).max() < 1e-13


def test_one_dls_step_solves_the_damped_normal_equations(models):
    """dq = -J^T (J J^T + lambda^2 I)^-1 e (reference ik/ik/dls.cpp:39-53), checked with numpy's LU."""
    for name, frames in (("cassie_fixed", ["LeftFootFront"]), ("cassie", ["LeftFootFront", "RightFootFront", "pelvis"])):
        tm, om = models[name]
        rng = np.random.default_rng(5)
        nj0 = 7 if name == "cassie" else 0
        q = T.neutral(tm)
        q[nj0:] = rng.uniform(tm.lower[nj0:], tm.upper[nj0:])
        qs = q.copy()
        qs[nj0:] = rng.uniform(tm.lower[nj0:], tm.upper[nj0:])
        tasks = O.make_tasks([(om.frame_id(f), 0, 2, 0, None) for f in frames])
        tg = O.fk(om, qs)[1][[om.frame_id(f) for f in frames]]
        e, J = O.evaluate(om, tasks, tg, q)
        lam = 1e-2
        dq = -J.T @ np.linalg.solve(J @ J.T + lam * lam * np.eye(len(e)), e)
        q1, ok, it = O.dls(om, tasks, tg, q, O.params(1, lam, 1.0, -1.0))
        want = np.minimum(om.upper, np.maximum(O.integrate(om, q, dq), om.lower))
        assert np.abs(q1 - want).max() < 1e-9 and not ok and it == 1
