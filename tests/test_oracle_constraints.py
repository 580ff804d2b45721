"""ik::FrameConstraint (reference ik/ik/frame.hpp:325-449) and the null-space projection of ik::dls (reference
ik/ik/dls.cpp:26-34,43-53) in the C oracle: the constraint Jacobian against finite differences and against the numpy twin,
the projected step, and the invariants the projection buys."""
import numpy as np
import pytest

from conftest import urdf_path

import oracle as O
import twin as T

CASES = [
    ("cassie", True, [("LeftFootFront", "pelvis", 0), ("pelvis", "universe", 2)], [("RightFootFront", "universe", 0)]),
    ("cassie", True, [("pelvis", "universe", 2)], [("RightFootFront", "universe", 2), ("LeftFootFront", "RightFootFront", 0)]),
    ("ur5", False, [("tool0", "universe", 0)], [("tool0", "universe", 1)]),
    ("cassie_fixed", False, [("LeftFootFront", "universe", 2)], [("RightFootFront", "LeftFootBack", 1)]),
]


def _setup(k):
    name, ff, tspec, cspec = CASES[k]
    rng = np.random.default_rng(40 + k)
    m = T.load_urdf(urdf_path(name), free_flyer=ff)
    om = O.OracleModel(O.flat_from_twin(m))
    s = 7 if ff else 0
    lo, hi = np.maximum(m.lower, -1.0), np.minimum(m.upper, 1.0)
    q0, qs = T.neutral(m), T.neutral(m)
    q0[s:] = np.clip(0.5 * (lo[s:] + hi[s:]) + rng.uniform(-0.2, 0.2, m.nq - s), m.lower[s:], m.upper[s:])
    qs[s:] = np.clip(q0[s:] + rng.uniform(-0.15, 0.15, m.nq - s), m.lower[s:], m.upper[s:])
    if ff:
        qs[:3] = [0.05, -0.03, 0.02]
    oMf = T.fk(m, qs)[1]
    tasks, ospec, tg = [], [], []
    for f, r, t in tspec:
        ft = T.FrameTask(m, f, t, r)
        ft.target = T.se3_inv(oMf[ft.reference]) @ oMf[ft.frame]
        tasks.append(ft)
        ospec.append((ft.frame, ft.reference, t, 0, None))
        tg.append(np.concatenate([ft.target[:3, :3].ravel(), ft.target[:3, 3]]))
    cons = [T.FrameTask(m, f, t, r) for f, r, t in cspec]
    oc = O.make_tasks([(c.frame, c.reference, c.type, 0, None) for c in cons])
    return m, om, q0, tasks, O.make_tasks(ospec), np.array(tg), cons, oc, rng


@pytest.mark.parametrize("k", range(len(CASES)))
def test_constraint_jacobian_is_the_local_relative_velocity(native_built, k):
    m, om, q0, tasks, ot, tg, cons, oc, rng = _setup(k)
    Jc = O.constraint_jacobian(om, oc, q0)
    assert np.abs(Jc - T.constraint_jacobian(m, cons, q0)).max() < 1e-13
    # finite differences: the motion of the frame relative to the reference frame, seen from the frame itself
    v = rng.normal(size=m.nv)
    h = 1e-6
    row = 0
    for c in cons:
        def rel(q):
            oMf = T.fk(m, q)[1]
            return T.se3_inv(oMf[c.reference]) @ oMf[c.frame]
        d = T.log6(T.se3_inv(rel(T.integrate(m, q0, -h * v))) @ rel(T.integrate(m, q0, h * v))) / (2 * h)
        dim = 6 if c.type == 2 else 3
        assert np.abs(Jc[row:row + dim] @ v - d[c.rows()]).max() < 1e-7
        row += dim


@pytest.mark.parametrize("k", range(len(CASES)))
def test_constrained_dls_matches_the_twin_and_stays_in_the_null_space(native_built, k):
    m, om, q0, tasks, ot, tg, cons, oc, rng = _setup(k)
    q_t, ok_t, it_t = T.dls(m, tasks, q0, 30, 1e-2, 1.0, 1e-10, constraints=cons)
    q_o, ok_o, it_o = O.dls_constrained(om, ot, oc, tg, q0, O.params(30, 1e-2, 1.0, 1e-10))
    assert ok_t == ok_o and it_t == it_o and np.abs(q_t - q_o).max() < 1e-10
    # one step: Jc dq = 0 (dq = -N (...), N = I - pinv(Jc) Jc)
    q1, _, _ = O.dls_constrained(om, ot, oc, tg, q0, O.params(1, 1e-2, 1.0, -1.0))
    dq = q1 - q0 if not m.names[1] == "root_joint" else None
    if dq is not None and np.all((q1 > m.lower + 1e-9) & (q1 < m.upper - 1e-9)):
        assert np.abs(O.constraint_jacobian(om, oc, q0) @ dq).max() < 1e-12
    # no constraints: the plain loop
    qa, oka, ita = O.dls_constrained(om, ot, O.make_tasks([]), tg, q0, O.params(7, 1e-2, 1.0, -1.0))
    qb, okb, itb = O.dls(om, ot, tg, q0, O.params(7, 1e-2, 1.0, -1.0))
    assert np.array_equal(qa, qb) and oka == okb and ita == itb
    # batch entry point
    qq, okk, itt = O.dls_batch_constrained(om, ot, oc, tg[None], q0[None], O.params(30, 1e-2, 1.0, 1e-10), nthreads=2)
    assert np.array_equal(qq[0], q_o) and bool(okk[0]) == ok_o and itt[0] == it_o
