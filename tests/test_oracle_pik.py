"""ik::pik (reference ik/ik/pik.cpp:5-103) in the C oracle: known answers of its two matrix functions, agreement with the
independent numpy twin, and the committed golden vectors (tests/golden/P_pik.json, produced by the twin)."""
import json
import os

import numpy as np
import pytest

from conftest import urdf_path

import oracle as O
import twin as T

HERE = os.path.dirname(os.path.abspath(__file__))


def test_damp_pseudoinverse_known_answers(native_built):
    # diagonal: sigma / (lambda^2 + sigma^2) on the diagonal of the transpose (pik.cpp:14-18)
    A = np.zeros((2, 3))
    A[0, 0], A[1, 1] = 2.0, 0.5
    R = O.damp_pseudoinverse(A, 0.5)
    want = np.zeros((3, 2))
    want[0, 0], want[1, 1] = 2.0 / (0.25 + 4.0), 0.5 / (0.25 + 0.25)
    assert np.abs(R - want).max() < 1e-15
    rng = np.random.default_rng(0)
    for m, n in ((3, 7), (6, 7), (7, 3), (6, 6), (10, 22), (18, 16), (1, 5), (5, 1)):
        A = rng.normal(size=(m, n))
        # lambda = 0 on a full-rank matrix is the Moore-Penrose inverse
        assert np.abs(O.damp_pseudoinverse(A, 0.0) - np.linalg.pinv(A)).max() < 1e-12
        # the closed form A^T (A A^T + lambda^2 I)^-1 and the twin (LAPACK SVD)
        lam = 0.3
        assert np.abs(O.damp_pseudoinverse(A, lam) - A.T @ np.linalg.inv(A @ A.T + lam * lam * np.eye(m))).max() < 1e-13
        assert np.abs(O.damp_pseudoinverse(A, lam) - T.damp_pseudoinverse(A, lam)).max() < 1e-14
    assert np.abs(O.damp_pseudoinverse(np.zeros((3, 5)), 1.0)).max() == 0.0


def test_rowspace_projector_properties_and_rank(native_built):
    rng = np.random.default_rng(1)
    for m, n, r in ((3, 7, 3), (6, 7, 4), (7, 3, 3), (7, 3, 2), (10, 22, 9), (18, 16, 16), (4, 4, 1)):
        A = rng.normal(size=(m, r)) @ rng.normal(size=(r, n))      # rank r exactly (up to rounding)
        P = O.rowspace_projector(A)
        assert np.abs(P - P.T).max() < 1e-14 and np.abs(P @ P - P).max() < 1e-13      # an orthogonal projector
        assert abs(np.trace(P) - r) < 1e-12                                              # of rank r
        assert np.abs(A @ P - A).max() < 1e-12 * np.abs(A).max()                          # onto the row space of A
        assert np.abs(P - T.rowspace_projector(A)).max() < 1e-12
        assert np.abs(P - np.linalg.pinv(A, rcond=1e-12) @ A).max() < 1e-11              # pinv(A) A, pik.cpp:59-61
    assert np.abs(O.rowspace_projector(np.zeros((3, 5)))).max() == 0.0
    # Eigen's rank test is RELATIVE to the largest pivot: a matrix of pure rounding noise has full rank
    N = rng.normal(size=(3, 6)) * 1e-17
    assert abs(np.trace(O.rowspace_projector(N)) - 3) < 1e-12 and T.cod_rank(N) == 3


def _levels(m, spec, oMf, rng):
    levels, ospec, tg = [], [], []
    for p, lv in enumerate(spec):
        row = []
        for f, r, typ, w in lv:
            t = T.FrameTask(m, f, typ, r, weights=w)
            if typ >= T.ALIGN_X:
                t.target = np.eye(4)
                t.target[:3, 3] = rng.normal(size=3)
            else:
                t.target = T.se3_inv(oMf[t.reference]) @ oMf[t.frame]
            row.append(t)
            ospec.append((t.frame, t.reference, typ, p, w))
            tg.append(np.concatenate([t.target[:3, :3].ravel(), t.target[:3, 3]]))
        levels.append(row)
    return levels, O.make_tasks(ospec), np.array(tg)


LOOP_CASES = [
    ("ur5", False, [[("tool0", "universe", 0, None)], [("tool0", "universe", 1, None)]], [1.0, 1.0], None, 20, 1.0, -1.0),
    ("ur5", False, [[("tool0", "universe", 0, None)], [("tool0", "universe", 1, None)]], [0.05, 0.1], None, 40, 1.0, 1e-8),
    ("ur5", False, [[("tool0", "universe", 2, None)], [("forearm_link", "universe", 0, None)]], [0.1, 0.1], None, 30, 1.0, -1.0),
    ("cassie_fixed", False, [[("LeftFootFront", "universe", 2, None)], [("RightFootFront", "universe", 0, [1, 2, 0.5])]], [0.1, 0.5],
     "ramp", 15, 1.0, -1.0),
    ("cassie", True, [[("LeftFootFront", "pelvis", 0, None), ("pelvis", "universe", 2, None)], [("LeftFootFront", "universe", 4, None)]],
     [0.1, 0.1], None, 30, 0.5, -1.0),
    ("cassie", True, [[("LeftFootFront", "universe", 2, None), ("RightFootFront", "universe", 2, None)], [("pelvis", "universe", 2, None)]],
     [0.05, 0.05], None, 30, 1.0, 1e-10),
    ("cassie", True, [[("pelvis", "universe", 2, None)], [], [("LeftFootFront", "universe", 0, None)]], [0.1, 1.0, 0.1], None, 10, 1.0, -1.0),
]


@pytest.mark.parametrize("k", range(len(LOOP_CASES)))
def test_oracle_pik_matches_the_twin(native_built, k):
    name, ff, spec, lam, da, iters, step, tol = LOOP_CASES[k]
    rng = np.random.default_rng(100 + k)
    m = T.load_urdf(urdf_path(name), free_flyer=ff)
    om = O.OracleModel(O.flat_from_twin(m))
    s = 7 if ff else 0
    lo, hi = np.maximum(m.lower, -1.0), np.minimum(m.upper, 1.0)
    q0, qs = T.neutral(m), T.neutral(m)
    q0[s:] = np.clip(0.5 * (lo[s:] + hi[s:]) + rng.uniform(-0.2, 0.2, m.nq - s), m.lower[s:], m.upper[s:])
    qs[s:] = np.clip(q0[s:] + rng.uniform(-0.15, 0.15, m.nq - s), m.lower[s:], m.upper[s:])
    if ff:
        qs[:3] = [0.05, -0.03, 0.02]
    levels, ot, tg = _levels(m, spec, T.fk(m, qs)[1], rng)
    da = None if da is None else list(0.01 * np.arange(m.nv))
    trace = []
    q_t, ok_t, it_t = T.pik(m, levels, q0, iters, step, tol, lam, da, trace=trace)
    q_o, ok_o, it_o, tr = O.pik(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, da), trace=True)
    assert ok_t == ok_o and it_t == it_o
    assert np.abs(tr[0, -m.nv:] - trace[0]["dq"]).max() < 1e-12      # first step, before anything accumulates
    assert np.abs(q_t - q_o).max() < 1e-10
    # the batch entry point is the same loop
    qb, okb, itb = O.pik_batch(om, ot, tg[None], q0[None], O.pik_params(iters, step, tol, lam, da), nthreads=2)
    assert np.array_equal(qb[0], q_o) and bool(okb[0]) == ok_o and itb[0] == it_o


def test_pik_with_one_level_is_dls(native_built):
    """One level: dq = -damp_pseudoinverse(J, lambda) e = -J^T (J J^T + lambda^2 I)^-1 e, the step of ik::dls with
    damping = lambda (reference ik/ik/dls.cpp:39-53)."""
    m = T.load_urdf(urdf_path("cassie_fixed"))
    om = O.OracleModel(O.flat_from_twin(m))
    rng = np.random.default_rng(5)
    q0 = np.clip(rng.uniform(-0.3, 0.3, m.nq), m.lower, m.upper)
    qs = np.clip(q0 + rng.uniform(-0.2, 0.2, m.nq), m.lower, m.upper)
    fid = om.frame_id("LeftFootFront")
    ot = O.make_tasks([(fid, 0, 2, 0, None)])
    tg = O.fk(om, qs)[1][[fid]]
    qa, oka, ita = O.pik(om, ot, tg, q0, O.pik_params(40, 1.0, 1e-10, [1e-2]))
    qb, okb, itb = O.dls(om, ot, tg, q0, O.params(40, 1e-2, 1.0, 1e-10))
    assert oka == okb and ita == itb and np.abs(qa - qb).max() < 1e-9


def test_pik_level_count_must_match(native_built):
    m = T.load_urdf(urdf_path("ur5"))
    om = O.OracleModel(O.flat_from_twin(m))
    ot = O.make_tasks([(om.frame_id("tool0"), 0, 0, 0, None), (om.frame_id("tool0"), 0, 1, 1, None)])
    with pytest.raises(ValueError):
        O.pik(om, ot, np.zeros((2, 12)), np.zeros(m.nq), O.pik_params(1, 1.0, -1.0, [1.0]))


def test_pik_levels_without_tasks_are_no_ops(native_built):
    """The demo declares two priority levels and puts every task on the first (reference ik_ros/src/cassie.cpp:43,72-81): the
    reference's loop over 0..max_priority_level (ik/ik/pik.cpp:47) then meets a level with no rows, which changes nothing."""
    m = T.load_urdf(urdf_path("cassie"), free_flyer=True)
    om = O.OracleModel(O.flat_from_twin(m))
    rng = np.random.default_rng(77)
    q0, qs = T.neutral(m), T.neutral(m)
    q0[7:] = np.clip(rng.uniform(-0.2, 0.2, m.nq - 7), m.lower[7:], m.upper[7:])
    qs[7:] = np.clip(q0[7:] + rng.uniform(-0.1, 0.1, m.nq - 7), m.lower[7:], m.upper[7:])
    spec = [[("LeftFootFront", "pelvis", 0, None), ("pelvis", "universe", 2, None), ("LeftFootFront", "universe", 4, None)]]
    levels, ot, tg = _levels(m, spec, T.fk(m, qs)[1], rng)
    qa, oka, ita = O.pik(om, ot, tg, q0, O.pik_params(25, 0.5, 1e-9, [0.1]))
    qb, okb, itb = O.pik(om, ot, tg, q0, O.pik_params(25, 0.5, 1e-9, [0.1, 1.0]))
    assert np.array_equal(qa, qb) and oka == okb and ita == itb
    qt, okt, itt = T.pik(m, levels + [[]], q0, 25, 0.5, 1e-9, [0.1, 1.0])
    assert okt == oka and itt == ita and np.abs(qt - qa).max() < 1e-10


def test_oracle_pik_reproduces_golden_vectors(native_built):
    with open(os.path.join(HERE, "golden", "P_pik.json")) as fh:
        cases = json.load(fh)
    for g in cases:
        m = T.load_urdf(os.path.join(os.path.dirname(urdf_path("x")), g["urdf"]), g["free_flyer"])
        om = O.OracleModel(O.flat_from_twin(m))
        ot = O.make_tasks([(om.frame_id(t["frame"]), om.frame_id(t["reference"]), t["type"], p, None)
                           for p, lv in enumerate(g["levels"]) for t in lv])
        for pr in g["problems"]:
            q0, tg = np.array(pr["q0"]), np.array(pr["targets"])
            for iters, step, key in ((1, 1.0, "q_after_1"), (4, 1.0, "q_after_4"), (30, 0.5, "q_after_30_half_step")):
                q, ok, it = O.pik(om, ot, tg, q0, O.pik_params(iters, step, -1.0, g["lam"]))
                assert np.abs(q - pr[key]).max() < 1e-9, (g["name"], key)
            q, ok, it = O.pik(om, ot, tg, q0, O.pik_params(100, 1.0, 1e-4, g["lam"]))
            d = pr["default_stop"]
            assert ok == d["success"] and it == d["iterations"] and np.abs(q - d["q"]).max() < 1e-9
