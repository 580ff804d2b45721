"""ik::CentreOfMassTask (reference ik/ik/centre_of_mass.hpp:14-62; ik/ik/data.cpp:31-34) in the C oracle: known answers, finite
differences, and agreement with the independent numpy twin."""
import numpy as np
import pytest

from conftest import urdf_path

import oracle as O
import twin as T

CASES = [("cassie", True, "universe"), ("cassie", True, "pelvis"), ("cassie_fixed", False, "LeftFootFront"), ("ur5", False, "universe")]


def _pose(m, ff, rng):
    q = T.neutral(m)
    s = 7 if ff else 0
    q[s:] = np.clip(rng.uniform(-0.5, 0.5, m.nq - s), m.lower[s:], m.upper[s:])
    if ff:
        q = T.integrate(m, q, np.concatenate([rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.3, np.zeros(m.nv - 6)]))
    return q


def test_total_mass_and_a_hand_computed_centre_of_mass(native_built):
    # two unit-length links hinged about z, masses 2 and 3 at the link mid-points: com by hand
    xml = ("<robot><link name='base'/>"
           "<link name='a'><inertial><origin xyz='0.5 0 0'/><mass value='2'/></inertial></link>"
           "<link name='b'><inertial><origin xyz='0.5 0 0'/><mass value='3'/></inertial></link>"
           "<joint name='j1' type='revolute'><parent link='base'/><child link='a'/><axis xyz='0 0 1'/><limit lower='-3' upper='3'/></joint>"
           "<joint name='j2' type='revolute'><origin xyz='1 0 0'/><parent link='a'/><child link='b'/><axis xyz='0 0 1'/>"
           "<limit lower='-3' upper='3'/></joint></robot>")
    m = T.load_urdf(xml)
    assert m.mass.tolist() == [0.0, 2.0, 3.0] and np.allclose(m.lever[1:], [[0.5, 0, 0], [0.5, 0, 0]])
    q = np.array([0.3, -0.8])
    ca = 0.5 * np.array([np.cos(0.3), np.sin(0.3), 0])
    cb = np.array([np.cos(0.3), np.sin(0.3), 0]) + 0.5 * np.array([np.cos(-0.5), np.sin(-0.5), 0])
    want = (2 * ca + 3 * cb) / 5
    om = O.OracleModel(O.flat_from_twin(m))
    ot = O.make_tasks([(0, 0, 7, 0, None)])
    e, J = O.evaluate(om, ot, np.zeros((1, 12)), q)
    assert np.abs(e - want).max() < 1e-15
    # d com / d q1 = z x com;  d com / d q2 = (3/5) z x (cb - p2)
    z = np.array([0, 0, 1.0])
    assert np.abs(J[:, 0] - np.cross(z, want)).max() < 1e-15
    assert np.abs(J[:, 1] - 0.6 * np.cross(z, cb - np.array([np.cos(0.3), np.sin(0.3), 0]))).max() < 1e-15


@pytest.mark.parametrize("name,ff,ref", CASES)
def test_centre_of_mass_task_matches_twin_and_finite_differences(native_built, name, ff, ref):
    rng = np.random.default_rng(7)
    m = T.load_urdf(urdf_path(name), free_flyer=ff)
    om = O.OracleModel(O.flat_from_twin(m))
    q = _pose(m, ff, rng)
    t = T.CentreOfMassTask(m, ref, target=[0.01, 0.02, 0.9], weights=[1, 2, 0.5])
    e, J = T.evaluate(m, [t], q)
    tg = np.zeros((1, 12))
    tg[0, 9:] = t.target
    eo, Jo = O.evaluate(om, O.make_tasks([(0, t.reference, 7, 0, [1, 2, 0.5])]), tg, q)
    assert np.abs(e - eo).max() < 1e-14 and np.abs(J - Jo).max() < 1e-14
    com, Jcom = T.centre_of_mass(m, q)
    v, h = rng.normal(size=m.nv), 1e-6
    fd = (T.centre_of_mass(m, T.integrate(m, q, h * v))[0] - T.centre_of_mass(m, T.integrate(m, q, -h * v))[0]) / (2 * h)
    assert np.abs(Jcom @ v - fd).max() < 1e-9
    if ff:   # pushing the base along a world axis moves the centre of mass by the same amount
        assert np.abs(Jcom[:, :3] - T.fk(m, q)[0][1][:3, :3]).max() < 1e-15


@pytest.mark.parametrize("name,ff,ref", CASES)
def test_dls_with_a_centre_of_mass_task_matches_the_twin_and_converges(native_built, name, ff, ref):
    rng = np.random.default_rng(17)
    m = T.load_urdf(urdf_path(name), free_flyer=ff)
    om = O.OracleModel(O.flat_from_twin(m))
    q0, qs = _pose(m, ff, rng), None
    s = 7 if ff else 0
    qs = q0.copy()
    qs[s:] = np.clip(q0[s:] + rng.uniform(-0.1, 0.1, m.nq - s), m.lower[s:], m.upper[s:])
    t = T.CentreOfMassTask(m, ref)
    t.target = T.evaluate(m, [t], qs)[0]          # reachable: the centre of mass at q*, seen from the reference frame
    tg = np.zeros((1, 12))
    tg[0, 9:] = t.target
    ot = O.make_tasks([(0, t.reference, 7, 0, None)])
    q_t, ok_t, it_t = T.dls(m, [t], q0, 60, 1e-2, 1.0, 1e-16)
    q_o, ok_o, it_o = O.dls(om, ot, tg, q0, O.params(60, 1e-2, 1.0, 1e-16))
    assert ok_t == ok_o and it_t == it_o and np.abs(q_t - q_o).max() < 1e-9
    if ref == "universe":   # (seen from a moving frame the reference's Jacobian ignores that frame's motion: no such promise)
        assert np.abs(O.evaluate(om, ot, tg, q_o)[0]).max() < 1e-6
