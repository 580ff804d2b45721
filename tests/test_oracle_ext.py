"""The extended-precision builds of the oracle (oracle/ik_oracle_ext.c: the same statements in _Float128 / x87 long double) against
the double oracle: they arbitrate, in the GPU parity tests, the lanes the perturbation probes exclude -- so they must BE the same
algorithm (agreement to rounding on well-conditioned problems, same flags) and must be more precise (the two wide builds agree with
each other far better than either agrees with the double build)."""
import numpy as np
import pytest

from conftest import urdf_path


@pytest.fixture(scope="module")
def case(native_built):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("LeftFootFront")
    B = 256
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "near")
    return O, om, O.make_tasks([(fid, 0, 2, 0, None)]), O.fk_batch(om, qs, [fid]), q0


def test_wide_builds_report_their_significands(case):
    O = case[0]
    assert O.ext_lib("q").iko_ext_bits() == 113 and O.ext_lib("ld").iko_ext_bits() == 64


@pytest.mark.parametrize("stop", [-1.0, 1e-4])
def test_wide_builds_are_the_same_algorithm_and_more_precise(case, stop):
    O, om, tasks, tg, q0 = case
    prm = O.params(50, 1e-2, 1.0, stop)
    q, ok, it = O.dls_batch(om, tasks, tg, q0, prm, 4)
    ql, okl, itl = O.dls_batch(om, tasks, tg, q0, prm, 4, ext="ld")
    qq, okq, itq = O.dls_batch(om, tasks, tg, q0, prm, 4, ext="q")
    assert np.array_equal(ok, okq) and np.array_equal(it, itq) and np.array_equal(ok, okl) and np.array_equal(it, itl)
    d_double, d_ld = np.abs(q - qq).max(), np.abs(ql - qq).max()
    assert d_double < 1e-11, d_double          # converging problems: rounding-level agreement
    assert d_ld <= 1e-15, d_ld                 # both wide results round to (nearly) the same double
    assert np.median(np.abs(ql - qq).max(axis=1)) == 0.0


def test_wide_builds_take_constraints_and_pik(native_built):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    om = O.OracleModel(model.flat())
    B = 8
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0)
    frames = [model.getFrameId("LeftFootFront"), model.getFrameId("pelvis")]
    tasks = O.make_tasks([(frames[0], 0, 2, 0, None), (frames[1], 0, 2, 1, None)])
    tg = O.fk_batch(om, qs, frames)
    cons = O.make_tasks([(model.getFrameId("RightFootFront"), 0, 0, 0, None)])
    prm = O.params(20, 1e-2, 1.0, -1.0)
    a, _, _ = O.dls_batch_constrained(om, tasks, cons, tg, q0, prm, 2)
    b, _, _ = O.dls_batch_constrained(om, tasks, cons, tg, q0, prm, 2, ext="q")
    assert np.abs(a - b).max() < 1e-10
    pp = O.pik_params(20, 1.0, -1.0, [0.1, 0.1])
    a, _, _ = O.pik_batch(om, tasks, tg, q0, pp, 2)
    b, _, _ = O.pik_batch(om, tasks, tg, q0, pp, 2, ext="q")
    assert np.abs(a - b).max() < 1e-9
