"""ik::pik in the cooperative LDS-resident form (device/pik_coop.hpp: Cholesky for the damped step, pivoted Gram-Schmidt for the
projector), compiled for the host and run with one lane that takes every item of a phase, against the C oracle (iko_pik:
one-sided Jacobi SVD, column-pivoted Householder QR) and against the per-lane program (pik_solver.hpp: one-sided Jacobi)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from test_lane_emulation import PIK_CASES, _generic_case, emu, pik_prm, run_pik  # noqa: F401  (emu is a fixture)


def run_pik_coop(L, urdf, tasks, q0, tg, prm, root=0):
    B = q0.shape[0]
    qo = np.empty_like(q0)
    ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.lane_emu_pik_coop(urdf, C.c_size_t(len(urdf)), root, tasks, len(tasks), C.c_int64(B), p(q0), p(tg), C.byref(prm), p(qo), p(ok), p(it), 1)
    if rc == 2:
        pytest.skip("four PIK workspaces of this problem do not fit 64 KB of LDS: the per-lane program runs it")
    assert rc == 0, L.lane_emu_last_error()
    return qo, ok, it


@pytest.mark.parametrize("projector", ["factored", "dense"])
@pytest.mark.parametrize("case", sorted(PIK_CASES))
def test_cooperative_pik_program_matches_oracle(emu, monkeypatch, case, projector):  # noqa: F811
    """projector: the default keeps P = I - V^T V as the stacked orthonormal bases V of the levels (where the coefficients fit
    behind V, see generic_tables.hpp); IKGPU_PIK_PROJECTOR=dense forms the nv x nv matrix as the reference does."""
    monkeypatch.setenv("IKGPU_PIK_PROJECTOR", projector)
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
        prm = pik_prm(iters, step, tol, lam, da)
        qo, ok, it = run_pik_coop(emu, urdf, tasks, q0, tg, prm, root=root)
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, da))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())
        ql, okl, itl = run_pik(emu, urdf, tasks, q0, tg, prm, root=root)   # the per-lane program
        assert np.array_equal(ok, okl) and np.array_equal(it, itl) and np.abs(qo - ql).max() < 1e-8
