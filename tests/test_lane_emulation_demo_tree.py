"""The tree lane program's general build with what the reference's demo adds to pose tasks (ik_ros/src/cassie.cpp:45-81): a chain
task whose reference frame rides on the floating base, and an AlignAxisTask row on a chain task's own frame
(device/tree_solver.hpp: TreeParams::ref_base / align_*).  Compiled for the host and run lane after lane against the C oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from test_lane_emulation import _generic_case, emu, run  # noqa: F401  (emu is a fixture)

CASES = {
    # the demo itself: foot position w.r.t. the pelvis, pelvis pose, foot Y axis along a direction
    "demo": [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None), ("LeftFootFront", "universe", 4, 0, None)],
    # the alignment row one priority level down (it then leaves the stop test), weighted
    "align_at_priority_1": [("LeftFootFront", "universe", 2, 0, None), ("pelvis", "universe", 2, 0, None),
                            ("LeftFootFront", "universe", 5, 1, [0.5])],
    # two chains, both given in the pelvis frame, the row on the second chain, no base task
    "two_chains_in_the_pelvis_frame": [("LeftFootFront", "pelvis", 2, 0, None), ("RightFootFront", "pelvis", 0, 0, [1.0, 2.0, 0.5]),
                                       ("RightFootFront", "universe", 3, 0, None)],
    # reference on the base but not the base link's own frame: a frame welded to the pelvis with an offset
    "reference_with_an_offset": [("LeftFootFront", "vectornav", 2, 0, None), ("RightFootFront", "universe", 2, 0, None),
                                 ("pelvis", "universe", 1, 0, None)],
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_tree_program_with_the_demo_extras_matches_oracle(emu, case):  # noqa: F811
    import ik_amd
    from ik_amd import capi
    specs = CASES[case]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, specs, B, seed=5)
    # the host analysis picks the tree kernel for these
    problem_tasks = (capi.Task * len(tasks))(*tasks)
    buf = C.create_string_buffer(160)
    capi.check(capi.lib().ikgpu_problem_plan(model._h, problem_tasks, len(tasks), buf, len(buf)))
    assert buf.value.decode().startswith("dls_tree<NJ=7,"), buf.value
    nt = len(tasks)
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (60, 1e-1, 0.3, 1e-6), (200, 1e-1, 1e-1, 1e-4)):
        prm = capi.DlsParams(iters, damping, step, tol)
        qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=1, ntasks=nt)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters, np.abs(qo - q_ref).max())
    # the forced generic program agrees as well
    qg, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=3, ntasks=nt)
    assert np.abs(qg - qo).max() < 1e-8
    assert ik_amd is not None


@pytest.mark.parametrize("case", ["demo", "two_chains_in_the_pelvis_frame"])
def test_device_general_build_with_the_demo_extras(emu, monkeypatch, case):  # noqa: F811
    """The same problems through the device's general build (SPEC = 0: compile-time "skip nothing", sin / cos by dsincos_fast),
    the build the GPU runs for them; the test above runs the all-runtime build (SPEC = -1)."""
    from ik_amd import capi
    specs = CASES[case]
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, specs, 24, seed=6)
    monkeypatch.setenv("LANE_EMU_TRIG", "0")
    for iters, damping, step, tol in ((1, 1e-2, 1.0, -1.0), (60, 1e-1, 0.3, 1e-6)):
        prm = capi.DlsParams(iters, damping, step, tol)
        qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=1, ntasks=len(tasks))
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(qo - q_ref).max() < 1e-8, (case, iters)


# PostureTask rows (reference ik/ik/posture.hpp:51-68) in the tree kernel's posture build: rows on chain joints join the chain's
# normal equations, rows on joints outside the chains are 1x1 systems stepped in place (device/tree_solver.hpp)
POSTURE_CASES = {
    # the demo with the regulariser its source declares and leaves commented out (ik_ros/src/cassie.cpp:63-64,76): priority 1
    "demo_with_posture": [("LeftFootFront", "pelvis", 0, 0, None), ("pelvis", "universe", 2, 0, None), ("LeftFootFront", "universe", 4, 0, None),
                          ("@posture", 16, 6, 1, ([1.0] * 16, [1.0] * 16))],
    # both legs as chains, the rows at priority 0 (they enter the stop test), uneven weights, a mask with holes
    "two_chains_posture_in_the_stop_test": [("LeftFootFront", "universe", 2, 0, None), ("RightFootFront", "universe", 2, 0, None),
                                            ("@posture", 16, 6, 0, ([0.05 + 0.02 * k for k in range(16)], [0.0 if k in (2, 7, 12) else 1.0 for k in range(16)]))],
    # a PostureTask over the last nine joints only (the right leg and one joint of the left), one chain, no base task
    "posture_on_a_suffix_of_the_joints": [("LeftFootFront", "universe", 2, 0, None), ("@posture", 9, 6, 0, ([0.3] * 9, [1.0] * 9))],
}


@pytest.mark.parametrize("case", sorted(POSTURE_CASES))
def test_tree_program_with_posture_rows_matches_oracle(emu, monkeypatch, case):  # noqa: F811
    from ik_amd import capi
    specs = POSTURE_CASES[case]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case("cassie", True, specs, B, seed=8)
    q0[::5, 15] += 3.0      # a joint outside every chain, beyond its limit: clamped by the first step that is taken
    nt = len(tasks)
    buf = C.create_string_buffer(160)
    capi.check(capi.lib().ikgpu_problem_plan(model._h, (capi.Task * nt)(*tasks), nt, buf, len(buf)))
    assert buf.value.decode().startswith("dls_tree<NJ=7,") and buf.value.decode().endswith(",posture>"), buf.value
    for env in (None, "0"):      # the all-runtime build, then the device's posture build (SPEC = 1 << kSpecPost)
        if env is not None:
            monkeypatch.setenv("LANE_EMU_TRIG", env)
        for iters, damping, step, tol in ((0, 1e-2, 1.0, 1e-4), (1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (60, 1e-1, 0.3, 1e-6), (200, 1e-1, 1e-1, 1e-4)):
            prm = capi.DlsParams(iters, damping, step, tol)
            qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=1, ntasks=nt)
            q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
            assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, env, iters)
            assert np.abs(qo - q_ref).max() < 1e-8, (case, env, iters, np.abs(qo - q_ref).max())
    monkeypatch.delenv("LANE_EMU_TRIG")
    # problem-major and component-major inputs give the same bits (the outside joints live in the q_out column meanwhile)
    prm = capi.DlsParams(25, 1e-1, 0.5, 1e-6)
    qa, oka, ita, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=1, ntasks=nt)
    qs, oks, its, *_ = run(emu, urdf, tasks, 0, np.ascontiguousarray(q0.T), np.ascontiguousarray(tg.transpose(1, 2, 0)), prm, model.nv, M,
                           layout=0, root=1, ntasks=nt)
    assert np.array_equal(qs.T, qa) and np.array_equal(oks, oka) and np.array_equal(its, ita)
    # the forced generic program agrees
    qg, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=3, ntasks=nt)
    assert np.abs(qg - qa).max() < 1e-8



# Fixed-base models on the tree program (TreeParams::fixed_base: the base block is solved and dropped)
FIXED_BASE_CASES = {
    "two_feet": ("cassie_fixed", [("LeftFootFront", "universe", 2, 0, None), ("RightFootFront", "universe", 2, 0, None)], "dls_tree<NJ=7,chains=2,fixed_base>"),
    "leg_with_alignment_and_posture": ("cassie_fixed", [("LeftFootFront", "universe", 0, 0, None), ("LeftFootFront", "universe", 5, 0, None),
                                                        ("@posture", 16, 6, 1, ([0.1] * 16, [1.0] * 16))],
                                       "dls_tree<NJ=7,chains=1,align_axis,posture,fixed_base>"),
    "arm_position_with_posture": ("ur5", [("@posture", 4, 6, 0, ([1.0] * 4, [1.0] * 4)), ("tool0", "universe", 0, 0, None)],
                                  "dls_tree<NJ=6,chains=1,posture,fixed_base>"),
}


@pytest.mark.parametrize("case", sorted(FIXED_BASE_CASES))
def test_tree_program_on_a_fixed_base_matches_oracle(emu, monkeypatch, case):  # noqa: F811
    from ik_amd import capi
    name, specs, kernel = FIXED_BASE_CASES[case]
    B = 24
    urdf, model, om, tasks, ot, q0, tg, M = _generic_case(name, False, specs, B, seed=9)
    nt = len(tasks)
    buf = C.create_string_buffer(160)
    capi.check(capi.lib().ikgpu_problem_plan(model._h, (capi.Task * nt)(*tasks), nt, buf, len(buf)))
    assert buf.value.decode() == kernel
    for env in (None, "0"):      # the all-runtime build, then the device's general / posture build
        if env is not None:
            monkeypatch.setenv("LANE_EMU_TRIG", env)
        for iters, damping, step, tol in ((0, 1e-2, 1.0, 1e-4), (1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (60, 1e-1, 0.3, 1e-6)):
            prm = capi.DlsParams(iters, damping, step, tol)
            qo, ok, it, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=0, ntasks=nt)
            q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol))
            assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, env, iters)
            assert np.abs(qo - q_ref).max() < 1e-8, (case, env, iters, np.abs(qo - q_ref).max())
    monkeypatch.delenv("LANE_EMU_TRIG")
    qg, *_ = run(emu, urdf, tasks, 0, q0, tg, prm, model.nv, M, root=2, ntasks=nt)       # the forced generic program agrees
    assert np.abs(qg - qo).max() < 1e-8
