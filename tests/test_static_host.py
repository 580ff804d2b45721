"""The static lane programs (rtc.cpp generic_static_source: the generic lane program specialised for ONE problem, what a generic
problem runs on by default) checked WITHOUT a GPU: the generated translation unit is dumped (IKGPU_RTC_DUMP), its __global__ wrapper
is replaced by a host loop over the lanes, g++ compiles it, and the results are compared with the C oracle -- the same source the
device compiles, with the host arms of the reciprocal / square-root helpers (lane_math.hpp).  Covers the dense solve, the constraint
projection and the eliminated-posture solve (TB::elim) of the reference demo with every line switched on."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

import oracle as O
from test_gpu_generic import CASES as GENERIC, build
from test_gpu_constraints import CASES as CONSTRAINED

HOST_DRIVER = r'''
#include "primal_solver.hpp"
template <class TB>
void one_lane(const ikdev::GenericKernelArgs &a, long long gid) {
    double w[TB::ws_words];
    if constexpr (TB::primal != 0) {   // the primal tree-sparse form (device/primal_solver.hpp): its LDS slab is a plain array here
        double slab[TB::lds_words];
        ikdev::dls_primal_body_ws(a, TB{}, gid, ikdev::WsReg{w}, ikdev::LdsColumn{slab, 1}, [](bool act) { return act; });
    } else {
        ikdev::dls_generic_body_ws(a, TB{}, gid, ikdev::WsReg{w}, [](bool act) { return act; });
    }
}
extern "C" int static_host_solve(long long B, const double *q0, const double *targets, int max_it, double damping, double step, double stop_tol,
                                 double *q_out, unsigned char *ok, int *iters) {
    ikdev::GenericKernelArgs a{};
    a.prm.max_iterations = max_it; a.prm.lam2 = damping * damping; a.prm.step_length = step; a.prm.stop_sq_tol = stop_tol;
    a.layout = ikdev::LAYOUT_AOS; a.B = B; a.q0 = q0; a.targets = targets; a.q_out = q_out; a.success = ok; a.iters = iters;
    for (long long gid = 0; gid < B; ++gid) one_lane<T>(a, gid);
    return T::elim + 2 * T::primal;
}
'''


def host_program(tmp_path, problem, want_static_name=True, opt="-O1"):
    import ik_amd
    src_dir, cache = tmp_path / "src", tmp_path / "cache"
    src_dir.mkdir(), cache.mkdir()
    old = {k: os.environ.get(k) for k in ("IKGPU_RTC_DUMP", "IKGPU_CACHE_DIR", "IKGPU_DLS_KERNEL", "IKGPU_TREE_STATIC_ROWS")}
    os.environ.update(IKGPU_RTC_DUMP=str(src_dir), IKGPU_CACHE_DIR=str(cache), IKGPU_DLS_KERNEL="generic")
    try:
        try:
            name = ik_amd.precompile(problem)
        except ik_amd.capi.IkgpuError as e:
            pytest.skip("run-time compilation unavailable here: %s" % e)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if not name.endswith(",static>"):
        pytest.skip("no static program for this problem here: %s" % name)
    srcs = [f for f in glob.glob(str(src_dir / "generic_static_*.hip")) if "refill" not in os.path.basename(f)]
    assert len(srcs) == 1, srcs
    text = open(srcs[0]).read()
    cut = text.index('extern "C" __global__')
    host = tmp_path / "program.cpp"
    host.write_text(text[:cut] + HOST_DRIVER)
    lib = tmp_path / "program.so"
    dev = os.path.join(ROOT, "ik_amd", "csrc", "device")
    subprocess.check_call(["g++", opt, "-std=c++17", "-fPIC", "-shared", "-I" + dev, "-I" + os.path.join(ROOT, "ik_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-o", str(lib), str(host)])
    L = C.CDLL(str(lib))
    L.static_host_solve.restype = C.c_int
    return name, L


def solve(L, q0, tg, iters, damping, step, tol):
    B = q0.shape[0]
    q0 = np.ascontiguousarray(q0)
    tg = np.ascontiguousarray(tg)
    q, ok, it = np.empty_like(q0), np.zeros(B, np.uint8), np.zeros(B, np.int32)
    p = lambda x: C.c_void_p(x.ctypes.data)
    elim = L.static_host_solve(C.c_longlong(B), p(q0), p(tg), iters, C.c_double(damping), C.c_double(step), C.c_double(tol), p(q), p(ok), p(it))
    return q, ok, it, elim


@pytest.mark.parametrize("case,elim", [("shared_joints", 0), ("com_of_the_arm", 0), ("demo_task_set", 0), ("demo_everything_on", 1),
                                       ("demo_everything_on_unconstrained", 1), ("pelvis_with_both_feet_locked", 0)])
def test_static_program_on_the_host_matches_the_oracle(tmp_path, native_built, case, elim):
    import ik_amd
    constrained = case in CONSTRAINED
    if case == "demo_everything_on_unconstrained":
        name, ff, specs, _ = CONSTRAINED["demo_everything_on"]
        cspecs, edit = [], None
    elif constrained:
        name, ff, specs, cspecs = CONSTRAINED[case]
        edit = None
    else:
        (name, ff, specs, edit), cspecs = GENERIC[case], []
    B = 24
    ik, _, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=11, xml_edit=edit, device=False)
    for i, (f, t, r) in enumerate(cspecs):
        problem.add_frame_constraint("c%d" % i, ik_amd.FrameConstraint.create(model, f, ik_amd.KinematicType(t), r))
    kernel, L = host_program(tmp_path, problem)
    oc = O.make_tasks([(model.getFrameId(f), model.getFrameId(r), t, 0, None) for f, t, r in cspecs]) if cspecs else None
    for iters, damping, step, tol, bar in ((1, 1e-2, 1.0, -1.0, 1e-9), (3, 1e-2, 1.0, -1.0, 1e-8), (40, 1e-1, 0.5, 1e-4, 1e-6)):
        prm = O.params(iters, damping, step, tol)
        if oc is not None:
            q_ref, ok_ref, it_ref = O.dls_batch_constrained(om, ot, oc, tg, q0, prm, 1)
        else:
            q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, prm, 1)
        q, ok, it, is_elim = solve(L, q0, tg, iters, damping, step, tol)
        assert is_elim == elim, (kernel, is_elim)
        assert np.array_equal(it, it_ref) and np.array_equal(ok, ok_ref), (case, iters, it, it_ref)
        d = np.abs(q - q_ref).max()
        assert d <= bar, (case, kernel, iters, d)


# case -> (force the form with IKGPU_STATIC_FORM, expected T::primal)
PRIMAL_CASES = {
    "three_feet_frames": (None, 1), "rows_16": (None, 1), "feet_frames_beyond_the_register_solve": (None, 1), "nv_30": (None, 1),
    # forced onto the form (IKGPU_STATIC_FORM=primal; with 12 solved rows or fewer the dual program is the default):
    "posture_regulariser": ("primal", 1),       # 12 frame rows + 16 posture rows (diagonal entries of H)
    "shared_joints": ("primal", 1), "moving_reference_prismatic": ("primal", 1), "fixed_two_feet_priorities": ("primal", 1),
    "com_under_feet": (None, 0),                # a centre-of-mass task: rows dense over every direction -- the dual program keeps it
}


@pytest.mark.parametrize("case", sorted(PRIMAL_CASES))
def test_primal_tree_sparse_program_on_the_host_matches_the_oracle(tmp_path, native_built, monkeypatch, case):
    """The primal, tree-sparse form of the static program (device/primal_solver.hpp: what more than 12 solved rows run on): per-task
    path FK, blocks accumulated into the no-fill normal matrix, leaf-to-root elimination with parked columns, against the oracle's
    dense dual solve.  Bars: one step 1e-9 (kappa(H) u ~ 1e-11 of the step), the converging runs 1e-6."""
    force, want = PRIMAL_CASES[case]
    name, ff, specs, edit = GENERIC[case]
    if force:
        monkeypatch.setenv("IKGPU_STATIC_FORM", force)
    B = 24
    ik, _, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=11, xml_edit=edit, device=False)
    kernel, L = host_program(tmp_path, problem, opt="-O0")    # (the host compiler spends a minute per program on the inlined loops at -O1)
    for iters, damping, step, tol, bar in ((1, 1e-2, 1.0, -1.0, 1e-9), (3, 1e-2, 1.0, -1.0, 1e-7), (40, 1e-1, 0.5, 1e-4, 1e-6), (0, 1e-2, 1.0, 1e-4, 0.0)):
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), 1)
        q, ok, it, flags = solve(L, q0, tg, iters, damping, step, tol)
        assert flags >> 1 == want, (kernel, flags)
        assert np.array_equal(it, it_ref) and np.array_equal(ok, ok_ref), (case, iters, it, it_ref)
        d = np.abs(q - q_ref).max()
        assert d <= bar, (case, kernel, iters, d)


# ---- ik::pik (reference ik/ik/pik.cpp:31-103) on its compiled lane program (device/pik_solver.hpp static_pik) --------------------------
PIK_DRIVER = r"""
extern "C" int static_host_pik(long long B, const double *q0, const double *targets, int max_it, double step, double stop_tol, int nl,
                               const double *lambda, const double *da, double *q_out, unsigned char *ok, int *iters) {
    ikdev::PikKernelArgs a{};
    a.prm.max_iterations = max_it; a.prm.step_length = step; a.prm.stop_sq_tol = stop_tol;
    for (int l = 0; l < ikdev::kMaxPikLevels; ++l) a.prm.lam2[l] = l < nl ? lambda[l] * lambda[l] : 1.0;
    a.prm.has_da = da != nullptr;
    if (da) for (int k = 0; k < T::nv; ++k) a.prm.da[k] = da[k];
    a.layout = ikdev::LAYOUT_AOS; a.B = B; a.q0 = q0; a.targets = targets; a.q_out = q_out; a.success = ok; a.iters = iters;
    for (long long gid = 0; gid < B; ++gid) {
        double w[T::ws_words];
        if (da) ikdev::pik_static_body<true>(a, T{}, gid, ikdev::WsReg{w}, [](bool act) { return act; });
        else ikdev::pik_static_body<false>(a, T{}, gid, ikdev::WsReg{w}, [](bool act) { return act; });
    }
    return T::pik_basis_rows;
}
"""


def pik_host_program(tmp_path, problem):
    """ikgpu_problem_precompile also compiles the ik::pik program of a problem with several priority levels; its dumped source with
    the __global__ wrapper replaced by a host loop."""
    import ik_amd
    src_dir, cache = tmp_path / "src", tmp_path / "cache"
    src_dir.mkdir(), cache.mkdir()
    old = {k: os.environ.get(k) for k in ("IKGPU_RTC_DUMP", "IKGPU_CACHE_DIR")}
    os.environ.update(IKGPU_RTC_DUMP=str(src_dir), IKGPU_CACHE_DIR=str(cache))
    try:
        try:
            ik_amd.precompile(problem)
        except ik_amd.capi.IkgpuError as e:
            pytest.skip("run-time compilation unavailable here: %s" % e)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    srcs = glob.glob(str(src_dir / "pik_static_*.hip"))
    if not srcs:
        pytest.skip("no compiled ik::pik program for this problem here")
    assert len(srcs) == 1, srcs
    text = open(srcs[0]).read()
    host = tmp_path / "pik_program.cpp"
    host.write_text(text[:text.index('extern "C" __global__')] + PIK_DRIVER)
    lib = tmp_path / "pik_program.so"
    dev = os.path.join(ROOT, "ik_amd", "csrc", "device")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + dev, "-I" + os.path.join(ROOT, "ik_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-o", str(lib), str(host)])
    L = C.CDLL(str(lib))
    L.static_host_pik.restype = C.c_int
    return L


@pytest.mark.parametrize("case", ["ur5_pos_then_ori", "ur5_full_then_elbow", "fixed_two_feet", "feet_then_pelvis", "demo_two_levels"])
def test_static_pik_program_on_the_host_matches_the_oracle(tmp_path, native_built, case):
    """The level loop with the projector in factored form and each level's damped pseudo-inverse as a dual Cholesky solve, against
    the oracle's ik::pik (Jacobi SVD + complete orthogonal decomposition, oracle/ik_oracle.c iko_pik)."""
    from test_gpu_pik import PIK_CASES
    name, ff, specs, edit, projector_determined = PIK_CASES[case]
    B = 24
    ik, _, model, problem, _, om, ot, q0, tg = build(name, ff, specs, B, seed=4, xml_edit=edit, device=False)
    levels = problem.max_priority_level() + 1
    L = pik_host_program(tmp_path, problem)
    p = lambda x: C.c_void_p(x.ctypes.data)
    q0c, tgc = np.ascontiguousarray(q0), np.ascontiguousarray(tg)
    for iters, step, tol, lam, da in ((1, 1.0, -1.0, [1.0] * levels, None), (4, 1.0, -1.0, [0.1] * levels, None),
                                      (30, 0.5, 1e-8, [0.05, 0.1, 0.2][:levels], None),
                                      (6, 1.0, -1.0, [0.1] * levels, 0.01 * np.cos(np.arange(model.nv)))):
        if da is not None and not projector_determined:
            continue
        q, ok, it = np.empty_like(q0c), np.zeros(B, np.uint8), np.zeros(B, np.int32)
        lam_a = np.asarray(lam, float)
        L.static_host_pik(C.c_longlong(B), p(q0c), p(tgc), iters, C.c_double(step), C.c_double(tol), levels, p(lam_a),
                          p(np.ascontiguousarray(da)) if da is not None else None, p(q), p(ok), p(it))
        q_ref, ok_ref, it_ref = O.pik_batch(om, ot, tg, q0, O.pik_params(iters, step, tol, lam, None if da is None else list(da)), 1)
        assert np.array_equal(ok, ok_ref) and np.array_equal(it, it_ref), (case, iters)
        assert np.abs(q - q_ref).max() <= 1e-6, (case, iters, np.abs(q - q_ref).max())
