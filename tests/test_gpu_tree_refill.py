"""Lane refill of the free-flyer tree kernels (device/tree_kernel_body.hpp TreeRefill, kernels_tree_refill.hip): what a stop-rule solve
of a tree problem runs BY DEFAULT once the batch exceeds the lanes the device keeps resident (kernels.hip refill_wanted).  A lane whose
visitor fired (reference ik/ik/visitor.hpp:15-21, ik/ik/dls.cpp:61-64) or whose count reached max_iterations (dls.cpp:76-77) stores
its result and takes the next unsolved problem; the rest of q is clipped afterwards by the pass-through kernel (common.hpp:53-56).

Every lock-step build of the tree kernel that has a refill twin is covered -- the hot build (BASELINE.json's config 3: both feet + the
pelvis), the mask-only build (weights / task types general), the general build with the placement mask folded (the reference demo's
task set with its base-relative reference and alignment row, ik_ros/src/cassie.cpp:45-81; a fixed-base model with both feet), one
chain and two, and the general build without a folded mask (a free-floating 7-joint arm that shares nothing with Cassie's structure):
  * IKGPU_REFILL in {0, unset, 1}: q, success and iterations bit-identical at B = 300 017 (default visitor, max_iterations = 100);
  * ragged sizes {1, 63, 65, 1000, 65536 + 67, 65536 + 64*3 + 5}, both layouts, max_iterations in {1, 2, 100}, optional outputs null;
  * q, flags and iteration counts against the CPU oracle on 8192 problems, with a block whose visitor fires at iteration 0 (the
    returned q is the unclipped q0, dls.cpp:61-63) and entries outside every chain pushed beyond their limits."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import urdf_path
from test_gpu_refill import env

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


# name -> (urdf, free flyer, [(kind, frame, reference, type / axis, priority, weights)], kernel name, damping, step length)
# (full steps at damping 1e-2 are chaotic on the demo's task set beyond a few iterations -- tests/test_gpu_generic.py -- so that
# case takes the demo's own damping and a half step)
CASES = {
    "full_body": ("cassie", True, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                   ("frame", "pelvis", "universe", 2, 0, None)], "dls_tree<NJ=7,chains=2,base_task>", 1e-2, 1.0),
    "full_body_weighted": ("cassie", True, [("frame", "LeftFootFront", "universe", 0, 0, [2.0, 1.0, 0.5]),
                                            ("frame", "RightFootFront", "universe", 2, 0, [1, 1, 1, 0.3, 0.3, 0.3]),
                                            ("frame", "pelvis", "universe", 2, 0, None)], "dls_tree<NJ=7,chains=2,base_task>", 1e-2, 1.0),
    "leg_and_pelvis": ("cassie", True, [("frame", "RightFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 2, 0, None)],
                       "dls_tree<NJ=7,chains=1,base_task>", 1e-2, 1.0),
    "demo_task_set": ("cassie", True, [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                       ("align", "LeftFootFront", "universe", 1, 0, None)],
                      "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis>", 1e-1, 0.5),
    "fixed_two_feet": ("cassie_fixed", False, [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None)],
                       "dls_tree<NJ=7,chains=2,fixed_base>", 1e-2, 1.0),
    "floating_arm7": ("arm7", True, [("frame", "tool", "universe", 2, 0, None), ("frame", "base", "universe", 2, 0, None)],
                      "dls_tree<NJ=7,chains=1,base_task>", 1e-2, 1.0),
}


def _problem(torch, case, B, mode, seed=0):
    """The problem on the tree kernel, its inputs on the device (SoA) and what the oracle needs: targets are the task frames' poses at a
    reachable q* (expressed in each task's reference frame; the alignment row asks for the foot's axis at q*)."""
    import ik_amd
    import oracle as O
    from ik_amd import workload
    name, ff, specs, kernel, damping, step = CASES[case]
    model = ik_amd.Model.from_urdf_file(urdf_path(name), free_flyer=ff)
    problem = ik_amd.InverseKinematicsProblem(model)
    ospec = []
    for i, (kind, f, r, t, p, w) in enumerate(specs):
        if kind == "align":
            task = problem.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(model, f, ik_amd.AlignAxisType(t), r), p)
        else:
            task = problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t), r), p)
        if w is not None:
            task.weighting()[:] = w
        ospec.append((model.getFrameId(f), model.getFrameId(r), 3 + t if kind == "align" else t, p, w))
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel == kernel, data.kernel
    lo, hi = model.lowerPositionLimit, model.upperPositionLimit
    idx = np.arange(B)
    if ff:
        nominal = np.zeros(model.nq - 7) if name == "arm7" else workload.cassie_nominal(model.names)
        q0, qs = workload.freeflyer_workload(lo, hi, nominal, idx, seed=seed, mode=mode)
    else:
        q0, qs = workload.chain_workload(lo, hi, workload.cassie_nominal(model.names), idx, seed, mode)
    om = O.OracleModel(model.flat())
    tg = _targets(O, om, ospec, qs)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    return ik_amd, O, model, problem, data, om, O.make_tasks(ospec), ospec, q0, tg, Q0, T, damping, step


def _targets(O, om, ospec, qs):
    B = qs.shape[0]
    frames = sorted({s[0] for s in ospec} | {s[1] for s in ospec})
    poses = O.fk_batch(om, qs, frames)                       # [B, nframes, 12], world
    at = {f: k for k, f in enumerate(frames)}
    tg = np.zeros((B, len(ospec), 12))
    for i, (fid, rid, typ, _, _) in enumerate(ospec):
        Rf, pf = poses[:, at[fid], :9].reshape(B, 3, 3), poses[:, at[fid], 9:]
        Rr, pr = poses[:, at[rid], :9].reshape(B, 3, 3), poses[:, at[rid], 9:]
        if typ >= 3:      # AlignAxisTask: the direction (in the reference frame) of the frame's axis at q*
            tg[:, i, :9] = np.eye(3).ravel()
            tg[:, i, 9:] = np.einsum("bki,bk->bi", Rr, Rf[:, :, typ - 3])
        else:             # rMf = oMr^-1 oMf
            tg[:, i, :9] = np.einsum("bki,bkj->bij", Rr, Rf).reshape(B, 9)
            tg[:, i, 9:] = np.einsum("bki,bk->bi", Rr, pf - pr)
    return tg


def _solve(ik_amd, problem, data, Q0, T, refill, damping, step, max_it=100, tol=1e-4, layout="soa"):
    with env(IKGPU_REFILL=refill):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol),
                                     ik_amd.dls_parameters(max_iterations=max_it, damping=damping, step_length=step), layout=layout)
    return Q.cpu().numpy(), ok.cpu().numpy(), it.cpu().numpy()


@pytest.mark.parametrize("case", sorted(CASES))
def test_tree_refill_is_bit_identical_to_lock_step_on_a_batch_larger_than_the_machine(torch_cuda, case):
    torch = torch_cuda
    B = 300000 + 17
    ik_amd, O, model, problem, data, om, ot, ospec, q0, tg, Q0, T, damping, step = _problem(torch, case, B, "uniform")
    a = _solve(ik_amd, problem, data, Q0, T, "0", damping, step)
    b = _solve(ik_amd, problem, data, Q0, T, None, damping, step)      # the default policy: B > resident lanes -> two phases (lock-step, then refill)
    c = _solve(ik_amd, problem, data, Q0, T, "1", damping, step)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z), (case, data.kernel)
    assert 0 < a[1].mean() and np.unique(a[2]).size > 3, "the workload should spread the outcomes"
    print("%s [%s]: %d problems, success %.4f, mean iterations %.2f, max %d" % (case, data.kernel, B, a[1].mean(), a[2].mean(), a[2].max()))


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("case", ["full_body", "full_body_weighted", "leg_and_pelvis", "demo_task_set", "fixed_two_feet", "floating_arm7"])
def test_tree_refill_forced_at_small_and_ragged_batches(torch_cuda, case, layout):
    torch = torch_cuda
    sizes = (1, 63, 65, 1000, 65536 + 67, 65536 + 64 * 3 + 5)
    ik_amd, O, model, problem, data, om, ot, ospec, q0, tg, Q0, T, damping, step = _problem(torch, case, max(sizes), "uniform", seed=4)
    for B in sizes:
        if layout == "aos":
            q, t = Q0[:, :B].t().contiguous(), T[:, :, :B].permute(2, 0, 1).contiguous()
        else:
            q, t = Q0[:, :B].contiguous(), T[:, :, :B].contiguous()
        for max_it in ((1, 2, 5, 100) if B <= 1000 else (100,)):
            a = _solve(ik_amd, problem, data, q, t, "0", damping, step, max_it=max_it, layout=layout)
            b = _solve(ik_amd, problem, data, q, t, "1", damping, step, max_it=max_it, layout=layout)
            c = _solve(ik_amd, problem, data, q, t, "2", damping, step, max_it=max_it, layout=layout)   # two phases (kernels.hpp stop_rule_mode)
            for x, y, z in zip(a, b, c):
                assert np.array_equal(x, y) and np.array_equal(x, z), (case, B, layout, max_it)
            assert (a[2] <= max_it).all() and (a[2][a[1] == 0] == max_it).all()
            if max_it == 100 and B in (65, 65536 + 67):   # the switch point moves the work between the phases, never the bits
                for after, active in (("1", "63"), ("16", "1")):
                    with env(IKGPU_TWO_PHASE_ITERS=after, IKGPU_TWO_PHASE_ACTIVE=active):
                        d = _solve(ik_amd, problem, data, q, t, "2", damping, step, max_it=max_it, layout=layout)
                    for x, y in zip(a, d):
                        assert np.array_equal(x, y), (case, B, layout, after, active)


@pytest.mark.parametrize("case", ["full_body", "demo_task_set"])
def test_tree_refill_without_the_optional_outputs(torch_cuda, case):
    """success / iters may be NULL (include/ikgpu.h): the refill launch then keeps its own iteration counts for the pass-through step."""
    torch = torch_cuda
    from ik_amd import capi
    B = 70001
    ik_amd, O, model, problem, data, om, ot, ospec, q0, tg, Q0, T, damping, step = _problem(torch, case, B, "uniform", seed=2)
    ref = _solve(ik_amd, problem, data, Q0, T, "0", damping, step)
    Q = torch.full_like(Q0, float("nan"))
    prm = capi.DlsParams(100, damping, step, 1e-4)
    for mode in ("1", "2"):
        Q.fill_(float("nan"))
        with env(IKGPU_REFILL=mode):
            capi.check(capi.lib().ikgpu_dls_solve_batch(data._h, B, Q0.data_ptr(), T.data_ptr(), C.byref(prm), Q.data_ptr(), None, None, capi.SOA,
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert np.array_equal(Q.cpu().numpy(), ref[0]), mode


@pytest.mark.parametrize("case", ["full_body", "full_body_weighted", "demo_task_set", "fixed_two_feet"])
def test_tree_refill_against_the_oracle_and_iteration_zero_stops(torch_cuda, case):
    """Default visitor, max_iterations = 100, through the refill kernel (forced: 8192 problems fit the machine).  Problems 0..999: the
    target IS the start pose, so the visitor fires at iteration 0 and the reference returns q0 untouched -- also the entries outside
    every chain, which sit BEYOND their limits here (dls.cpp:61-63: no step, no clipping).  Problems 1000..4095: targets near the
    start (every lane converges within a few iterations; a clipped outside entry each).  The rest: targets anywhere in the limits --
    the lanes whose visitor fires are held to 1e-6 rad with equal iteration counts; the lanes that run out of iterations are held to
    flags and counts here and to the float128-arbitrated rule in tests/test_gpu_full_size.py (a non-converging trajectory amplifies
    rounding differences; refill returns the lock-step kernel's bits for them, asserted above)."""
    torch = torch_cuda
    B, NZ, NN = 8192, 1000, 4096
    ik_amd, O, model, problem, data, om, ot, ospec, q0, tg, Q0, T, damping, step = _problem(torch, case, B, "uniform", seed=1)
    _, _, _, _, _, _, _, _, q0n, tgn, _, _, _, _ = _problem(torch, case, NN, "near", seed=1)
    assert np.array_equal(q0n, q0[:NN])
    tg[:NN] = tgn
    outside = np.flatnonzero(~data.support)
    outside = outside[outside >= (7 if CASES[case][1] else 0)]
    assert outside.size, "the case should have entries of q no task moves"
    q0[::3, outside[0]] = model.upperPositionLimit[outside[0]] + 0.05
    tg[:NZ] = _targets(O, om, ospec, q0[:NZ])
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    q_gpu, ok, it = _solve(ik_amd, problem, data, Q0, T, "1", damping, step)
    q_gpu = q_gpu.T
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(100, damping, step, 1e-4), os.cpu_count() or 1)
    assert (it[:NZ] == 0).all() and ok[:NZ].all() and np.array_equal(q_gpu[:NZ], q0[:NZ])      # unclipped, untouched
    assert (it_ref[:NZ] == 0).all()
    same = it == it_ref
    assert same.mean() > 0.999 and np.array_equal(ok[same], ok_ref[same]), (case, same.mean())
    fired = same & (ok_ref != 0)
    assert fired[NZ:NN].mean() > 0.99, (case, fired[NZ:NN].mean())
    d = np.abs(q_gpu - q_ref).max(axis=1)
    assert d[fired].max() <= TOL, (case, d[fired].max())
    # entries no task moves: q0 when the solve stopped at iteration 0, else q0 clipped (exactly)
    clipped = np.clip(q0[:, outside], model.lowerPositionLimit[outside], model.upperPositionLimit[outside])
    want = np.where((it == 0)[:, None], q0[:, outside], clipped)
    assert np.array_equal(q_gpu[:, outside], want)
    ran_out = same & (ok_ref == 0)
    print("%s [%s]: iterations equal on %.5f of %d problems; visitor fired on %d (max |dq| %.2e rad), %d ran to max_iterations (median |dq| %.2e)"
          % (case, data.kernel, same.mean(), B, fired.sum(), d[fired].max(), ran_out.sum(), np.median(d[ran_out]) if ran_out.any() else 0.0))
