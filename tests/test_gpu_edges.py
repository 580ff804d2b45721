"""GPU edge cases of the hot path through the C ABI: batch sizes around the wave width, one problem, zero
iterations, huge but finite angles, unreachable targets, large damping, streams, and HIP-graph capture of the
launch (the solve entry point allocates nothing and never synchronises, so it can be captured and replayed)."""
import os

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def ctx(native_built):
    import torch
    import ik_amd
    import oracle as O
    from ik_amd import workload
    assert torch.cuda.is_available()
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("lf", ik_amd.FrameTask.create(model, "LeftFootFront"))
    data = ik_amd.dls_data(problem)
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("LeftFootFront")
    ot = O.make_tasks([(fid, 0, 2, 0, None)])

    def inputs(B, mode="near", seed=0):
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                         workload.cassie_nominal(model.names), np.arange(B), seed, mode)
        return q0, O.fk_batch(om, qs, [fid])

    def dev(q0, tg):
        return (torch.from_numpy(np.ascontiguousarray(q0.T)).cuda(), torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda())
    return dict(torch=torch, ik=ik_amd, O=O, model=model, problem=problem, data=data, om=om, ot=ot, inputs=inputs, dev=dev)


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 127, 129, 4095])
def test_batch_sizes_around_the_wave_width(ctx, B):
    ik, O = ctx["ik"], ctx["O"]
    q0, tg = ctx["inputs"](B)
    Q0, T = ctx["dev"](q0, tg)
    Q, ok, it = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])
    q_ref, ok_ref, it_ref = O.dls_batch(ctx["om"], ctx["ot"], tg, q0, O.params())
    assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref)
    assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL


def test_zero_iterations_returns_q0_untouched(ctx):
    ik = ctx["ik"]
    q0, tg = ctx["inputs"](100)
    q0[:, 9] += 7.0  # outside its limit: must come back unclipped (reference ik/ik/dls.cpp:14,76-77 with no iteration)
    Q0, T = ctx["dev"](q0, tg)
    Q, ok, it = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"], p=ik.dls_parameters(max_iterations=0))
    assert ctx["torch"].equal(Q, Q0) and not ok.any() and not it.any()


def test_large_angles_unreachable_targets_large_damping(ctx):
    ik, O = ctx["ik"], ctx["O"]
    q0, tg = ctx["inputs"](512, "uniform", seed=4)
    q0[:, :8] += 2.0 * np.pi * np.arange(512)[:, None] % 37      # winds up to ~230 rad: the sincos range reduction
    tg[::2, 0, 9:] += [3.0, -2.0, 4.0]                            # far outside the workspace
    Q0, T = ctx["dev"](q0, tg)
    for iters, damping, step in ((1, 1e-2, 1.0), (2, 5.0, 0.3), (1, 1e-3, 1.0)):
        Q, ok, it = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"], ik.never_stop_visitor(),
                                 ik.dls_parameters(max_iterations=iters, damping=damping, step_length=step))
        q_ref, _, _ = O.dls_batch(ctx["om"], ctx["ot"], tg, q0, O.params(iters, damping, step, -1.0))
        assert np.isfinite(Q.cpu().numpy()).all()
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL
    # a vanishing damping makes JJ^T + damping^2 I numerically singular for unreachable targets: the step is then
    # dominated by rounding on both sides (Eigen's pivoted LDL^T in the reference, Cholesky here); it must stay finite
    Q, _, _ = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"], ik.never_stop_visitor(), ik.dls_parameters(max_iterations=3, damping=1e-6))
    assert np.isfinite(Q.cpu().numpy()).all()


def test_concurrent_streams_share_one_problem_handle(ctx):
    torch, ik = ctx["torch"], ctx["ik"]
    q0, tg = ctx["inputs"](8192)
    Q0, T = ctx["dev"](q0, tg)
    ref, _, _ = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for s in streams:
        with torch.cuda.stream(s):
            outs.append(ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])[0])
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)


def test_hip_graph_capture_and_replay(ctx):
    """The chain / tree solve entry points are graph-capturable: one capture, many replays, same bits."""
    torch, ik = ctx["torch"], ctx["ik"]
    q0, tg = ctx["inputs"](4096)                                   # BASELINE.json configs[1]: a launch-bound size
    Q0, T = ctx["dev"](q0, tg)
    ref, ok_ref, it_ref = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])
    out = (torch.empty_like(Q0), torch.empty(4096, dtype=torch.uint8, device="cuda"), torch.empty(4096, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            ik.dls_batch(ctx["problem"], Q0, T, ctx["data"], out=out)
    out[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref) and torch.equal(out[1], ok_ref) and torch.equal(out[2], it_ref)
    T2 = T.clone()
    T.copy_(torch.roll(T2, 1, dims=2))                             # new targets in the captured buffers
    g.replay()
    torch.cuda.synchronize()
    ref2, _, _ = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])
    assert torch.equal(out[0], ref2)


def test_config4_batch_and_a_million_problems_are_batch_size_independent(ctx):
    """BASELINE.json's largest batch (config 4: 262144 problems, there spread over eight GPUs) and four times that on one GPU:
    a problem's result does not depend on the batch it travels in (bit-identical with the same problem in a batch of 4096), and
    reachable targets are reached (FK(q) = target) in every lane."""
    torch, ik = ctx["torch"], ctx["ik"]
    from ik_amd import workload
    model, problem, data = ctx["model"], ctx["problem"], ctx["data"]
    p = ik.dls_parameters(max_iterations=50)
    small = None
    for B in (4096, 262144, 1 << 20):
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names),
                                         np.arange(B), 0, "near")
        Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
        T = ik.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
        Q, ok, it = ik.dls_batch(problem, Q0, T, data, ik.never_stop_visitor(), p)
        assert (ik.task_frames_fk_batch(problem, Q, data) - T).abs().max().item() < 1e-9
        assert not ok.any() and bool((it == 50).all())
        if small is None:
            small = Q.clone()
        else:
            assert torch.equal(Q[:, :4096], small)


def test_a_nan_lane_stays_alone(ctx):
    """Lanes are independent: a problem whose q0 or target holds a NaN leaves the results of its 63 wave neighbours
    bit-identical.  (What the NaN problem itself returns is not specified by the reference: its clamp, like the oracle's and the
    kernel's, may turn a NaN joint value into a joint limit.)"""
    torch, ik = ctx["torch"], ctx["ik"]
    q0, tg = ctx["inputs"](256)
    Q0, T = ctx["dev"](q0, tg)
    Q, ok, it = ik.dls_batch(ctx["problem"], Q0, T, ctx["data"])
    Q0b, Tb = Q0.clone(), T.clone()
    Q0b[2, 70] = float("nan")        # a chain joint of problem 70
    Tb[0, 9, 133] = float("nan")     # the target translation of problem 133
    Qb, okb, itb = ik.dls_batch(ctx["problem"], Q0b, Tb, ctx["data"])
    keep = torch.ones(256, dtype=torch.bool, device="cuda")
    keep[70] = keep[133] = False
    assert torch.equal(Qb[:, keep], Q[:, keep]) and torch.equal(okb[keep], ok[keep]) and torch.equal(itb[keep], it[keep])
    assert okb[133].item() == 0      # a NaN target never passes the stop test


def test_host_pointer_calls_from_two_threads_on_one_problem(ctx):
    """The host-pointer entry points keep a staging area per problem for small batches (one thread at a time; a second caller
    takes the allocate-per-call path): results are the same either way, from any number of threads, for any batch size around the
    staging limit."""
    import threading
    ik = ctx["ik"]
    problem, data = ctx["problem"], ctx["data"]
    q0, tg = ctx["inputs"](4000)                       # 4000 problems: 4000 * (2 * 128 + 96 + 5) B > 1 MiB, beyond the staging limit
    want, ok_w, it_w = ik.dls_batch(problem, q0, tg, data, layout="aos")
    for B in (1, 7, 2000):                             # staged
        Q, ok, it = ik.dls_batch(problem, q0[:B], tg[:B], data, layout="aos")
        assert np.array_equal(Q, want[:B]) and np.array_equal(ok, ok_w[:B]) and np.array_equal(it, it_w[:B])
    errors = []

    def worker(lo):
        try:
            for k in range(40):
                b = lo + k
                Q, ok, it = ik.dls_batch(problem, q0[b:b + 1], tg[b:b + 1], data, layout="aos")
                if not (np.array_equal(Q[0], want[b]) and ok[0] == ok_w[b] and it[0] == it_w[b]):
                    errors.append(b)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(lo,)) for lo in (0, 100, 200)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("case", ["full_body_tree", "demo_with_posture_tree", "generic_cooperative", "generic_cooperative_beyond_64kb", "pik_cooperative"])
def test_hip_graph_capture_of_the_other_kernels(ctx, case, monkeypatch):
    """The tree kernels (hot build; posture build with its dynamic LDS) and the cooperative generic / PIK kernels allocate nothing
    and never synchronise either: one capture, replays with the same bits."""
    torch, ik = ctx["torch"], ctx["ik"]
    from test_gpu_generic import build
    monkeypatch.delenv("IKGPU_DLS_KERNEL", raising=False)
    if case == "generic_cooperative_beyond_64kb":   # (beyond 31 rows the dispatch prefers the per-lane form: keep the cooperative one here)
        monkeypatch.setenv("IKGPU_GENERIC_COOP_ANY_SIZE", "1")
    specs = {
        "full_body_tree": [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                           ("frame", "pelvis", "universe", 2, 0, None)],
        "demo_with_posture_tree": [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                                   ("posture", 16, None, None, 1, ([0.1] * 16, [1.0] * 16))],
        "generic_cooperative": [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                ("frame", "LeftFootBack", "universe", 0, 0, None)],
        # M = 36: four workspaces + the tables take more than the default 64 KB of dynamic LDS (the launch raises the kernel's limit once)
        "generic_cooperative_beyond_64kb": [("frame", f, "universe", 2, 0, None) for f in
                                            ("LeftFootFront", "LeftFootBack", "RightFootFront", "RightFootBack", "pelvis", "LeftFootFront")],
        "pik_cooperative": [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "pelvis", "universe", 2, 1, None)],
    }[case]
    B = 1000
    ik_amd, O, model, problem, data, om, ot, q0, tg = build("cassie", True, specs, B, seed=23)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    if case == "pik_cooperative":
        data = ik.pik_data(problem, device=0)
        data.lambda_ = [0.1, 0.1]
        solve, p = ik.pik_batch, ik.pik_parameters(max_iterations=12, step_length=0.5)
        assert data.kernel.startswith("pik_generic<")
    else:
        solve, p = ik.dls_batch, ik.dls_parameters(max_iterations=12, damping=1e-1, step_length=0.5)
        assert data.kernel.startswith({"full_body_tree": "dls_tree<NJ=7,chains=2,base_task>", "demo_with_posture_tree": "dls_tree<NJ=7,chains=1,base_task,base_reference,posture>",
                                       "generic_cooperative": "dls_generic<", "generic_cooperative_beyond_64kb": "dls_generic<M=36"}[case])
    v = ik.never_stop_visitor()
    ref, ok_ref, it_ref = solve(problem, Q0, T, data, v, p)
    out = (torch.empty_like(Q0), torch.empty(B, dtype=torch.uint8, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            solve(problem, Q0, T, data, v, p, out=out)
    out[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref) and torch.equal(out[1], ok_ref) and torch.equal(out[2], it_ref)


def test_concurrent_streams_on_the_cooperative_generic_kernel(ctx, monkeypatch):
    """The persistent workgroups of the cooperative kernels take their groups of problems from a per-launch work queue (a slot of
    a ring the launch zeroes on its stream): launches of ONE problem handle in flight on several streams must not share a head."""
    torch, ik = ctx["torch"], ctx["ik"]
    from test_gpu_generic import build
    monkeypatch.delenv("IKGPU_DLS_KERNEL", raising=False)
    specs = [("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
             ("frame", "LeftFootBack", "universe", 0, 0, None)]
    B = 6000
    ik_amd, O, model, problem, data, om, ot, q0, tg = build("cassie", True, specs, B, seed=29)
    assert data.kernel.startswith("dls_generic<")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    p = ik.dls_parameters(max_iterations=20, damping=1e-1, step_length=0.5)
    v = ik.never_stop_visitor()
    ref, ok_ref, it_ref = ik.dls_batch(problem, Q0, T, data, v, p)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for rep in range(3):
        for s in streams:
            with torch.cuda.stream(s):
                outs.append(ik.dls_batch(problem, Q0, T, data, v, p))
    torch.cuda.synchronize()
    assert all(torch.equal(o[0], ref) and torch.equal(o[2], it_ref) for o in outs)


def test_problem_support_marks_the_entries_a_solve_can_move(native_built):
    """ikgpu_problem_support (what the compact multi-GPU gather ships): the task supports and the floating base; every other entry of
    q only passes through the joint clipping -- and the solve indeed leaves those at clip(q0)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ik_amd
    from ik_amd import distributed as ikdist, workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    chain = ["LeftHipRoll", "LeftHipYaw", "LeftHipPitch", "LeftKneePitch", "LeftShinPitch", "LeftTarsusPitch", "LeftFootPitch"]
    flat = model.flat()
    want = np.zeros(model.nq, bool)
    want[[int(flat["idx_q"][model.names.index(n)]) for n in chain]] = True
    assert np.array_equal(data.support, want)
    # a consumer holding q0 rebuilds the whole configuration from the support rows
    B = 500
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 3, "near")
    q0[:, 9] += 3.0                                        # a right-leg entry beyond its limit
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    for visitor, iters in ((ik_amd.never_stop_visitor(), 20), (ik_amd.inverse_kinematics_visitor(), 100), (ik_amd.inverse_kinematics_visitor(1e30), 5)):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, visitor, ik_amd.dls_parameters(max_iterations=iters))
        rows = torch.from_numpy(np.flatnonzero(data.support)).cuda()
        full = ikdist.expand_rows(Q[rows], rows, Q0, torch.from_numpy(model.lowerPositionLimit).cuda(), torch.from_numpy(model.upperPositionLimit).cuda(), it)
        assert torch.equal(full, Q)
    # free-flyer full body: the base and both legs, not the two spring joints
    m2 = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    p2 = ik_amd.InverseKinematicsProblem(m2)
    for i, f in enumerate(["LeftFootFront", "RightFootFront", "pelvis"]):
        p2.add_frame_task("t%d" % i, ik_amd.FrameTask.create(m2, f, ik_amd.KinematicType.Full))
    d2 = ik_amd.dls_data(p2, device=0)
    f2 = m2.flat()
    springs = [int(f2["idx_q"][m2.names.index(n)]) for n in ("LeftAchillesSpring", "RightAchillesSpring")]
    assert d2.support.sum() == m2.nq - 2 and not d2.support[springs].any()
