"""Lane refill (device/chain_kernel_body.hpp chain_refill_loop): the stop-rule mode of the chain kernels for batches larger than the
machine.  A lane whose visitor fired (reference ik/ik/visitor.hpp:15-21, ik/ik/dls.cpp:61-64) or whose iteration count reached
max_iterations (dls.cpp:76-77) stores its result and takes the next unsolved problem.  Results are bit-identical to the lock-step
kernel's by construction -- asserted here on every build of the chain kernel (hot, hot-rtc, general; Full / Position tasks), in both
layouts, at batch sizes around the wave and machine boundaries, with and without the optional output arrays -- and the oracle agrees
on flags, iteration counts and q.  Also: the queue slots of the persistent kernels (kernels.hpp QueuePool) under a captured graph
replayed next to live launches on a second stream (the old 64-slot ring wrapped)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _problem(torch, name, frame, ktype, build, B, mode="uniform"):
    import ik_amd
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path(name))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ktype))
    with env(IKGPU_CHAIN_HOT="0" if build == "general" else None):
        data = ik_amd.dls_data(problem, device=0)
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else np.zeros(model.nq) if name == "arm7" else workload.cassie_nominal(model.names)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, mode)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    return model, problem, data, q0, Q0, T


def _solve(ik_amd, problem, data, Q0, T, refill, max_it=100, tol=1e-4, layout="soa"):
    with env(IKGPU_REFILL=refill):
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), ik_amd.dls_parameters(max_iterations=max_it), layout=layout)
    return Q.cpu().numpy(), ok.cpu().numpy(), it.cpu().numpy()


CASES = [
    ("cassie_fixed", "LeftFootFront", 2, "default"),   # hot
    ("cassie_fixed", "LeftFootFront", 2, "general"),
    ("cassie_fixed", "LeftFootFront", 0, "default"),   # a Position task: general builds only
    ("ur5", "tool0", 2, "default"),                    # nq == nj: no entries outside the chain
    ("arm7", "tool", 2, "default"),                    # hot-rtc (general when hipRTC is absent)
    ("arm7", "tool", 2, "general"),
]


@pytest.mark.parametrize("name,frame,ktype,build", CASES)
def test_refill_is_bit_identical_to_lock_step_on_a_batch_larger_than_the_machine(torch_cuda, name, frame, ktype, build):
    torch = torch_cuda
    import ik_amd
    B = 300000 + 17
    model, problem, data, q0, Q0, T = _problem(torch, name, frame, ik_amd.KinematicType(ktype), build, B)
    a = _solve(ik_amd, problem, data, Q0, T, "0")
    b = _solve(ik_amd, problem, data, Q0, T, None)      # the default policy: B > resident lanes -> two phases (lock-step, then refill on the unfinished)
    c = _solve(ik_amd, problem, data, Q0, T, "1")
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z), (name, build, data.kernel)
    # ... on targets near the start too (every problem is done in the first phase: the worklist is empty)
    _, _, _, _, Q0n, Tn = _problem(torch, name, frame, ik_amd.KinematicType(ktype), build, 200000 + 5, mode="near")
    an, bn = _solve(ik_amd, problem, data, Q0n, Tn, "0"), _solve(ik_amd, problem, data, Q0n, Tn, None)
    for x, y in zip(an, bn):
        assert np.array_equal(x, y), (name, build, "near")
    assert 0 < a[1].mean() < 1 or name != "cassie_fixed"     # the workload has both outcomes
    print("%s [%s]: %d problems, success %.4f, mean iterations %.2f" % (name, data.kernel, B, a[1].mean(), a[2].mean()))


def test_four_million_problems_through_the_default_policy(torch_cuda):
    """64 times the resident lanes (the worklist, the queue head and every index beyond 2^22): the default policy -- two phases --
    and the refill kernel alone against the lock-step kernel, bit for bit."""
    torch = torch_cuda
    import ik_amd
    B = (1 << 22) + 3
    model, problem, data, q0, Q0, T = _problem(torch, "cassie_fixed", "LeftFootFront", ik_amd.KinematicType.Full, "default", B)
    a = _solve(ik_amd, problem, data, Q0, T, "0")
    for mode in (None, "1"):
        b = _solve(ik_amd, problem, data, Q0, T, mode)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), mode
    assert 0.9 < a[1].mean() < 1.0 and a[2].max() == 100


@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000, 65536 + 64 + 3])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_refill_forced_at_small_and_ragged_batches(torch_cuda, B, layout):
    torch = torch_cuda
    import ik_amd
    model, problem, data, q0, Q0, T = _problem(torch, "cassie_fixed", "LeftFootFront", ik_amd.KinematicType.Full, "default", B)
    if layout == "aos":
        Q0, T = Q0.t().contiguous(), T.permute(2, 0, 1).contiguous()
    for max_it in (1, 2, 5, 100):
        a = _solve(ik_amd, problem, data, Q0, T, "0", max_it=max_it, layout=layout)
        b = _solve(ik_amd, problem, data, Q0, T, "1", max_it=max_it, layout=layout)
        c = _solve(ik_amd, problem, data, Q0, T, "2", max_it=max_it, layout=layout)    # two phases: lock-step until a wave's stragglers are few, refill on those
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x, z), (B, layout, max_it)
        if max_it == 100:   # the switch point moves the work between the phases, never the bits: hand over at once / early / late
            for after, active in (("1", "63"), ("2", "48"), ("16", "1")):
                with env(IKGPU_TWO_PHASE_ITERS=after, IKGPU_TWO_PHASE_ACTIVE=active):
                    d = _solve(ik_amd, problem, data, Q0, T, "2", max_it=max_it, layout=layout)
                for x, y in zip(a, d):
                    assert np.array_equal(x, y), (B, layout, after, active)


def test_refill_against_the_oracle_and_iteration_zero_stops(torch_cuda):
    """Default visitor, max_iterations 100, 'near' targets on top of a block of problems whose target IS the start pose (the visitor
    fires at iteration 0: the returned q is the unclipped q0, reference ik/ik/dls.cpp:61-63)."""
    torch = torch_cuda
    import ik_amd
    import oracle as O
    B = 8192
    model, problem, data, q0, Q0, T = _problem(torch, "cassie_fixed", "LeftFootFront", ik_amd.KinematicType.Full, "default", B, mode="near")
    # first 1000 problems: target = FK(q0) and q0 pushed OUTSIDE the limits on a joint that is not in the chain (entry 7, the spring)
    q0[:1000, 7] = model.upperPositionLimit[7] + 0.05
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T[:, :, :1000] = ik_amd.task_frames_fk_batch(problem, Q0[:, :1000].contiguous(), data)
    q_gpu, ok, it = _solve(ik_amd, problem, data, Q0, T, "1")
    om = O.OracleModel(model.flat())
    tasks = O.make_tasks([(model.getFrameId("LeftFootFront"), 0, 2, 0, None)])
    tg = T.permute(2, 0, 1).contiguous().cpu().numpy()
    q_ref, ok_ref, it_ref = O.dls_batch(om, tasks, tg, q0, O.params(100, 1e-2, 1.0, 1e-4), os.cpu_count() or 1)
    assert (it[:1000] == 0).all() and ok[:1000].all() and np.array_equal(q_gpu.T[:1000], q0[:1000])   # unclipped, untouched
    same = it == it_ref
    assert same.mean() > 0.999 and np.array_equal(ok[same], ok_ref[same])
    assert np.abs(q_gpu.T[same] - q_ref[same]).max() <= 1e-6


def test_refill_without_the_optional_outputs(torch_cuda):
    """success / iters may be NULL (include/ikgpu.h): the refill launch then keeps its own iteration counts for the pass-through step."""
    torch = torch_cuda
    import ik_amd
    from ik_amd import capi
    B = 70001
    model, problem, data, q0, Q0, T = _problem(torch, "cassie_fixed", "LeftFootFront", ik_amd.KinematicType.Full, "default", B)
    ref = _solve(ik_amd, problem, data, Q0, T, "0")
    Q = torch.full_like(Q0, float("nan"))
    prm = capi.DlsParams(100, 1e-2, 1.0, 1e-4)
    for mode in ("1", "2"):
        Q.fill_(float("nan"))
        with env(IKGPU_REFILL=mode):
            capi.check(capi.lib().ikgpu_dls_solve_batch(data._h, B, Q0.data_ptr(), T.data_ptr(), C.byref(prm), Q.data_ptr(), None, None, capi.SOA,
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert np.array_equal(Q.cpu().numpy(), ref[0]), mode


def _generic_problem(torch, B):
    """Two frame tasks that share joints on the UR5: the cooperative generic kernel (a persistent kernel with a work queue)."""
    import ik_amd
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("ur5"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("a", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Position))
    problem.add_frame_task("b", ik_amd.FrameTask.create(model, "wrist_1_link", ik_amd.KinematicType.Position))
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel.startswith("dls_generic<")
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.UR5_NOMINAL, np.arange(B), 0, "near")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    return problem, data, Q0, T


@pytest.mark.parametrize("which", ["refill", "generic"])
def test_queue_slots_under_a_replayed_graph_and_many_live_launches(torch_cuda, which):
    """A captured launch keeps its own queue slot; more than 64 live launches on a second stream while the graph is replayed must
    not disturb it (the round-2 ring of 64 slots wrapped onto the graph's slot)."""
    torch = torch_cuda
    import ik_amd
    if which == "refill":
        B = 150000
        _, problem, data, _, Q0, T = _problem(torch, "cassie_fixed", "LeftFootFront", ik_amd.KinematicType.Full, "default", B)
        vis, prm = ik_amd.inverse_kinematics_visitor(), ik_amd.dls_parameters(max_iterations=20)
    else:
        B = 6000
        problem, data, Q0, T = _generic_problem(torch, B)
        vis, prm = ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=10)
    with env(IKGPU_REFILL="1"):
        ref = [t.clone() for t in ik_amd.dls_batch(problem, Q0, T, data, vis, prm)]
        torch.cuda.synchronize()
        side, live = torch.cuda.Stream(), torch.cuda.Stream()
        out_g = tuple(torch.zeros_like(t) for t in ref)
        out_l = tuple(torch.zeros_like(t) for t in ref)
        with torch.cuda.stream(side):
            ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out_g)      # warm the stream's slot before capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out_g)
        for round_ in range(3):
            for t in out_g:
                t.zero_()
            torch.cuda.synchronize()
            g.replay()
            with torch.cuda.stream(live):
                for _ in range(70):
                    ik_amd.dls_batch(problem, Q0, T, data, vis, prm, out=out_l)
            torch.cuda.synchronize()
            for a, b, c in zip(ref, out_g, out_l):
                assert torch.equal(a, b) and torch.equal(a, c), (which, round_)
