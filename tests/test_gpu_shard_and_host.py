"""The C ABI's one-process multi-GPU entry (ikgpu_dls_solve_batch_sharded: per-device handles and streams, one RCCL communicator per
device, one ncclAllGather of the packed slots) on the devices this box has (the driver's GPU box has one: the whole path runs,
the collective is RCCL's all-gather over a communicator of one), and the pipelined host-pointer entry (chunks: H2D || solve || D2H)."""
import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(native_built):
    import torch
    import ik_amd
    from ik_amd import workload
    assert torch.cuda.is_available()
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)

    def inputs(B):
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "uniform")
        Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
        T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
        return Q0, T
    return torch, ik_amd, model, problem, data, inputs


@pytest.mark.parametrize("stop", [False, True])
def test_sharded_solve_matches_the_single_device_solve(setup, stop):
    torch, ik_amd, model, problem, data, inputs = setup
    from ik_amd import distributed as D
    ndev = min(torch.cuda.device_count(), 2)
    total = 20001
    Q0, T = inputs(total)
    vis = ik_amd.inverse_kinematics_visitor() if stop else ik_amd.never_stop_visitor()
    prm = ik_amd.dls_parameters(max_iterations=30)
    ref = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
    torch.cuda.synchronize()
    group = D.ShardGroup(problem, list(range(ndev)))
    assert group.uses_rccl, "librccl should be loadable on the GPU box"
    q_parts, t_parts = [], []
    for r in range(ndev):
        lo, hi = D.shard_range(total, r, ndev)
        q_parts.append(Q0[:, lo:hi].contiguous().to("cuda:%d" % r))
        t_parts.append(T[:, :, lo:hi].contiguous().to("cuda:%d" % r))
    torch.cuda.synchronize()
    out = group.solve(total, q_parts, t_parts, vis, prm)
    for d in range(ndev):                                   # every device holds every rank's slot
        parts = group.decode(out[d], total)
        Q = torch.cat([p[0].to("cuda:0") for p in parts], dim=1)
        ok = torch.cat([p[1].to("cuda:0") for p in parts])
        it = torch.cat([p[2].to("cuda:0") for p in parts])
        assert torch.equal(Q, ref[0]) and torch.equal(ok, ref[1]) and torch.equal(it, ref[2])
    group.close()


@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_group_code_path_with_several_ranks_on_one_device(setup, ndev, monkeypatch):
    """The whole group code path with MORE THAN ONE rank on the one-GPU box: IKGPU_SHARD_LOOPBACK=1 lets the ranks share device 0
    and replaces the all-gather by event-ordered device-to-device copies (shard.cpp) -- one worker thread per rank issuing its
    shard in parallel, uneven shards (total not a multiple of ndev), the slot layout and its decode on EVERY rank's gathered buffer,
    twice on one group (buffers reused), with and without the stop rule.  What stays untested here is RCCL over more than one
    device; the one-device communicator runs in the test above."""
    torch, ik_amd, model, problem, data, inputs = setup
    from ik_amd import distributed as D
    monkeypatch.setenv("IKGPU_SHARD_LOOPBACK", "1")
    total = 30000 + ndev + 1
    Q0, T = inputs(total)
    group = D.ShardGroup(problem, [0] * ndev)
    assert not group.uses_rccl
    for stop, iters in ((False, 20), (True, 100), (False, 3)):
        vis = ik_amd.inverse_kinematics_visitor() if stop else ik_amd.never_stop_visitor()
        prm = ik_amd.dls_parameters(max_iterations=iters)
        ref = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
        q_parts = [Q0[:, lo:hi].contiguous() for lo, hi in (D.shard_range(total, r, ndev) for r in range(ndev))]
        t_parts = [T[:, :, lo:hi].contiguous() for lo, hi in (D.shard_range(total, r, ndev) for r in range(ndev))]
        assert sum(q.shape[1] for q in q_parts) == total and len({q.shape[1] for q in q_parts}) == 2    # uneven shards
        out = group.solve(total, q_parts, t_parts, vis, prm)
        for d in range(ndev):
            parts = group.decode(out[d], total)
            Q = torch.cat([p[0] for p in parts], dim=1)
            ok = torch.cat([p[1] for p in parts])
            it = torch.cat([p[2] for p in parts])
            assert torch.equal(Q, ref[0]) and torch.equal(ok, ref[1]) and torch.equal(it, ref[2]), (ndev, d, stop)
        us = group.issue_us()
        assert len(us) == ndev and all(0.0 < x < 5e5 for x in us), us
    print("ndev %d: per-rank issue time of the last solve %s us" % (ndev, ["%.0f" % x for x in us]))
    group.close()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("B", [8192, 20000, 65536 + 77])
def test_pipelined_host_entry_matches_the_device_entry(setup, layout, B):
    """numpy arrays take ikgpu_dls_solve_batch_host: above 1 MiB the chunked pipeline (pageable memory here; pinned in bench.py)."""
    torch, ik_amd, model, problem, data, inputs = setup
    Q0, T = inputs(B)
    vis, prm = ik_amd.inverse_kinematics_visitor(), ik_amd.dls_parameters(max_iterations=25)
    ref = ik_amd.dls_batch(problem, Q0, T, data, vis, prm)
    q_h, t_h = Q0.cpu().numpy(), T.cpu().numpy()
    if layout == "aos":
        q_h, t_h = np.ascontiguousarray(q_h.T), np.ascontiguousarray(t_h.transpose(2, 0, 1))
    Q, ok, it = ik_amd.dls_batch(problem, q_h, t_h, data, vis, prm, layout=layout)
    if layout == "aos":
        Q = Q.T
    assert np.array_equal(Q, ref[0].cpu().numpy()) and np.array_equal(ok, ref[1].cpu().numpy()) and np.array_equal(it, ref[2].cpu().numpy())
    Q2, _, _ = ik_amd.dls_batch(problem, q_h, t_h, data, vis, prm, layout=layout)      # the arena is reused
    assert np.array_equal(Q2.T if layout == "aos" else Q2, ref[0].cpu().numpy())


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("B", [100, 40000])
def test_pose7_targets(setup, layout, B):
    """Targets as (x y z qx qy qz qw): the device-side expansion (ikgpu_targets_from_pose7) and the host entry's IKGPU_TARGETS_POSE7 flag
    (small staged path and the chunked pipeline) give what the 12-double targets give."""
    import ctypes as C
    from scipy.spatial.transform import Rotation
    torch, ik_amd, model, problem, data, inputs = setup
    from ik_amd import capi
    Q0, T = inputs(B)
    t12 = T.cpu().numpy()                                            # [1, 12, B]
    R = t12[0, :9].T.reshape(B, 3, 3)
    quat = Rotation.from_matrix(R).as_quat()                         # (x y z w), unit
    pose7 = np.concatenate([t12[0, 9:].T, quat], axis=1)             # [B, 7]
    p7 = np.ascontiguousarray(pose7.T[None]) if layout == "soa" else np.ascontiguousarray(pose7[:, None, :])
    lay = capi.SOA if layout == "soa" else capi.AOS
    # (1) device expansion
    d7 = torch.from_numpy(p7).cuda()
    out = torch.empty((1, 12, B) if layout == "soa" else (B, 1, 12), dtype=torch.float64, device="cuda")
    capi.check(capi.lib().ikgpu_targets_from_pose7(B, 1, d7.data_ptr(), out.data_ptr(), lay, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    ref12 = T if layout == "soa" else T.permute(2, 0, 1).contiguous()
    assert (out - ref12).abs().max().item() < 1e-14
    # (2) host entry with the flag against the host entry with the expanded targets
    q_h = Q0.cpu().numpy() if layout == "soa" else np.ascontiguousarray(Q0.cpu().numpy().T)
    t_h = out.cpu().numpy()
    prm = capi.DlsParams(30, 1e-2, 1.0, 1e-4)
    res = []
    for tg, flag in ((t_h, 0), (p7, capi.TARGETS_POSE7)):
        Q = np.empty_like(q_h)
        ok, it = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        capi.check(capi.lib().ikgpu_dls_solve_batch_host(data._h, B, q_h.ctypes.data, tg.ctypes.data, C.byref(prm), Q.ctypes.data, ok.ctypes.data, it.ctypes.data, lay | flag))
        res.append((Q, ok, it))
    for a, b in zip(*res):
        assert np.array_equal(a, b)
