"""`gloo` tests (world size 2, and 3 with a batch it does not divide) of the multi-GPU path's host logic:
contiguous sharding of the batch, per-rank generation of exactly its own shard, ONE all-gather of each rank's packed
(q rows | iterations | success) block into the [world][slot] layout, with the full and the compact (support rows only) payload.
On CPU the per-rank solver is the oracle (the device kernels need a GPU); the collective code is the same function bench.py
runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, urdf_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, outdir, compact):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as O
    import ik_amd
    from ik_amd import distributed as ikdist, workload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
        om = O.OracleModel(model.flat())
        fid = model.getFrameId("LeftFootFront")
        tasks = O.make_tasks([(fid, 0, 2, 0, None)])
        lo, hi = ikdist.shard_range(total, rank, world)
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                         workload.cassie_nominal(model.names), np.arange(lo, hi), seed=0, mode="near")
        tg = O.fk_batch(om, qs, [fid])
        q, ok, it = O.dls_batch(om, tasks, tg, q0, O.params(20, 1e-2, 1.0, 1e-4))
        rows = SUPPORT_ROWS if compact else None
        bufs = ikdist.ShardBuffers(model.nq, total, rank, world, torch.device("cpu"), rows=rows)
        Q, OK, IT = bufs.out()                     # the views a device solve writes into
        assert tuple(Q.shape) == (model.nq, hi - lo)
        Q.copy_(torch.from_numpy(np.ascontiguousarray(q.T)))
        OK.copy_(torch.from_numpy(ok))
        IT.copy_(torch.from_numpy(it))
        bufs.all_gather(async_op=True)             # the same call bench.py issues over RCCL
        bufs.wait()
        Qs, OKs, ITs = bufs.gathered()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{"Q%d" % r: x.numpy() for r, x in enumerate(Qs)},
                 OK=np.concatenate([x.numpy() for x in OKs]), IT=np.concatenate([x.numpy() for x in ITs]))
    finally:
        dist.destroy_process_group()


# q rows of the LeftFootFront support in cassie_fixed (HipRoll .. TarsusPitch, FootPitch; row 6 is the AchillesSpring leaf)
SUPPORT_ROWS = [0, 1, 2, 3, 4, 5, 7]


def test_shard_range_partitions_the_batch():
    from ik_amd.distributed import shard_range
    for total, world in ((10, 3), (262144, 8), (7, 8), (64, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,total,compact", [(2, 96, False), (3, 10, False), (3, 10, True), (2, 7, True)])
def test_sharded_solve_equals_single_process(native_built, tmp_path, world, total, compact):
    import oracle as O
    import ik_amd
    from ik_amd import workload
    from ik_amd.distributed import expand_rows, shard_range
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path), compact), nprocs=world, join=True)
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("LeftFootFront")
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                     workload.cassie_nominal(model.names), np.arange(total), seed=0, mode="near")
    q, ok, it = O.dls_batch(om, O.make_tasks([(fid, 0, 2, 0, None)]), O.fk_batch(om, qs, [fid]), q0, O.params(20, 1e-2, 1.0, 1e-4))
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for other in ranks[1:]:
        for key in ranks[0].files:
            assert np.array_equal(ranks[0][key], other[key])          # every rank ends with everything
    r0 = ranks[0]
    assert np.array_equal(r0["OK"], ok) and np.array_equal(r0["IT"], it)
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        Qr = r0["Q%d" % r]                                            # [rows, shard of rank r]
        if not compact:
            assert Qr.shape == (model.nq, hi - lo)
            assert np.array_equal(Qr.T, q[lo:hi])
        else:
            assert Qr.shape == (len(SUPPORT_ROWS), hi - lo)
            full = expand_rows(torch.from_numpy(Qr), torch.tensor(SUPPORT_ROWS), torch.from_numpy(np.ascontiguousarray(q0[lo:hi].T)),
                               torch.from_numpy(model.lowerPositionLimit), torch.from_numpy(model.upperPositionLimit),
                               torch.from_numpy(it[lo:hi]))
            assert np.array_equal(full.numpy().T, q[lo:hi])           # a consumer holding q0 rebuilds the whole configuration


def test_support_rows_of_the_leg_problem(native_built):
    """The rows the compact payload ships are what the ABI reports as the problem's support (host-only plan: needs no GPU)."""
    import ik_amd
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    chain = ["LeftHipRoll", "LeftHipYaw", "LeftHipPitch", "LeftKneePitch", "LeftShinPitch", "LeftTarsusPitch", "LeftFootPitch"]
    flat = model.flat()
    rows = sorted(int(flat["idx_q"][model.names.index(n)]) for n in chain)
    assert rows == SUPPORT_ROWS
