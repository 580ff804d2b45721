"""World-size-2 `gloo` test of the multi-GPU path's host logic: contiguous sharding of the
batch, per-rank generation of exactly its own shard, ONE all-gather of each rank's packed
(q | iterations | success) block into the [world][nq][B/world] layout.  On CPU the per-rank solver is the oracle (the device kernels need a GPU);
the collective code is the same function bench.py runs over RCCL."""
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


def _worker(rank, world, port, total, outdir):
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
        bufs = ikdist.ShardBuffers(model.nq, hi - lo, world, torch.device("cpu"))
        Q, OK, IT = bufs.out()                     # the views a device solve writes into
        Q.copy_(torch.from_numpy(np.ascontiguousarray(q.T)))
        OK.copy_(torch.from_numpy(ok))
        IT.copy_(torch.from_numpy(it))
        bufs.all_gather(async_op=True)             # the same call bench.py issues over RCCL
        bufs.wait()
        Qs, OKs, ITs = bufs.gathered()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), Q=np.stack([x.numpy() for x in Qs]),
                 OK=np.stack([x.numpy() for x in OKs]), IT=np.stack([x.numpy() for x in ITs]))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_the_batch():
    from ik_amd.distributed import shard_range
    for total, world in ((10, 3), (262144, 8), (7, 8), (64, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_solve_equals_single_process(native_built, tmp_path):
    import oracle as O
    import ik_amd
    from ik_amd import workload
    world, total = 2, 96
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("LeftFootFront")
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                     workload.cassie_nominal(model.names), np.arange(total), seed=0, mode="near")
    q, ok, it = O.dls_batch(om, O.make_tasks([(fid, 0, 2, 0, None)]), O.fk_batch(om, qs, [fid]), q0, O.params(20, 1e-2, 1.0, 1e-4))
    r0, r1 = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world))
    for key in ("Q", "OK", "IT"):
        assert np.array_equal(r0[key], r1[key])                       # every rank ends with everything
    Q = r0["Q"]                                                        # [world, nq, B/world]
    assert Q.shape == (world, model.nq, total // world)
    assert np.array_equal(np.concatenate([Q[r].T for r in range(world)]), q)
    assert np.array_equal(r0["OK"].reshape(-1), ok) and np.array_equal(r0["IT"].reshape(-1), it)
