"""Multi-GPU sharding of a batch of independent IK problems.

One process per GPU.  Problems are independent, the model + task table are replicated, so the only exchange
step is ONE all-gather of the solved configurations together with the success / iteration flags --
`torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; "gloo" runs the same code on CPU for tests.

A rank's results travel in one packed byte buffer  [ q rows: R x b float64 | iterations: b int32 | success: b uint8 ]
so the exchange is a single collective per step (three small collectives cost three launch latencies); the gathered
layout is [world][slot]: each rank's component-major block stays contiguous and nothing is transposed on either side.

* Shards may differ by one problem (`shard_range`, B % world != 0): every rank's slot has the size of the LARGEST
  shard (`all_gather_into_tensor` needs equal contributions), a rank with the smaller shard leaves the tail of its
  slot unused, and the receiver decodes slot r with rank r's own shard size.
* Payload "full": R = nq rows, the solve kernel writes straight into the send buffer through typed views.
  Payload "compact": only the rows of q a solve can move (`ikgpu_problem_support`: 7 of the 16 for a Cassie leg --
  the other entries are q0 clipped to the limits, which a consumer holding q0 rebuilds with `expand_rows`) + the
  flags: 61 instead of 133 bytes per problem on the wire.  The kernel then writes a full [nq, b] result and one
  row-gather packs the support rows into the send buffer.
* xGMI is a point-to-point mesh (7 links per GPU): the all-gather of step k is issued asynchronously and overlaps
  the solve of step k + 1 (two buffer sets alternate), so a step costs max(compute, exchange) instead of their sum.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import capi


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of problem indices owned by `rank` (sizes differ by at most one): ikgpu_shard_range, the rule the
    C ABI's one-process form (ikgpu_dls_solve_batch_sharded) splits by."""
    lo, hi = C.c_int64(), C.c_int64()
    capi.lib().ikgpu_shard_range(int(total), int(rank), int(world), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_size(total, rank, world):
    lo, hi = shard_range(total, rank, world)
    return hi - lo


def _layout(rows, b):
    """Byte offsets of (q rows, iterations, success) in a slot holding b problems, and the bytes used (ikgpu_shard_slot_layout)."""
    oq, oi, os_ = C.c_size_t(), C.c_size_t(), C.c_size_t()
    used = capi.lib().ikgpu_shard_slot_layout(int(rows), int(b), C.byref(oq), C.byref(oi), C.byref(os_))
    return (oq.value, oi.value, os_.value), used


class ShardBuffers:
    """Packed send buffer of one rank (with typed views for the kernel) and the gathered receive buffer.

    nq: rows of a configuration; total: problems in the whole job; rank / world: this process;
    rows: None for the full payload, or the indices of the q rows to ship (compact payload)."""

    def __init__(self, nq, total, rank, world, device, rows=None):
        self.nq, self.total, self.rank, self.world = int(nq), int(total), int(rank), int(world)
        self.b = shard_size(self.total, self.rank, self.world)
        self.b_max = shard_size(self.total, 0, self.world)          # rank 0 always holds a largest shard
        self.rows = None if rows is None else torch.as_tensor(rows, dtype=torch.int64, device=device)
        self.R = self.nq if rows is None else int(self.rows.numel())
        self.nbytes = capi.lib().ikgpu_shard_slot_bytes(self.R, self.total, self.world)   # slot size: the largest shard, 16-byte aligned
        self.local = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.all = torch.zeros((self.world, self.nbytes), dtype=torch.uint8, device=device)
        self.Qs, self.it, self.ok = self._views(self.local, self.b)
        # compact payload: the kernel needs a full [nq, b] result to write to; the support rows are packed from it
        self.Q = self.Qs if rows is None else torch.zeros((self.nq, self.b), dtype=torch.float64, device=device)
        self.work = None

    def _views(self, flat, b):
        o, _ = _layout(self.R, b)
        Q = flat[o[0]:o[1]].view(torch.float64).view(self.R, b)
        it = flat[o[1]:o[2]].view(torch.int32)
        ok = flat[o[2]:o[2] + b]
        return Q, it, ok

    def out(self):
        """(Q, success, iterations) views the solve writes to, in the order ik_amd.dls_batch(out=...) takes."""
        return self.Q, self.ok, self.it

    def pack(self):
        """Compact payload: gather the support rows of the result into the send buffer (a no-op for the full payload)."""
        if self.rows is not None:
            torch.index_select(self.Q, 0, self.rows, out=self.Qs)

    def gathered(self):
        """Per-rank views of the receive buffer: lists of Q rows [R, b_r], success [b_r], iterations [b_r]."""
        parts = [self._views(self.all[r], shard_size(self.total, r, self.world)) for r in range(self.world)]
        return [p[0] for p in parts], [p[2] for p in parts], [p[1] for p in parts]

    def all_gather(self, async_op=False, group=None):
        """One collective: every rank ends with every rank's packed block."""
        self.pack()
        self.work = dist.all_gather_into_tensor(self.all.view(-1), self.local, group=group, async_op=async_op)
        return self.work

    def wait(self):
        """Make the current stream wait for the last asynchronous all-gather on this buffer set."""
        if self.work is not None:
            self.work.wait()
            self.work = None


def expand_rows(Q_rows, rows, q0, lower, upper, iterations):
    """Rebuild full configurations from a compact payload: the shipped rows are the solver's result, every other entry is
    what the loop does to it -- q0 clipped to the limits once a step was taken (reference ik/ik/dls.cpp:67-71,
    ik/ik/common.hpp:53-56), q0 itself when the solve stopped at iteration 0 (dls.cpp:61-63 returns the unclipped q).
    Q_rows [R, b], rows [R], q0 [nq, b], lower / upper [nq], iterations [b] -> [nq, b]."""
    clipped = torch.minimum(torch.maximum(q0, lower[:, None]), upper[:, None])
    Q = torch.where((iterations > 0)[None, :], clipped, q0)
    Q[rows] = Q_rows
    return Q


class ShardGroup:
    """ikgpu_shard_group: ONE process driving several GPUs through the C ABI (ikgpu_dls_solve_batch_sharded) -- per-device problem
    handles and streams, one RCCL communicator per device, one ncclAllGather of the packed slots per step.  The C++ caller's form
    of what ShardBuffers + torch.distributed do with one process per GPU; same shard rule, same slot layout."""

    def __init__(self, problem, devices):
        from . import api
        arr = api._task_table(problem)
        cons, ncons = api._constraint_table(problem)
        self.devices = [int(d) for d in devices]
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        capi.check(capi.lib().ikgpu_shard_group_create(problem.model()._h, arr, len(arr), cons, ncons, dev, len(self.devices), C.byref(self._h)))
        self.nq = problem.model().nq
        self.uses_rccl = bool(capi.lib().ikgpu_shard_group_uses_rccl(self._h))

    def close(self):
        if self._h:
            capi.lib().ikgpu_shard_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot_bytes(self, total):
        return capi.lib().ikgpu_shard_slot_bytes(self.nq, int(total), len(self.devices))

    def solve(self, total, Q0, targets, visitor, p):
        """Q0[r] [nq, b_r], targets[r] [ntasks, 12, b_r]: float64 CUDA tensors on device r (complete: synchronise their producers
        first).  Returns per-device uint8 tensors [ndev, slot_bytes] holding every rank's slot; synchronises the group."""
        from . import api
        n = len(self.devices)
        prm = api._params(visitor, p)
        # (torch.empty, not zeros: a fill kernel would run on torch's current stream, which the group's own non-blocking streams do not
        # wait for -- it could land AFTER the all-gather and wipe gathered slots.  Slot padding is never read by decode().)
        out = [torch.empty((n, self.slot_bytes(total)), dtype=torch.uint8, device="cuda:%d" % d) for d in self.devices]
        for d in set(self.devices):
            torch.cuda.synchronize(d)    # the inputs' producers (and anything else touching these buffers) have completed
        q = (C.c_void_p * n)(*[t.data_ptr() for t in Q0])
        tg = (C.c_void_p * n)(*[t.data_ptr() for t in targets])
        g = (C.c_void_p * n)(*[t.data_ptr() for t in out])
        capi.check(capi.lib().ikgpu_dls_solve_batch_sharded(self._h, int(total), q, tg, C.byref(prm), g))
        capi.check(capi.lib().ikgpu_shard_group_synchronize(self._h))
        return out

    def issue_us(self):
        """Per rank: microseconds its worker thread spent enqueueing the last solve (the ranks issue in parallel)."""
        f = capi.lib().ikgpu_shard_group_last_issue_us
        f.restype = C.c_double
        return [float(f(self._h, r)) for r in range(len(self.devices))]

    def decode(self, gathered, total):
        """gathered [ndev, slot_bytes] (one device's copy) -> lists of (Q [nq, b_r], success [b_r], iterations [b_r]) per rank."""
        n = len(self.devices)
        res = []
        for r in range(n):
            b = shard_size(total, r, n)
            (oq, oi, os_), _ = _layout(self.nq, b)
            flat = gathered[r]
            res.append((flat[oq:oi].view(torch.float64).view(self.nq, b), flat[os_:os_ + b], flat[oi:os_].view(torch.int32)))
        return res
