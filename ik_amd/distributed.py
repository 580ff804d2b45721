"""Multi-GPU sharding of a batch of independent IK problems.

One process per GPU.  Problems are independent, the model + task table are replicated, so the only exchange
step is ONE all-gather of the solved configurations together with the success / iteration flags --
`torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; "gloo" runs the same code on CPU for tests.

A rank's results live in one packed byte buffer  [ q: nq x b float64 | iterations: b int32 | success: b uint8 ]
that the solve kernel writes through typed views, so the exchange is a single collective per step (three small
collectives cost three launch latencies); the gathered layout is [world][nq][b]: each rank's component-major
block stays contiguous and nothing is transposed on either side.  xGMI is a point-to-point mesh (7 links per
GPU): the all-gather of step k is issued asynchronously and overlaps the solve of step k + 1 (two buffer
sets alternate), so a step costs max(compute, exchange) instead of their sum.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of problem indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardBuffers:
    """Packed send buffer of one rank (with typed views for the kernel) and the gathered receive buffer."""

    def __init__(self, nq, b_local, world, device):
        self.nq, self.b, self.world = int(nq), int(b_local), int(world)
        q_bytes, it_bytes = self.nq * self.b * 8, self.b * 4
        self.nbytes = (q_bytes + it_bytes + self.b + 15) // 16 * 16
        self._off = (0, q_bytes, q_bytes + it_bytes)
        self.local = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.all = torch.zeros((self.world, self.nbytes), dtype=torch.uint8, device=device)
        self.Q, self.it, self.ok = self._views(self.local)
        self.work = None

    def _views(self, flat):
        o = self._off
        Q = flat[o[0]:o[1]].view(torch.float64).view(self.nq, self.b)
        it = flat[o[1]:o[2]].view(torch.int32)
        ok = flat[o[2]:o[2] + self.b]
        return Q, it, ok

    def out(self):
        """(Q, success, iterations) views of the send buffer, in the order ik_amd.dls_batch(out=...) takes."""
        return self.Q, self.ok, self.it

    def gathered(self):
        """Per-rank views of the receive buffer: lists of Q [nq, b], success [b], iterations [b]."""
        parts = [self._views(self.all[r]) for r in range(self.world)]
        return [p[0] for p in parts], [p[2] for p in parts], [p[1] for p in parts]

    def all_gather(self, async_op=False, group=None):
        """One collective: every rank ends with every rank's packed block."""
        self.work = dist.all_gather_into_tensor(self.all.view(-1), self.local, group=group, async_op=async_op)
        return self.work

    def wait(self):
        """Make the current stream wait for the last asynchronous all-gather on this buffer set."""
        if self.work is not None:
            self.work.wait()
            self.work = None
