"""Multi-GPU sharding of a batch of independent IK problems.

One process per GPU.  Problems are independent, the model + task table are replicated, so the
only exchange step is ONE all-gather of the converged configurations (plus the success / iteration
flags) -- `torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; "gloo" runs the same code
on CPU for tests.  The gathered layout is [world][nq][B/world]: each rank's SoA block stays
contiguous, so no transpose is needed on either side of the collective.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of problem indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GatherBuffers:
    """Pre-allocated receive buffers, so the timed region holds no allocation."""

    def __init__(self, nq, b_local, world, device):
        self.Q = torch.empty((world, nq, b_local), dtype=torch.float64, device=device)
        self.ok = torch.empty((world, b_local), dtype=torch.uint8, device=device)
        self.it = torch.empty((world, b_local), dtype=torch.int32, device=device)


def all_gather_solutions(Q_local, ok_local, it_local, bufs, group=None):
    """Every rank ends with all configurations: bufs.Q [world, nq, b], bufs.ok / bufs.it [world, b].
    Shards must have equal size (pad the batch to a multiple of the world size)."""
    dist.all_gather_into_tensor(bufs.Q.view(-1, bufs.Q.shape[-1]), Q_local, group=group)
    dist.all_gather_into_tensor(bufs.ok.view(-1), ok_local, group=group)
    dist.all_gather_into_tensor(bufs.it.view(-1), it_local, group=group)
    return bufs.Q, bufs.ok, bufs.it
