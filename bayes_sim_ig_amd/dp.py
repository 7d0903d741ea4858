"""Data-parallel fit over the GPUs of one node.

The reference is single-device; this is the one exchange the path needs:
every rank holds a full replica and its shard of the (theta, trajectory)
pairs, computes the gradient of its B/R minibatch rows scaled by 1/(B_global)
and the flat fp32 gradient buffer is summed with ONE all-reduce per update
(RCCL over xGMI: torch.distributed backend "nccl"), followed by the identical
Adam step on every rank.  Summarizers and the RFF projection are
per-trajectory independent and need no communication.

The orchestration below is backend-agnostic (gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def broadcast(self, flat):
        dist.broadcast(flat, src=dist.get_global_rank(self.group, 0)
                       if self.group is not None else 0, group=self.group)

    def allreduce_sum(self, flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def mean_losses(self, train_loss, test_loss, n_test):
        """Global mean train loss per update (equal local batches) and the
        count-weighted mean of the held-out NLL over all shards."""
        tl = train_loss.clone()
        dist.all_reduce(tl, op=dist.ReduceOp.SUM, group=self.group)
        tl /= self.world
        packed = torch.cat([test_loss * float(n_test),
                            torch.full((1,), float(n_test), dtype=test_loss.dtype,
                                       device=test_loss.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        return tl, packed[:-1] / packed[-1].clamp_min(1.0)


def run_updates(n_updates, eval_set, grad, allreduce, apply, evaluate):
    """mdnn.py:228-242 with the gradient exchange between backward and the
    optimizer step."""
    for it in range(n_updates):
        grad()
        allreduce()
        apply()
        if it in eval_set:
            evaluate()


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n pairs for ``rank`` (first n % world
    ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
