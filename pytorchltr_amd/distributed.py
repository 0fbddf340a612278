"""Data-parallel use of the hot path: one process per GPU, queries sharded across ranks.

Every loss/metric on the path is a per-query function of (scores[b,:], relevance[b,:], n[b])
(reference: loss/pairwise_additive.py:45,84; evaluation/dcg.py:94-98) -- there is no
cross-query term, so the data path needs NO collective.  The only exchange in a training step
is the standard gradient all-reduce of the scorer's parameters plus (loss-sum, count) for
logging: F+3 floats, latency-bound.  It is issued as ONE flattened bucket so that it costs a
single RCCL launch (xGMI ring/tree choice is irrelevant below a few KB).

Backend-agnostic: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world_size):
    """Contiguous, balanced [lo, hi) chunk of `total` queries for `rank`."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(total), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank=None, world_size=None):
    """Slices every tensor of a padded batch (features, scores, relevance, n, ...) along dim 0
    to this rank's contiguous chunk of queries."""
    if rank is None:
        rank = dist.get_rank()
    if world_size is None:
        world_size = dist.get_world_size()
    total = tensors[0].shape[0]
    for t in tensors:
        if t.shape[0] != total:
            raise ValueError("all tensors must share the batch dimension")
    lo, hi = shard_bounds(total, rank, world_size)
    return tuple(t[lo:hi] for t in tensors)


def allreduce_step(grads, loss_sum, count, group=None):
    """Sums parameter gradients and (loss_sum, count) over ranks in one flattened all-reduce.

    `grads`: list of gradient tensors whose local values are d(sum of this rank's per-query
    losses)/d(param).  After the call each holds d(global mean loss)/d(param) -- i.e. what a
    single process computes for `loss_fn(...).mean().backward()` on the unsharded batch.
    Returns (global_mean_loss, global_count) as python floats' tensors on the grads' device.
    """
    dev = grads[0].device if grads else loss_sum.device
    sizes = [g.numel() for g in grads]
    flat = torch.empty(sum(sizes) + 2, dtype=torch.float32, device=dev)
    off = 0
    for g, sz in zip(grads, sizes):
        flat[off:off + sz] = g.reshape(-1).float()
        off += sz
    flat[off] = loss_sum.reshape(()).float() if torch.is_tensor(loss_sum) else float(loss_sum)
    flat[off + 1] = float(count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    total = flat[off + 1].clamp(min=1.0)
    off = 0
    for g, sz in zip(grads, sizes):
        g.copy_((flat[off:off + sz] / total).reshape(g.shape).to(g.dtype))
        off += sz
    return flat[off] / total, flat[off + 1]


def allreduce_metric(metric_values, group=None):
    """Global mean of a per-query metric (e.g. ndcg@10) over all ranks' queries."""
    v = metric_values.reshape(-1).float()
    acc = torch.stack([v.sum(), torch.tensor(float(v.numel()), device=v.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc[0] / acc[1].clamp(min=1.0)
