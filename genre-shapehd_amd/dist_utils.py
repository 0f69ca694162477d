"""Multi-GPU plumbing for the hot path.

The path shards embarrassingly by batch item: no kernel reads across `n`, so every rank runs the
whole path on its own slice of the batch and there is NO data-path collective.  What is here is
only what a launcher needs: rank/world discovery, contiguous batch sharding, a fence, and the
max-over-ranks reduction used for timing.  One process per GPU; backend "nccl" is RCCL on ROCm
(gloo is used by the CPU tests).
"""
import os

import torch


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend="nccl", device=None):
    """torch.distributed init from the torchrun environment; returns the module or None when
    WORLD_SIZE == 1.  MASTER_ADDR defaults to 127.0.0.1 (single node)."""
    rank, _, world = env_rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    import datetime
    # a rank that dies inside a step leaves the others in a collective it never joins: bounded, not forever
    kw = {"timeout": datetime.timedelta(minutes=int(os.environ.get("GENRE_DIST_TIMEOUT_MIN", "15")))}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_bounds(total, rank, world):
    """contiguous [lo, hi) slice of `total` batch items owned by `rank`; sizes differ by <= 1"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, rank, world):
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def fence(dist, device_sync=None):
    """barrier bracketed by device synchronisation (the timed-region fence of bench.py)"""
    if device_sync is not None:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync is not None:
        device_sync()


def max_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def all_agree(dist, ok, device="cpu"):
    """True iff `ok` holds on EVERY rank (a MIN all-reduce): a rank that failed to build a model or ran out of memory tells the
    others before they enter a collective it would never join"""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.item() > 0.5


def per_rank(dist, value, device="cpu"):
    """[value on rank 0, value on rank 1, ...] on every rank (one all-gather of a float64 scalar): the per-rank figures that
    bench.py prints beside the max-over-ranks one, so that a scaling record can be checked for N ranks at a glance"""
    if dist is None:
        return [float(value)]
    world = dist.get_world_size()
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [p.item() for p in parts]


def average_float_buffers(dist, nets):
    """BatchNorm running statistics stay PER RANK during training (train.py: broadcast_buffers=False, no SyncBN -- the reference
    has neither).  A checkpoint is written by rank 0 only, so without this it would carry rank 0's statistics alone and every
    rank would resume from them: the launcher calls this once before saving / evaluating -- `running_mean` / `running_var` of
    every batch-norm module are replaced by their mean over the ranks.  Nothing else is touched: constant buffers
    that are identical on every rank (the renderer's `grid` / `depth_weight`, camera constants) would come back as
    (x * world) / world, which is not exact for world sizes that are not a power of two, and integer buffers
    (num_batches_tracked) are identical already."""
    if dist is None:
        return
    world = dist.get_world_size()
    for net in nets:
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                for b in (m.running_mean, m.running_var):
                    if b is not None:
                        dist.all_reduce(b, op=dist.ReduceOp.SUM)
                        b.div_(world)


def gather_batch(dist, local, total):
    """assemble variable-size batch shards into [total, ...] on every rank (a convenience for
    callers; the hot path itself never needs it)"""
    if dist is None:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(torch.empty((hi - lo,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device))
    if len({p.shape for p in parts}) == 1:
        dist.all_gather(parts, local.contiguous())
    else:                                   # uneven shards: one broadcast per rank
        for r, buf in enumerate(parts):
            if r == rank:
                buf.copy_(local)
            dist.broadcast(buf, src=r)
    return torch.cat(parts, 0)
