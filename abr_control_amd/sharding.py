"""Batch sharding across GPUs (one process per GPU).  Rows are independent, so a shard is
a contiguous row range and NO collective sits on the data path;
`torch.distributed` is only used by bench.py for the barrier / max-over-ranks timing."""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment"""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(B, rank, world):
    """contiguous rows [lo, hi) of rank `rank`: sizes differ by at most one, cover [0, B) exactly"""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(arrays, rank, world):
    """slice every [B, ...] array (None passes through) to this rank's rows"""
    B = next(a.shape[0] for a in arrays if a is not None)
    lo, hi = shard_range(B, rank, world)
    return [None if a is None else a[lo:hi] for a in arrays]
