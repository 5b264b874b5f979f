"""Frame-pair sharding across ranks: pairs are independent, so a batch is cut into contiguous
ranges, pair k -> rank k*world/n; no collective is on the data path.  RCCL/gloo is only used by callers for the
barrier and to reduce timings."""


def shard_range(n_pairs: int, rank: int, world: int):
    """[begin, end) of the pairs rank `rank` of `world` processes (sizes differ by at most one)"""
    if world < 1 or not (0 <= rank < world) or n_pairs < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_pairs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """wall time of the slowest rank (what bench.py divides the global pair count by)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
