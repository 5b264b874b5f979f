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


def scatter_pairs(I0, I1, dist, src: int = 0, device=None):
    """The scatter of SURVEY.md 8e: rank `src` holds the whole batch (n, ...) of both frames, every rank gets its contiguous
    shard (shard_range).  One grouped send/recv per peer (torch.distributed.scatter: RCCL on GPUs, gloo on CPU); the shards
    are padded to equal size on the wire and trimmed on arrival.  Non-source ranks pass I0 = I1 = None plus the per-frame
    `shape`/`dtype` through the returned metadata broadcast.  Returns (I0_shard, I1_shard, (begin, end))."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [None]
    if rank == src:
        meta = [(int(I0.shape[0]), tuple(I0.shape[1:]), str(I0.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(meta, src=src)
    n, shape, dtname = meta[0]
    dtype = getattr(torch, dtname)
    dev = device if device is not None else (I0.device if rank == src else "cpu")
    per = -(-n // world)                                  # padded shard size
    out = torch.empty((2, per) + shape, dtype=dtype, device=dev)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            b, e = shard_range(n, r, world)
            c = torch.zeros((2, per) + shape, dtype=dtype, device=dev)
            c[0, : e - b] = I0[b:e]
            c[1, : e - b] = I1[b:e]
            chunks.append(c)
    dist.scatter(out, chunks, src=src)
    b, e = shard_range(n, rank, world)
    return out[0, : e - b], out[1, : e - b], (b, e)


def gather_flows(flow, n_pairs: int, dist, dst: int = 0):
    """The gather of SURVEY.md 8e: every rank's flows (its shard, in pair order) back to rank `dst` as one (n_pairs, ...)
    tensor in the original pair order; other ranks get None."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    per = -(-n_pairs // world)
    pad = torch.zeros((per,) + tuple(flow.shape[1:]), dtype=flow.dtype, device=flow.device)
    pad[: flow.shape[0]] = flow
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, parts, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        b, e = shard_range(n_pairs, r, world)
        out.append(parts[r][: e - b])
    return torch.cat(out)
