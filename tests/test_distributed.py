"""world_size-2 CPU (gloo) tests of the N>1 path: contiguous frame-pair sharding covers every pair exactly once, the step
time is the max over ranks, and the optional scatter of frames from rank 0 / gather of flows back (the only collectives the
north star allows) deliver every pair once and in order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowonthego_amd.shard import gather_flows, max_over_ranks, scatter_pairs, shard_range


def test_shard_ranges():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256)          # BASELINE configs[4]: 64 pairs per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(n_pairs, rank, world)
    owned = torch.zeros(n_pairs, dtype=torch.int64)
    owned[b:e] = 1
    dist.all_reduce(owned)                               # test-only: count owners per pair
    t = max_over_ranks(0.5 + rank, dist)
    dist.barrier()
    q.put((rank, owned.tolist(), t))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, owned, t in res:
        assert owned == [1] * 9                           # every pair owned by exactly one rank
        assert t == 1.5                                   # max over ranks


def _sg_worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I0 = I1 = None
    if rank == 0:                                        # frame k is filled with k (I0) and 100 + k (I1)
        I0 = torch.arange(n_pairs, dtype=torch.float32).view(-1, 1, 1).expand(n_pairs, 6, 8).contiguous()
        I1 = I0 + 100
    a, b, (lo, hi) = scatter_pairs(I0, I1, dist, src=0)
    assert (lo, hi) == shard_range(n_pairs, rank, world) and a.shape == (hi - lo, 6, 8)
    assert all(float(a[k, 0, 0]) == lo + k and float(b[k, 0, 0]) == 100 + lo + k for k in range(hi - lo))
    flow = torch.stack([a[:, :3, :4], b[:, :3, :4]], -1)  # a stand-in "flow" (n, 3, 4, 2) that identifies its pair
    full = gather_flows(flow, n_pairs, dist, dst=0)
    ok = True
    if rank == 0:
        ok = full.shape == (n_pairs, 3, 4, 2) and all(float(full[k, 0, 0, 0]) == k and float(full[k, 0, 0, 1]) == 100 + k for k in range(n_pairs))
    else:
        ok = full is None
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 7])
def test_two_rank_scatter_gather_gloo(n_pairs):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sg_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
