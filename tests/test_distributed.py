"""world_size-2 CPU (gloo) tests of the N>1 path: contiguous frame-pair sharding covers every pair exactly once, the step
time is the max over ranks, and the optional scatter of frames from rank 0 / gather of flows back (the only collectives the
north star allows) deliver every pair once and in order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowonthego_amd.shard import (chunk_plan, gather_flows, gather_flows_exact, max_over_ranks, pipelined_scatter_compute,
                                   scatter_pairs, shard_range)


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


def test_chunk_plan_covers_every_pair_once():
    """the chunk scheduler of the pipelined scatter: every pair exactly once, in order within a rank, chunks <= chunk,
    BASELINE configs[4] (512 pairs over 8 ranks in chunks of 16) takes 4 steps of 16 pairs per rank"""
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            for chunk in (1, 3, 16, 64):
                steps = chunk_plan(n, world, chunk)
                for r in range(world):
                    got = [k for row in steps if row[r] is not None for k in range(*row[r])]
                    assert got == list(range(*shard_range(n, r, world)))
                    assert all(row[r] is None or 0 < row[r][1] - row[r][0] <= chunk for row in steps)
                    seen_none = False
                    for row in steps:                     # a rank's chunks are contiguous steps from 0
                        assert not (seen_none and row[r] is not None)
                        seen_none |= row[r] is None
    steps = chunk_plan(512, 8, 16)
    assert len(steps) == 4 and steps[1][3] == (192 + 16, 192 + 32)
    with pytest.raises(ValueError):
        chunk_plan(8, 2, 0)


def _pipe_worker(rank, world, port, n_pairs, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I0 = I1 = None
    if rank == 0:
        I0 = torch.arange(n_pairs, dtype=torch.float32).view(-1, 1, 1).expand(n_pairs, 6, 8).contiguous()
        I1 = I0 + 100
    calls, log = [], []

    class FakeEngine:
        """stand-in for PipeEngine(FlowPipeline): a "flow" that identifies its pair, and a log of the order of events"""
        depth = 2

        def new_out(self, n):
            return torch.zeros((n, 3, 4, 2))

        def submit(self, a, b, out):
            calls.append(a.shape[0])
            out.copy_(torch.stack([a[:, :3, :4], b[:, :3, :4]], -1))
            log.append(("submit", len(calls) - 1))
            return len(calls) - 1

        def wait(self, ticket):
            log.append(("wait", ticket))

        def sync(self):
            log.append(("sync",))

    real_batch = dist.batch_isend_irecv

    def logged_batch(ops):                               # every launch of a grouped transfer, in order
        log.append(("launch", sum(1 for e in log if e[0] == "launch")))
        return real_batch(ops)
    dist.batch_isend_irecv = logged_batch
    flow, (lo, hi) = pipelined_scatter_compute(I0, I1, n_pairs, (6, 8), torch.float32, dist, FakeEngine(), chunk, src=0)
    dist.batch_isend_irecv = real_batch
    ok = (lo, hi) == shard_range(n_pairs, rank, world) and sum(calls) == hi - lo and max(calls, default=0) <= chunk
    ok = ok and all(float(flow[k, 0, 0, 0]) == lo + k and float(flow[k, 0, 0, 1]) == 100 + lo + k for k in range(hi - lo))
    # the order / overlap contract: exactly one host sync, at the end; chunk t+1 is launched BEFORE the host waits for anything
    # (it travels while chunks <= t are in the engine); on a receiving rank the transfer into a buffer that submit k read
    # (transfer k + depth + 1) is launched only after engine.wait(k) was enqueued
    ok = ok and log.count(("sync",)) == 1 and log[-1] == ("sync",)
    if rank != 0:
        pos = {e: i for i, e in enumerate(log)}
        nbuf = FakeEngine.depth + 1
        for t in range(len(calls)):
            if ("launch", t + 1) in pos:
                ok = ok and pos[("submit", t)] < pos[("launch", t + 1)]
            if t + nbuf < len(calls):
                ok = ok and ("wait", t) in pos and pos[("wait", t)] < pos[("launch", t + nbuf)]
    full = gather_flows_exact(flow, n_pairs, dist, dst=0)
    if rank == 0:
        ok = ok and full.shape == (n_pairs, 3, 4, 2) and all(float(full[k, 0, 0, 0]) == k and float(full[k, 0, 0, 1]) == 100 + k for k in range(n_pairs))
    else:
        ok = ok and full is None
    dist.barrier()
    q.put((rank, bool(ok), calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,chunk", [(8, 2), (7, 3), (9, 16), (23, 2)])
def test_two_rank_pipelined_scatter_gloo(n_pairs, chunk):
    """chunked scatter under an asynchronous engine (the FlowPipeline of the N = 1 path; here a fake that logs) + unpadded
    gather: every pair computed once, in order, in chunks of <= chunk, no host wait inside the loop, a receive buffer is
    re-used only behind a device-side wait for the submit that read it, the flows arrive on rank 0 in pair order"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, n_pairs, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
