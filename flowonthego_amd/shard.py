"""Frame-pair sharding across ranks: pairs are independent, so a batch is cut into contiguous
ranges -- rank r gets n // world pairs, the first n % world ranks one more (shard_range below == fotg_node_shard of include/fotg.h);
no collective is on the data path.  RCCL/gloo is only used by callers for the
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


# ----------------------------------------------------------------------------------------------------------------------
# Pipelined scatter ("grouped send/recv scatter of input frames from rank 0, pipelined in chunks under compute"):
# rank `src` holds the whole batch; every rank's contiguous shard (shard_range) is cut into chunks of at most `chunk` pairs;
# in step t every peer receives chunk t of its shard (one grouped batch_isend_irecv: RCCL ncclSend/ncclRecv groups on GPUs,
# gloo on CPU) into one of depth + 1 buffers while the chunks before it are in flight in the engine (the FlowPipeline of the
# N = 1 path).  Nothing is padded, nothing is cloned, the host never waits inside the loop, and rank `src` never builds a second
# copy of the batch: it sends views of its tensors and computes on views of its own shard.
# ----------------------------------------------------------------------------------------------------------------------
def chunk_plan(n_pairs: int, world: int, chunk: int):
    """steps[t][rank] = (begin, end) of the pairs rank `rank` receives (src: computes on) in step t, or None when its shard is
    exhausted; every pair appears exactly once, in order within a rank, chunks of <= `chunk` pairs"""
    if chunk < 1:
        raise ValueError("chunk must be >= 1")
    ranges = [shard_range(n_pairs, r, world) for r in range(world)]
    nsteps = max(-(-(e - b) // chunk) for b, e in ranges) if n_pairs else 0
    steps = []
    for t in range(nsteps):
        row = []
        for b, e in ranges:
            lo = b + t * chunk
            row.append((lo, min(e, lo + chunk)) if lo < e else None)
        steps.append(row)
    return steps


class PipeEngine:
    """adapter: a flowonthego_amd.FlowPipeline as the `engine` of pipelined_scatter_compute (submit / wait / sync)"""

    def __init__(self, pipe):
        self.pipe, self.depth = pipe, pipe.depth

    def new_out(self, n):
        return self.pipe.new_outflow(n)

    def submit(self, a, b, out):
        # ordered behind the current stream (where the receive of this chunk was waited for); returns at once
        return self.pipe.submit(a, b, None, out, after_current_stream=True)[0]

    def wait(self, ticket):
        self.pipe.wait(ticket, host=False)          # the current stream waits on the device; the host does not block

    def sync(self):
        self.pipe.synchronize()


def pipelined_scatter_compute(I0, I1, n_pairs, frame_shape, dtype, dist, engine, chunk: int, src: int = 0, device=None):
    """Runs this rank's shard through `engine` chunk by chunk while the NEXT chunks are in flight from rank `src`.
    I0 / I1: the whole batch on rank `src`, None elsewhere.  Returns (flows of this rank in shard order, (begin, end)).

    engine: the SAME asynchronous path bench.py times at N = 1 (PipeEngine over a FlowPipeline):
      engine.depth                   submits that may be in flight
      engine.new_out(n)              output tensor for n pairs
      engine.submit(a, b, out) -> t  enqueue one chunk behind the current stream; returns at once; results land in `out`
      engine.wait(t)                 the CURRENT STREAM (not the host) waits for submit t
      engine.sync()                  host wait for everything submitted
    No host synchronisation and no copy per chunk: a chunk is received into one of depth + 1 buffers (on GPUs a
    request's wait() orders the current stream behind the transfer), submitted behind that wait, and its buffer is only
    received into again after engine.wait() of the submit that read it has been enqueued in front of that receive; every
    submit writes its own slice of one preallocated output."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    steps = chunk_plan(n_pairs, world, chunk)
    b0, e0 = shard_range(n_pairs, rank, world)
    nbuf = int(engine.depth) + 1
    bufs = None
    if rank != src:
        bufs = [torch.empty((2, chunk) + tuple(frame_shape), dtype=dtype, device=device) for _ in range(nbuf)]
    out = engine.new_out(e0 - b0) if e0 > b0 else None
    tickets = []

    def launch(t):
        ops = []
        if rank == src:
            for r in range(world):
                if r == src or steps[t][r] is None:
                    continue
                lo, hi = steps[t][r]
                ops.append(dist.P2POp(dist.isend, I0[lo:hi], r))
                ops.append(dist.P2POp(dist.isend, I1[lo:hi], r))
        elif steps[t][rank] is not None:
            lo, hi = steps[t][rank]
            ops.append(dist.P2POp(dist.irecv, bufs[t % nbuf][0, : hi - lo], src))
            ops.append(dist.P2POp(dist.irecv, bufs[t % nbuf][1, : hi - lo], src))
        return dist.batch_isend_irecv(ops) if ops else []

    pending = launch(0) if steps else []
    for t in range(len(steps)):
        for req in pending:                      # chunk t has arrived (GPU: the current stream is ordered behind the transfer)
            req.wait()
        if steps[t][rank] is not None:
            lo, hi = steps[t][rank]
            if rank == src:
                a, b = I0[lo:hi], I1[lo:hi]
            else:
                a, b = bufs[t % nbuf][0, : hi - lo], bufs[t % nbuf][1, : hi - lo]
            tickets.append(engine.submit(a, b, out[lo - b0:hi - b0]))
        if t + 1 < len(steps):
            # the buffer chunk t+1 lands in was read by submit t+1-nbuf: order the receive behind that submit
            if rank != src and t + 1 >= nbuf and t + 1 - nbuf < len(tickets):
                engine.wait(tickets[t + 1 - nbuf])
            pending = launch(t + 1)              # chunk t+1 travels while chunks <= t are computed
        else:
            pending = []
    engine.sync()
    return out, (b0, e0)


def gather_flows_exact(flow, n_pairs: int, dist, dst: int = 0):
    """gather of the flows without padding: every rank sends exactly its shard (grouped send/recv), rank `dst` returns
    (n_pairs, ...) in pair order"""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        if flow is not None and flow.shape[0] > 0:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, flow.contiguous(), dst)]):
                req.wait()
        return None
    tail = tuple(flow.shape[1:])
    out = torch.empty((n_pairs,) + tail, dtype=flow.dtype, device=flow.device)
    ops = []
    for r in range(world):
        b, e = shard_range(n_pairs, r, world)
        if r == dst:
            out[b:e] = flow
        elif e > b:
            ops.append(dist.P2POp(dist.irecv, out[b:e], r))
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    return out
