"""Two ranks on ONE GPU through the real engine (shard.py:119): pipelined_scatter_compute with a PipeEngine over a
real FlowPipeline, more chunks than receive buffers (depth + 1), so that every buffer is received into again while earlier
chunks are still in flight -- the gathered flows must equal the single-context result bit for bit.

RCCL refuses two ranks on one device, so the transport is a gloo group whose point-to-point ops carry the device tensors
through pinned host staging (StagedGloo below): a receive's wait() enqueues the host-to-device copy on the CURRENT stream, i.e.
exactly where an RCCL request's wait() orders the current stream behind the transfer.  What is under test is the engine side of
the contract: submits behind the current stream, device-side engine.wait() in front of a buffer's reuse, one output slice per
submit."""
import os
import time
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StagedGloo:
    def __init__(self, dist):
        import torch
        self.d, self.torch, self.keep = dist, torch, []
        self.isend, self.irecv = "isend", "irecv"

    def get_rank(self):
        return self.d.get_rank()

    def get_world_size(self):
        return self.d.get_world_size()

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Send:
        def __init__(self, w):
            self.w = w

        def wait(self):
            self.w.wait()

    class _Recv:
        def __init__(self, w, cpu, dst):
            self.w, self.cpu, self.dst = w, cpu, dst

        def wait(self):
            self.w.wait()
            self.dst.copy_(self.cpu, non_blocking=True)        # on the current stream, like an RCCL request's wait()

    def batch_isend_irecv(self, ops):
        reqs = []
        for o in ops:
            if o.op == "isend":
                cpu = o.tensor.cpu()
                self.keep.append(cpu)
                reqs.append(self._Send(self.d.isend(cpu, o.peer)))
            else:
                cpu = self.torch.empty(tuple(o.tensor.shape), dtype=o.tensor.dtype, pin_memory=True)
                self.keep.append(cpu)
                reqs.append(self._Recv(self.d.irecv(cpu, o.peer), cpu, o.tensor))
        return reqs


def _worker(rank, world, port, n_pairs, chunk, depth, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        import flowonthego_amd as F
        from conftest import synth_pair
        from flowonthego_amd.oflow import OFClass
        from flowonthego_amd.pipeline import FlowPipeline
        from flowonthego_amd.shard import PipeEngine, gather_flows_exact, pipelined_scatter_compute, shard_range
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)                           # both ranks on the one GPU of the box
        torch.cuda.set_device(dev)
        h, w = 272, 480
        op = F.operating_point(2, w, 1)
        ip = F.img_params(width=w, height=h, padding=op.patch_size)
        pipe = FlowPipeline(op, ip, max_batch=chunk, depth=depth, device=0)
        G0 = G1 = None
        if rank == 0:
            pairs = [synth_pair(h, w, seed=100 + k) for k in range(n_pairs)]
            G0 = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
            G1 = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
        sd = StagedGloo(dist)
        ok = True
        for rep in range(3):                                     # (the buffers and the pipe's slots are re-used across calls too)
            flows, (b, e) = pipelined_scatter_compute(G0, G1, n_pairs, (h, w), torch.float32, sd, PipeEngine(pipe), chunk, src=0, device=dev)
            assert (b, e) == shard_range(n_pairs, rank, world)
            assert -(-(e - b) // chunk) > depth + 1            # more chunks than receive buffers: every buffer is re-used
            torch.cuda.synchronize()
            full = gather_flows_exact(flows, n_pairs, sd, dst=0)
            if rank == 0:
                torch.cuda.synchronize()
                ofc = OFClass(op, ip, max_batch=n_pairs, device=0)
                want = ofc.calc_batch(G0, G1)
                torch.cuda.synchronize()
                ok = ok and tuple(full.shape) == tuple(want.shape) and bool(torch.equal(full, want))
                ofc.close()
        dist.barrier()
        pipe.close()
        q.put((rank, bool(ok), ""))
        dist.destroy_process_group()
    except Exception as ex:                                      # (report instead of hanging the parent's q.get)
        import traceback
        q.put((rank, False, traceback.format_exc()))
        raise ex


def test_two_ranks_one_gpu_chunked_scatter_through_the_real_pipeline():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 22, 2, 2, q)) for r in range(2)]      # 11 pairs per rank, chunks of 2 -> 6 steps, 3 buffers
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err or "gathered flows differ from the single-context result")
    assert all(p.exitcode == 0 for p in procs)


# ---- one process, several device slots: fotg_node_* (include/fotg.h), two slots on the one GPU of the box ------------------------
def _node_setup(n, h=272, w=480, seed=300):
    import torch
    import flowonthego_amd as F
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import synth_pair
    from flowonthego_amd.oflow import OFClass
    pairs = [synth_pair(h, w, seed=seed + k) for k in range(n)]
    G0 = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    G1 = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    op = F.operating_point(2, w, 1)
    ip = F.img_params(width=w, height=h, padding=op.patch_size)
    ofc = OFClass(op, ip, max_batch=n)
    want = ofc.calc_batch(G0, G1)
    torch.cuda.synchronize()
    ofc.close()
    return F, op, ip, G0, G1, want


def test_node_two_slots_on_one_gpu_resident_shards():
    """fotg_node_submit with devices = {0, 0}: two shards of one batch through two pipes on explicitly named devices == the
    single-context batch, bit for bit; shards larger than max_batch run in pieces; a batch smaller than the slot count leaves a
    slot empty; 8-bit frames; several jobs in flight and waited for out of order"""
    import torch
    from flowonthego_amd.node import FlowNode, node_shard
    F, op, ip, G0, G1, want = _node_setup(13)
    assert [node_shard(13, 2, d) for d in range(2)] == [(0, 7), (7, 13)] and node_shard(512, 8, 3) == (192, 256)
    from flowonthego_amd.shard import shard_range
    assert all(node_shard(n, w_, d) == shard_range(n, d, w_) for n in (0, 1, 7, 64, 513) for w_ in (1, 2, 3, 8) for d in range(w_))
    node = FlowNode(op, ip, devices=[0, 0], max_batch=3, depth=2)                 # shards of 7 and 6 pairs in pieces of 3, 3, 1 / 3, 3
    sh = [node.shard(13, d) for d in range(2)]
    I0 = [G0[b:e].contiguous() for b, e in sh]; I1 = [G1[b:e].contiguous() for b, e in sh]
    torch.cuda.synchronize()
    t0, o0 = node.submit(13, I0, I1)
    t1, o1 = node.submit(13, I0, I1)
    node.wait(t1); node.wait(t0)
    for o in (o0, o1):
        assert torch.equal(torch.cat(o), want)
    U0 = [t.to(torch.uint8) for t in I0]; U1 = [t.to(torch.uint8) for t in I1]
    torch.cuda.synchronize()
    t2, o2 = node.submit(13, U0, U1)
    node.synchronize()
    assert torch.equal(torch.cat(o2), want)
    t3, o3 = node.submit(1, [G0[:1].contiguous(), None], [G1[:1].contiguous(), None])          # slot 1's shard is empty
    node.wait(t3)
    assert o3[1] is None and torch.equal(o3[0], want[:1])
    with pytest.raises(F.FotgError):
        node.wait(t3 + 1)                                                         # not submitted yet
    tickets = [node.submit(13, I0, I1, o0)[0] for _ in range(16)]                 # the ring of un-waited jobs is 16 deep
    with pytest.raises(F.FotgError):
        node.submit(13, I0, I1, o0)
    node.wait(tickets[-1])
    node.submit(13, I0, I1, o0)
    node.synchronize()
    assert torch.equal(torch.cat(o0), want)
    node.close()


def test_node_scatter_from_the_first_slot():
    """fotg_node_submit_scatter with devices = {0, 0}: slot 1 pulls its shard in chunks through depth + 1 staging buffers
    (peer copies, here on one device) while computing, more chunks than buffers, and pushes its flows back -- the caller's
    output array == the single-context batch; repeated so that buffers and events are re-used across jobs"""
    import torch
    from flowonthego_amd.node import FlowNode
    F, op, ip, G0, G1, want = _node_setup(23, seed=400)
    node = FlowNode(op, ip, devices=[0, 0], max_batch=2, depth=2)                 # slot 1: 11 pairs = 6 chunks of 2 through 3 buffers
    out = torch.full_like(want, float("nan"))
    torch.cuda.synchronize()
    for rep in range(3):
        t, o = node.submit_scatter(G0, G1, out, chunk=2)
        node.wait(t)
        assert o is out and torch.equal(out, want), rep
        out.fill_(float("nan")); torch.cuda.synchronize()
    ta, oa = node.submit_scatter(G0, G1, chunk=1)                                 # two scatter jobs in flight, other chunking
    tb, ob = node.submit_scatter(G0, G1, chunk=2)
    node.wait(tb)
    assert torch.equal(oa, want) and torch.equal(ob, want)
    with pytest.raises(F.FotgError):
        node.submit_scatter(G0, G1, chunk=3)                                      # chunk > max_batch
    # 8-bit frames: the shards travel as bytes (fotg_node_submit_scatter_u8); same flows (the frames hold integer values)
    U0, U1 = G0.to(torch.uint8).contiguous(), G1.to(torch.uint8).contiguous()
    assert torch.equal(U0.float(), G0)
    torch.cuda.synchronize()
    tu, ou = node.submit_scatter(U0, U1, chunk=2)
    node.wait(tu)
    assert torch.equal(ou, want)
    node.close()
    # ... and three-channel 8-bit frames of a gray context (u8_color): three bytes per pixel on the wire
    from flowonthego_amd.oflow import OFClass
    opc = F.operating_point(2, ip.width, 1)
    opc.u8_color = 1
    C0 = torch.stack([U0, (U0 // 2), (255 - U0)], -1).contiguous()
    C1 = torch.stack([U1, (U1 // 2), (255 - U1)], -1).contiguous()
    ofc = OFClass(opc, ip, max_batch=2)
    wantc = torch.cat([ofc.calc_batch_u8(C0[k:k + 2], C1[k:k + 2]).clone() for k in range(0, C0.shape[0], 2)])
    ofc.close()
    nodec = FlowNode(opc, ip, devices=[0, 0], max_batch=2, depth=2)
    torch.cuda.synchronize()
    tc, oc = nodec.submit_scatter(C0, C1, chunk=2)
    nodec.wait(tc)
    assert torch.equal(oc, wantc)
    nodec.close()


def test_cpp_multi_gpu_example(tmp_path):
    """examples/multi_gpu.cpp (include/fotg/node.h): resident shards and the chunked scatter over device slots 0,0 -- the flows
    written equal the oracle's for every pair"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import synth_pair
    from test_gpu_parity import oracle_params
    from test_host import _build_example
    import flowonthego_amd as F
    from oracle import oracle as O
    exe = _build_example(tmp_path, "multi_gpu")
    h, w, n = 272, 480, 7
    pairs = [synth_pair(h, w, seed=500 + k) for k in range(n)]
    a, b, out = (str(tmp_path / x) for x in ("f0.raw", "f1.raw", "flows.raw"))
    np.stack([p[0] for p in pairs]).astype(np.float32).tofile(a)
    np.stack([p[1] for p in pairs]).astype(np.float32).tofile(b)
    op = F.operating_point(2, w, 1)
    p = oracle_params(O, op)
    ref = np.stack([O.flow(O.pad_frame(x, p.sc_f), O.pad_frame(y, p.sc_f), p, 0) for x, y in pairs])
    for mode, chunk in (("resident", "4"), ("scatter", "1")):
        r = subprocess.run([exe, a, b, str(w), str(h), str(n), out, "0,0", mode, chunk, "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out, np.float32).reshape(ref.shape), ref), mode
        os.remove(out)


def test_bench_gpus_2_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver would type it, on a one-GPU box: without FOTG_BENCH_ALLOW_SHARED_GPU it
    refuses (1 GPU visible); with it the launcher starts two ranks that share the GPU and talk over gloo, so the whole N > 1 entry
    -- self-launch through torch.distributed.run, WORLD_SIZE == --gpus check, barriers, max-over-ranks reduction, per-rank times,
    rank placement -- runs for real and prints ONE line with n_gpus = 2, marked as a shared-GPU test run"""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FOTG_BENCH_ALLOW_SHARED_GPU")}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "8", "--warmup", "2", "--windows", "3",
            "--no-cpu-baseline", "--no-breakdown"]
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    r = subprocess.run(args, env=dict(env, FOTG_BENCH_ALLOW_SHARED_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1                                          # rank 0 prints, once
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["shared_gpu_test"] is True and "SHARED-GPU TEST RUN" in d["metric"]
    assert len(r.stdout.strip().splitlines()[-1]) < 6000 and json.loads(r.stdout.strip().splitlines()[-1]) == d      # the short line is the LAST stdout line
    assert set(d["ms_per_step_per_rank"]) == {"min", "max"} and 0 < d["ms_per_step_per_rank"]["min"] <= d["ms_per_step_per_rank"]["max"]
    assert d["config"]["global_batch"] == 32 and d["value"] > 0 and d["gpus_distinct"] == 1
    # the detail (per-rank times, placement, the pipe-vs-single-context check) is in the side file and on stderr
    det = json.loads([l for l in r.stderr.splitlines() if l.startswith("bench.py detail: ")][-1][len("bench.py detail: "):])
    assert json.load(open(os.path.join(ROOT, d["detail"]))) == det
    assert len(det["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in det["ms_per_step_per_rank"]) and det["pipeline_matches_single_context"]
    assert [p["rank"] for p in det["rank_placement"]] == [0, 1] and det["value"] == pytest.approx(d["value"], rel=1e-4)


def test_node_heals_or_reports_a_stalled_wait_per_job(monkeypatch):
    """several jobs in flight and a timed-out inter-workgroup wait in one context: fotg_node_wait recomputes the pieces whose frames
    are in place (resident shards) and succeeds; every job keeps its OWN status for repeated / out-of-order waits; pulled pieces of a
    scatter (staging buffers recycled) are reported as FOTG_ERR_STALL for exactly that job"""
    import ctypes as C
    import torch
    from flowonthego_amd.node import FlowNode
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")
    F, op, ip, G0, G1, want = _node_setup(4, seed=600)
    L = F.lib()
    node = FlowNode(op, ip, devices=[0, 0], max_batch=2, depth=2)
    I0 = [G0[:2].contiguous(), G0[2:].contiguous()]; I1 = [G1[:2].contiguous(), G1[2:].contiguous()]
    torch.cuda.synchronize()
    t, o = node.submit(4, I0, I1)
    node.wait(t)
    assert torch.equal(torch.cat(o), want)
    pipe, ctx = C.c_void_p(), C.c_void_p()
    assert L.fotg_node_pipe(node._h, 1, pipe) == 0 and L.fotg_pipe_context(pipe, 0, ctx) == 0
    # three resident jobs in flight, a stall flagged on slot 1 / context 0 while they run
    jobs = [node.submit(4, I0, I1) for _ in range(3)]
    assert L.fotg_ctx_counter(ctx, b"inject_stall") == 0
    for t, o in jobs:
        node.wait(t)                                               # healed: no error
        assert torch.equal(torch.cat(o), want)
    for t, o in jobs:
        node.wait(t)                                               # repeated waits: each job's own (good) status
    assert L.fotg_ctx_counter(ctx, b"stalls") == 1
    # scatter: the pulled pieces of slot 1 cannot be recomputed -- that job, and only that job, reports the stall
    o0 = torch.empty_like(want); o1 = torch.empty_like(want); o2 = torch.empty_like(want)
    torch.cuda.synchronize()
    t0, _ = node.submit_scatter(G0, G1, o0, chunk=1)
    node.wait(t0)
    assert torch.equal(o0, want)
    t1, _ = node.submit_scatter(G0, G1, o1, chunk=1)
    torch.cuda.synchronize()                                       # (job 1 has run; its tickets are unverified)
    assert L.fotg_ctx_counter(ctx, b"inject_stall") == 0
    assert L.fotg_node_wait(node._h, t1) == 5                      # FOTG_ERR_STALL
    t2, _ = node.submit_scatter(G0, G1, o2, chunk=1)
    node.wait(t2)                                                  # the next job is not blamed
    assert torch.equal(o2, want)
    assert L.fotg_node_wait(node._h, t1) == 5 and L.fotg_node_wait(node._h, t0) == 0 and L.fotg_node_wait(node._h, t2) == 0
    assert L.fotg_node_last_hip_error(node._h) == 0
    node.close()
    # a job with MORE pieces per slot than the pipe keeps arguments for (4 * depth = 8): 24 pairs on two slots in pieces of one pair =
    # 12 tickets per pipe.  A stall flagged when the host gets to them makes the four that have dropped out of the ring suspects that
    # can be neither recomputed nor cleared: the job reports FOTG_ERR_STALL (it used to return FOTG_OK with a
    # possibly invalid flow); the next job is clean
    node = FlowNode(op, ip, devices=[0, 0], max_batch=1, depth=2)
    R0 = [G0.repeat(3, 1, 1).contiguous() for _ in range(2)]; R1 = [G1.repeat(3, 1, 1).contiguous() for _ in range(2)]
    torch.cuda.synchronize()
    assert L.fotg_node_pipe(node._h, 1, pipe) == 0 and L.fotg_pipe_context(pipe, 0, ctx) == 0
    t, o = node.submit(24, R0, R1)
    time.sleep(0.5); torch.cuda.synchronize()                      # (issued and run; nothing verified yet)
    assert L.fotg_ctx_counter(ctx, b"inject_stall") == 0
    assert L.fotg_node_wait(node._h, t) == 5 and L.fotg_node_wait(node._h, t) == 5
    t2, o2 = node.submit(24, R0, R1)
    node.wait(t2)
    assert torch.equal(torch.cat(o2), want.repeat(6, 1, 1, 1))
    node.close()
