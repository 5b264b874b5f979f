#!/usr/bin/env python3
"""fuzz of the asynchronous entry points: random sequences of submit / wait (stream or host) / synchronize / close-with-work-pending
on pipes (fotg_pipe_*) and nodes (fotg_node_*, two slots on one GPU) with random depths, batch sizes and batch fill, every finished
batch compared with the single-context result of its frames.  usage: python tools/fuzz_api.py [rounds] [seed]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd.pipeline import FlowPipeline
from flowonthego_amd.node import FlowNode
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
checked = 0
for r in range(rounds):
    noc = int(rng.choice([1, 1, 3]))
    w, h = int(rng.integers(120, 700)), int(rng.integers(100, 420))
    opp = int(rng.integers(1, 4))
    op = F.operating_point(opp, w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 12)
    if rng.random() < 0.3:
        op.finest_scale = max(0, op.finest_scale - 2)           # taller finest levels: tile solver / level pipeline inside a pipe
    u8 = bool(rng.random() < 0.3)
    mb = int(rng.integers(1, 7))
    ip = F.img_params(width=w, height=h, padding=op.patch_size)
    try:
        ofc = OFClass(op, ip, max_batch=mb)
    except F.FotgError:
        continue
    # a pool of frames and their single-context flows
    pool = []
    for k in range(mb + 2):
        a, b = synth_pair(h, w, seed=int(rng.integers(0, 10 ** 6)), noc=noc)
        pool.append((torch.from_numpy(a.astype(np.uint8) if u8 else a).cuda(), torch.from_numpy(b.astype(np.uint8) if u8 else b).cuda()))
    one = ofc.calc_batch_u8 if u8 else ofc.calc_batch
    refs = [one(a[None], b[None])[0].clone() for a, b in pool]
    torch.cuda.synchronize()
    use_node = rng.random() < 0.3
    if os.environ.get("FUZZ_VERBOSE"):
        print(r, "node" if use_node else "pipe", w, h, noc, opp, mb, u8, op.finest_scale, flush=True)
    if not use_node:
        depth = int(rng.integers(1, 5))
        pipe = FlowPipeline(op, ip, max_batch=mb, depth=depth)
        pending = []                                             # (ticket, out, ids)
        def check(entry):
            global checked
            t, out, ids = entry
            for q, i in enumerate(ids):
                assert torch.equal(out[q], refs[i]), ("pipe", r, t, q)
            checked += len(ids)
        for step in range(int(rng.integers(5, 40))):
            act = rng.random()
            if act < 0.6 or not pending:
                n = int(rng.integers(1, mb + 1))
                ids = [int(x) for x in rng.integers(0, len(pool), n)]
                I0 = torch.stack([pool[i][0] for i in ids]); I1 = torch.stack([pool[i][1] for i in ids])
                out = pipe.new_outflow(n)
                torch.cuda.synchronize()
                t, _ = pipe.submit(I0, I1, None, out, after_current_stream=bool(rng.random() < 0.5))
                pending.append((t, out, ids, I0, I1))
            elif act < 0.8:
                e = pending.pop(int(rng.integers(0, len(pending))))
                host = bool(rng.random() < 0.5)
                pipe.wait(e[0], host=host)
                if not host: torch.cuda.current_stream().synchronize()
                check(e[:3])
            elif act < 0.9:
                pipe.synchronize()
                for e in pending: check(e[:3])
                pending = []
            else:
                pipe.take_stalls()
        if rng.random() < 0.5:
            pipe.synchronize()
            for e in pending: check(e[:3])
        pipe.close()                                             # possibly with work pending: must drain
        del pending
    else:
        depth = int(rng.integers(1, 4))
        node = FlowNode(op, ip, [0, 0], max_batch=mb, depth=depth)
        pending = []
        for step in range(int(rng.integers(3, 20))):
            if rng.random() < 0.65 or not pending:
                n = int(rng.integers(1, 2 * mb + 1))
                ids = [int(x) for x in rng.integers(0, len(pool), n)]
                if rng.random() < 0.5:
                    I0 = torch.stack([pool[i][0] for i in ids]); I1 = torch.stack([pool[i][1] for i in ids])
                    torch.cuda.synchronize()
                    t, out = node.submit_scatter(I0, I1, None, chunk=int(rng.integers(1, mb + 1)))
                    pending.append((t, [out], [ids], I0, I1))
                else:
                    sh = [node.shard(n, d) for d in range(2)]
                    I0 = [torch.stack([pool[i][0] for i in ids[b:e]]) if e > b else None for b, e in sh]
                    I1 = [torch.stack([pool[i][1] for i in ids[b:e]]) if e > b else None for b, e in sh]
                    torch.cuda.synchronize()
                    t, outs = node.submit(n, I0, I1)
                    pending.append((t, outs, [ids[b:e] for b, e in sh], I0, I1))
            else:
                e = pending.pop(int(rng.integers(0, len(pending))))
                node.wait(e[0])
                for out, ids in zip(e[1], e[2]):
                    for q, i in enumerate(ids):
                        assert torch.equal(out[q], refs[i]), ("node", r, e[0], q)
                    checked += len(ids)
        node.synchronize()
        for e in pending:
            for out, ids in zip(e[1], e[2]):
                for q, i in enumerate(ids):
                    assert torch.equal(out[q], refs[i]), ("node", r, e[0], q)
                checked += len(ids)
        node.close()
    ofc.close()
print("fuzz_api: %d rounds, %d flows compared with the single-context result, all equal, no crash" % (rounds, checked))
