#!/usr/bin/env python3
"""leak check: create / use / destroy contexts, pipes and nodes of several configurations a few hundred times; the free device memory
(hipMemGetInfo) afterwards must be what it was before, and the host RSS must not grow.  usage: python tools/leak_check.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd.pipeline import FlowPipeline
from flowonthego_amd.node import FlowNode
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
f0 = (torch.rand((2, 270, 480), device="cuda") * 255).floor(); f1 = torch.roll(f0, 2, 2)
c0 = (torch.rand((2, 270, 480, 3), device="cuda") * 255).floor(); c1 = torch.roll(c0, 2, 2)
def one(k):
    noc = 3 if k % 4 == 3 else 1
    op = F.operating_point(1 + k % 3, 480, noc)
    op.grad_descent_iter = 4
    op.depth_mode = (k % 5 == 4)
    op.use_fbcon = (k % 7 == 6)
    if k % 6 == 5: op.finest_scale = 0
    ip = F.img_params(width=480, height=270, padding=op.patch_size)
    a, b = (c0, c1) if noc == 3 else (f0, f1)
    kind = k % 3
    if kind == 0:
        o = OFClass(op, ip, max_batch=2); o.calc_batch(a, b); torch.cuda.synchronize(); o.close()
    elif kind == 1:
        p = FlowPipeline(op, ip, max_batch=2, depth=1 + k % 4)
        for _ in range(3): p.submit(a, b)
        if k % 2: p.synchronize()
        p.close()
    else:
        n = FlowNode(op, ip, [0, 0], max_batch=2, depth=2)
        t, out = n.submit_scatter(a, b); n.wait(t); n.close()
for k in range(12): one(k)                        # warm-up: runtime pools, code objects
torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info()[0]
import psutil
rss0 = psutil.Process().memory_info().rss
for k in range(rounds): one(k)
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
rss1 = psutil.Process().memory_info().rss
print("leak check: %d create / use / destroy rounds, free device memory %.1f MB -> %.1f MB (difference %.2f MB); host RSS %.1f -> %.1f MB" % (rounds, free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6, rss0 / 1e6, rss1 / 1e6))
sys.exit(0 if free0 - free1 < 64e6 and rss1 - rss0 < 256e6 else 1)
