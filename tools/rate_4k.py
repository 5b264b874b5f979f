#!/usr/bin/env python3
"""throughput of 4K operating-point-4 pairs: python tools/rate_4k.py [batch depth] ...   (pairs of B per submit, D submits in flight)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.pipeline import FlowPipeline
f0, f1 = synth_pair(2160, 3840, seed=5)
op = F.operating_point(4, 3840, 1)
ip = F.img_params(width=3840, height=2160, padding=op.patch_size)
args = [int(x) for x in sys.argv[1:]] or [1, 4]
for B, D in zip(args[0::2], args[1::2]):
    a = torch.from_numpy(np.stack([f0] * B)).cuda(); b = torch.from_numpy(np.stack([f1] * B)).cuda()
    pipe = FlowPipeline(op, ip, max_batch=B, depth=D)
    outs = [pipe.new_outflow(B) for _ in range(D)]
    for k in range(2 * D): pipe.submit(a, b, None, outs[k % D], after_current_stream=False)
    pipe.synchronize(); t = time.perf_counter()
    n = 6 * D
    for k in range(n): pipe.submit(a, b, None, outs[k % D], after_current_stream=False)
    pipe.synchronize()
    print("batch %d x %d in flight: %.2f ms per pair" % (B, D, (time.perf_counter() - t) / (n * B) * 1e3))
    pipe.close()
