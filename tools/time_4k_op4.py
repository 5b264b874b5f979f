#!/usr/bin/env python3
"""BASELINE configs[3]: one 3840x2160 pair at operating point 4 (quality preset): python tools/time_4k_op4.py [--in-flight]
(--in-flight: also the throughput of consecutive pairs through a FlowPipeline, 2 and 4 pairs in flight)"""
import sys, time
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")       # (a hardware queue per pipeline slot, see bench.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
f0, f1 = synth_pair(2160, 3840, seed=5)
op = F.operating_point(4, 3840, 1)
op.fast_math = "--fast" in sys.argv              # the tolerance mode of the patch loop (csrc/lk_fast.hip.h)
ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
for _ in range(2):
    ofc.calc(a, b)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    ofc.calc(a, b)
torch.cuda.synchronize()
print("4K op-pt 4 (scales %d..%d): %.2f ms per pair" % (op.coarsest_scale, op.finest_scale, (time.perf_counter() - t) / 5 * 1e3))
# consecutive pairs of a 4K video through a FlowPipeline (one pair per submit, several in flight)
from flowonthego_amd.pipeline import FlowPipeline
for depth in ((2, 4, 6, 8) if "--in-flight" in sys.argv else ()):
    pipe = FlowPipeline(op, F.img_params(width=3840, height=2160, padding=op.patch_size), max_batch=1, depth=depth)
    outs = [pipe.new_outflow(1) for _ in range(depth)]
    for k in range(2 * depth):
        pipe.submit(a[None], b[None], None, outs[k % depth], after_current_stream=False)
    pipe.synchronize(); t = time.perf_counter()
    n = 6 * depth
    for k in range(n):
        pipe.submit(a[None], b[None], None, outs[k % depth], after_current_stream=False)
    pipe.synchronize()
    print("4K op-pt 4, %d pairs in flight: %.2f ms per pair (throughput), same bits: %s" % (depth, (time.perf_counter() - t) / n * 1e3, bool(torch.equal(outs[0][0], ofc.calc(a, b)))))
    pipe.close()
