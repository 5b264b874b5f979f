#!/usr/bin/env python3
"""Host issue time against GPU time of a FlowPipeline: python tools/pipe_host.py [batch] [depth] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.pipeline import FlowPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda", 0)
op = F.operating_point(2, bench.W, 1)
pipe = FlowPipeline(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=B, depth=D)
slots = [bench.synth_batch(B, 5 + k, dev) + (pipe.new_outflow(B),) for k in range(D)]
for i in range(2 * D):
    pipe.submit(slots[i % D][0], slots[i % D][1], None, slots[i % D][2], after_current_stream=False)
pipe.synchronize()
t = time.perf_counter()
for i in range(N):
    pipe.submit(slots[i % D][0], slots[i % D][1], None, slots[i % D][2], after_current_stream=False)
th = time.perf_counter() - t
pipe.synchronize()
tt = time.perf_counter() - t
print("batch %d, %d in flight, %d steps: host issue %.3f ms per step, total %.3f ms per step -> %.0f pairs/s" % (B, D, N, th / N * 1e3, tt / N * 1e3, N * B / tt))
