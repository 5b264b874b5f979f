#!/usr/bin/env python3
"""pairs/s of the headline workload with D batches in flight, no result checks (A/B timing of experimental builds):
python tools/inflight_rate.py [depth] [batch] [steps] [windows] [channels]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.pipeline import FlowPipeline
D, B, K, NW, NOC = (int(x) for x in (sys.argv[1:6] + ["4", "64", "100", "9", "1"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
op = F.operating_point(2, 1920, NOC)
ip = F.img_params(width=1920, height=1080, padding=op.patch_size)
pipe = FlowPipeline(op, ip, max_batch=B, depth=D)
def frames(k):
    f0, f1 = bench.synth_batch(B, 1234 + 97 * k, dev)
    if NOC == 3:
        f0 = torch.stack([f0, f0.roll(3, 2), f0.roll(5, 1)], -1).contiguous(); f1 = torch.stack([f1, f1.roll(3, 2), f1.roll(5, 1)], -1).contiguous()
    return f0, f1
slots = [frames(k) + (pipe.new_outflow(B),) for k in range(D)]
torch.cuda.synchronize()
def run(n):
    for i in range(n):
        f0, f1, o = slots[i % D]
        pipe.submit(f0, f1, None, o, after_current_stream=False)
    pipe.synchronize()
run(2 * D)
els = []
for _ in range(NW):
    t0 = time.perf_counter(); run(K); els.append(time.perf_counter() - t0)
els.sort()
print("channels %d depth %d batch %d: %.0f pairs/s (median of %d windows of %d steps; %.4f ms per step)" % (NOC, D, B, B * K / els[len(els) // 2], NW, K, els[len(els) // 2] / K * 1e3))
