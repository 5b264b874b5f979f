"""a pipe created AFTER n idle pipes: does it still get a hardware queue per slot?  GPU_MAX_HW_QUEUES=8/16 python tools/two_pipes.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.pipeline import FlowPipeline
dev = torch.device("cuda", 0)
op = F.operating_point(2, 1920, 1)
ip = F.img_params(width=1920, height=1080, padding=op.patch_size)
ndummy = int(sys.argv[1])
dummies = [FlowPipeline(op, ip, max_batch=1, depth=4) for _ in range(ndummy)]
pipe = FlowPipeline(op, ip, max_batch=64, depth=4)
slots = [bench.synth_batch(64, 1234 + 97 * k, dev) + (pipe.new_outflow(64),) for k in range(4)]
torch.cuda.synchronize()
def run(n):
    for i in range(n):
        f0, f1, o = slots[i % 4]
        pipe.submit(f0, f1, None, o, after_current_stream=False)
    pipe.synchronize()
run(8)
els = []
for _ in range(7):
    t0 = time.perf_counter(); run(100); els.append(time.perf_counter() - t0)
els.sort()
print("GPU_MAX_HW_QUEUES=%s, %d idle pipes created first: %.0f pairs/s" % (os.environ.get("GPU_MAX_HW_QUEUES"), ndummy, 64 * 100 / els[3]))
