#!/usr/bin/env python3
"""one batch at a time: the 23 launches of fotg_calc_batch issued eagerly vs replayed from a captured hipGraph
(torch.cuda.CUDAGraph around OFClass.calc_batch): does the graph shorten the gaps between dependent launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
op = F.operating_point(2, bench.W, 1)
ofc = OFClass(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=B)
I0, I1 = bench.synth_batch(B, 1, dev)
out = ofc.new_outflow(B)
def eager(n):
    for _ in range(n): ofc.calc_batch(I0, I1, None, out)
eager(5); torch.cuda.synchronize()
t = time.perf_counter(); eager(200); torch.cuda.synchronize(); te = (time.perf_counter() - t) / 200 * 1e3
ref = out.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eager(3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize()
out.zero_()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 200 * 1e3
print("batch %d: eager %.4f ms per step (%.0f pairs/s), graph replay %.4f ms (%.0f pairs/s), same bits: %s" % (B, te, B / te * 1e3, tg, B / tg * 1e3, torch.equal(out, ref)))
