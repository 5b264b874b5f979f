#!/usr/bin/env python3
"""Do two batches in flight overlap?  K contexts on K streams, calc_batch calls issued round robin from one host thread;
aggregate pairs/s against one context on one stream: python tools/two_streams.py [batch] [contexts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
op = F.operating_point(2, bench.W, 1)
ctx = []
for k in range(K):
    I0, I1 = bench.synth_batch(B, 3 + k, dev)
    ofc = OFClass(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=B)
    ctx.append((ofc, I0, I1, ofc.new_outflow(B), torch.cuda.Stream()))
def run(n, kk):
    for _ in range(n):
        for ofc, I0, I1, out, st in ctx[:kk]:
            with torch.cuda.stream(st):
                ofc.calc_batch(I0, I1, None, out)
for kk in (1, K):
    run(3, kk); torch.cuda.synchronize()
    t = time.perf_counter(); n = 30
    run(n, kk); th = time.perf_counter() - t
    torch.cuda.synchronize(); tt = time.perf_counter() - t
    print("%d context(s) x batch %d: %.0f pairs/s (host issue time %.2f ms per call, GPU %.3f ms per call)" % (kk, B, n * kk * B / tt, th / (n * kk) * 1e3, tt / (n * kk) * 1e3))
