#!/usr/bin/env python3
"""Throughput of the four operating points at 1080p (gray f32, batch 16): python tools/time_presets.py [batch]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
I0, I1 = bench.synth_batch(B, 3, torch.device("cuda", 0))
for opp in (1, 2, 3, 4):
    op = F.operating_point(opp, bench.W, 1)
    ofc = OFClass(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=B)
    out = ofc.new_outflow(B)
    for _ in range(2):
        ofc.calc_batch(I0, I1, None, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10 if opp < 4 else 3
    for _ in range(n):
        ofc.calc_batch(I0, I1, None, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("1080p op-pt %d (scales %d..%d, ps %d, %d iterations, refinement %s) batch %d: %.2f ms/step, %.0f pairs/s"
          % (opp, op.coarsest_scale, op.finest_scale, op.patch_size, op.grad_descent_iter, op.use_var_ref, B, dt * 1e3, B / dt))
    ofc.close()
