#!/usr/bin/env python3
"""Determinism soak (not part of the test suite): the same batch many times, every result bit-identical to the first one
(a race in the barrier-stepped / streaming solver protocols would show up as an occasional mismatch).
usage: python tools/soak.py [iterations] [batch] [depth|flow] [op-point]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
op = F.operating_point(int(sys.argv[4]) if len(sys.argv) > 4 else 2, bench.W, 1)
op.depth_mode = len(sys.argv) > 3 and sys.argv[3] == "depth"
ofc = OFClass(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=batch)
I0, I1 = bench.synth_batch(batch, 99, dev)
ref = ofc.calc_batch(I0, I1).clone()
assert torch.isfinite(ref).all()
bad = 0
for k in range(it):
    out = ofc.calc_batch(I0, I1)
    if not torch.equal(out, ref):
        bad += 1
        print("iteration", k, "differs: max abs", float((out - ref).abs().max()), flush=True)
print("iterations", it, "batch", batch, "mismatches", bad)
sys.exit(1 if bad else 0)
