#!/usr/bin/env python3
"""Does a batch's result depend on what the context computed before?  X, Y, X again through one context: python tools/history_check.py [batch] [w] [h] [op]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
op = F.operating_point(int(sys.argv[4]) if len(sys.argv) > 4 else 2, bench.W, 1)
ip = F.img_params(width=bench.W, height=bench.H, padding=op.patch_size)
X = bench.synth_batch(B, 1, dev); Y = bench.synth_batch(B, 2, dev)
res = []
for trial in range(3):
    ofc = OFClass(op, ip, max_batch=B)
    if trial == 1: ofc.calc_batch(*Y)
    if trial == 2: ofc.calc_batch(*Y); ofc.calc_batch(*X); ofc.calc_batch(*Y)
    r = ofc.calc_batch(*X).clone()
    r2 = ofc.calc_batch(*X).clone()
    res.append(r)
    print("trial", trial, "repeat equal:", bool(torch.equal(r, r2)), "equal to trial 0:", bool(torch.equal(r, res[0])),
          "pairs differing from trial 0:", int(((r - res[0]).abs().flatten(1).max(1).values > 0).sum()), "finite:", bool(torch.isfinite(r).all()))
    ofc.close()
