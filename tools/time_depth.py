#!/usr/bin/env python3
"""Depth-mode (stereo) throughput on the bench frames: python tools/time_depth.py [batch] [steps]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from bench import synth_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 1920, 1080
I0, I1 = synth_batch(batch, 0, torch.device("cuda", 0))
op = F.operating_point(2, W, 1)
op.depth_mode = True
ofc = OFClass(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=batch)
out = ofc.new_outflow(batch)
for _ in range(3):
    ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize()
el = time.perf_counter() - t
print("depth mode: %.1f pairs/s, %.3f ms/step" % (batch * steps / el, el / steps * 1e3))
