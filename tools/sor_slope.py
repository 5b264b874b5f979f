#!/usr/bin/env python3
"""Per-diagonal cost of one sor_coupled call at level 4 (68 rows): calls timed at three widths, the slope is the time per
anti-diagonal step and the intercept the fixed cost of a launch: python tools/sor_slope.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pts = []
for w in (1920, 2880, 3840):
    f0, f1 = synth_pair(1080, w, seed=5)
    a = torch.from_numpy(f0).cuda()[None].expand(B, -1, -1).contiguous()
    b = torch.from_numpy(f1).cuda()[None].expand(B, -1, -1).contiguous()
    op = F.operating_point(2, 1920, 1)
    ofc = OFClass(op, F.img_params(width=w, height=1080, padding=op.patch_size), max_batch=B)
    ofc.calc_batch(a, b)
    torch.cuda.synchronize()
    for _ in range(5): check(lib().fotg_bench_sor_call(ofc._h, 4, B, None))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): check(lib().fotg_bench_sor_call(ofc._h, 4, B, None))
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 50 * 1e6
    S = w // 16 + 68 - 1
    pts.append((S, us))
    print("width %d: S = %d diagonals, %.2f us per call" % (w, S, us))
    ofc.close()
(s0, t0), (s1, t1), (s2, t2) = pts
slope = (t2 - t0) / (s2 - s0)
print("slope %.1f ns per diagonal, intercept %.2f us" % (slope * 1e3, t0 - slope * s0))
