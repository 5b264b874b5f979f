#!/usr/bin/env python3
"""One sor_coupled call of the tile solver at the tall levels of a 4K pair (operating point 4), us per call:
python tools/tile_call_time.py   (timing only: also meaningful for the -DFOTG_TILE_DBG elimination builds)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
f0, f1 = synth_pair(2160, 3840, seed=5)
a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
op = F.operating_point(4, 3840, 1)
ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
ofc.calc(a, b)
torch.cuda.synchronize()
out = []
for lvl in (4, 3, 2):
    for _ in range(3): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
    torch.cuda.synchronize()
    out.append("level %d: %.1f us" % (lvl, (time.perf_counter() - t) / 20 * 1e6))
print("; ".join(out))
