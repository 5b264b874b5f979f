#!/usr/bin/env python3
"""Does the tile solver's time depend on where its arrays were allocated?  Several contexts in one process (kept alive, with
dummy allocations in between), one sor call per tall level timed in each: python tools/tile_addr.py"""
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
keep = []
for trial in range(6):
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
    keep.append(ofc)
    ofc.calc(a, b)
    torch.cuda.synchronize()
    row = []
    for lvl in (4, 3, 2):
        for _ in range(3): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t) / 20 * 1e6)
    print("context %d: sor call us  lvl4 %.1f  lvl3 %.1f  lvl2 %.1f" % (trial, *row), flush=True)
    keep.append(torch.empty((trial + 1) * 3_000_001, device="cuda"))
