#!/usr/bin/env python3
"""One sor_coupled call (the launch the refinement issues once per inner iteration) per level, us per call:
python tools/tile_call_time.py [W H op-point batch level ...]   default: the tile solver at the tall levels of a 4K pair
(3840 2160 4 1 4 3 2); 1920 1080 2 64 4 = the streaming solver of the headline workload.  Timing only: also meaningful for
the -DFOTG_TILE_DBG / -DFOTG_STREAM_DBG elimination builds"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
args = [int(x) for x in sys.argv[1:]] or [3840, 2160, 4, 1, 4, 3, 2]
W, H, OP, B = args[:4]
levels = args[4:]
f0, f1 = synth_pair(H, W, seed=5)
a = torch.from_numpy(np.stack([f0] * B)).cuda(); b = torch.from_numpy(np.stack([f1] * B)).cuda()
op = F.operating_point(OP, W, 1)
ofc = OFClass(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=B)
ofc.calc_batch(a, b)
torch.cuda.synchronize()
out = []
for lvl in levels:
    for _ in range(3): check(lib().fotg_bench_sor_call(ofc._h, lvl, B, None))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): check(lib().fotg_bench_sor_call(ofc._h, lvl, B, None))
    torch.cuda.synchronize()
    out.append("level %d: %.1f us" % (lvl, (time.perf_counter() - t) / 20 * 1e6))
print("; ".join(out))
