#!/usr/bin/env python3
"""FOTG_DEBUG builds with FOTG_DEBUG_GUARD=1: run the headline batch (and a few other shapes) and count buffers that were written
past their end: python tools/guard_check.py"""
import os, sys
os.environ["FOTG_DEBUG_GUARD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib
from conftest import synth_pair
dev = torch.device("cuda", 0)
def run(w, h, opp, B, frames, fast=False, colour=False):
    op = F.operating_point(opp, w, 1)
    op.fast_math = fast
    op.u8_color = 1 if colour else 0
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=B)
    if colour:
        frames = tuple(torch.stack([f.to(torch.uint8)] * 3, -1).contiguous() for f in frames)
    for _ in range(3): (ofc.calc_batch_u8 if colour else ofc.calc_batch)(*frames)
    torch.cuda.synchronize()
    print("%dx%d op-pt %d batch %d%s%s: guard violations %d" % (w, h, opp, B, " fast_math" if fast else "", " u8 colour" if colour else "",
                                                              lib().fotg_ctx_counter(ofc._h, b"guard_violations")), flush=True)
    ofc.close()
run(1920, 1080, 2, 64, bench.synth_batch(64, 1, dev))
f0, f1 = synth_pair(1080, 1920, seed=3)
a, b = torch.from_numpy(f0).cuda()[None], torch.from_numpy(f1).cuda()[None]
run(1920, 1080, 2, 1, (a, b))
run(1920, 1080, 3, 1, (a, b))
run(1920, 1080, 2, 1, (a, b), fast=True)
run(1920, 1080, 3, 1, (a, b), fast=True)
run(1920, 1080, 2, 1, (a, b), colour=True)
f0, f1 = synth_pair(436, 1024, seed=3)
run(1024, 436, 2, 1, (torch.from_numpy(f0).cuda()[None], torch.from_numpy(f1).cuda()[None]))
f0, f1 = synth_pair(2160, 3840, seed=5)                 # BASELINE configs[3]: the level pipeline of the tall levels
a, b = torch.from_numpy(f0).cuda()[None], torch.from_numpy(f1).cuda()[None]
run(3840, 2160, 4, 1, (a, b))
run(3840, 2160, 4, 1, (a, b), fast=True)
