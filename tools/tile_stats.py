#!/usr/bin/env python3
"""-DFOTG_TILE_STATS builds: per-tile timeline of one tile-solver call per tall level, for several contexts (allocation
addresses differ): python tools/tile_stats.py"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
f0, f1 = synth_pair(2160, 3840, seed=5)
a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
op = F.operating_point(4, 3840, 1)
keep = []
hip = ctypes.CDLL("libamdhip64.so")
for trial in range(4):
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
    keep.append(ofc)
    ofc.calc(a, b)
    torch.cuda.synchronize()
    for lvl in (4, 2):
        for _ in range(3): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): check(lib().fotg_bench_sor_call(ofc._h, lvl, 1, None))
        torch.cuda.synchronize()
        us = (time.perf_counter() - t) / 10 * 1e6
        ptr = lib().fotg_ctx_counter(ofc._h, b"stamps_ptr")
        st = np.zeros((64, 32), np.int64)
        hip.hipMemcpy(st.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), st.nbytes, 2)
        st = st[st[:, 0] != 0]
        t0 = st[:, 0].min()
        print("context %d level %d: %.1f us per call, %d tiles" % (trial, lvl, us, len(st)))
        for r in st[np.argsort(st[:, 0])]:
            n, bb = (r[7] >> 16) & 0xffff, r[7] & 0xffff
            print("   (n %d, b %d) xcc %d se %d cu %2d  start %6.1f end %6.1f us  spins own %5d below %5d top %5d  writer done %6.1f | at step 0, 128, ..: %s" % (
                n, bb, r[11] & 15, (r[10] >> 13) & 7, (r[10] >> 8) & 15, (r[0] - t0) / 100, (r[1] - t0) / 100, r[2], r[3], r[4], (r[9] - t0) / 100,
                " ".join("%5.1f" % ((x - t0) / 100) for x in r[16:32] if x)))
    keep.append(torch.empty((trial + 1) * 3_000_001, device="cuda"))
