#!/usr/bin/env python3
"""FOTG_VR_LEVELPIPE=1 (all inner iterations of a tall level in one pipeline launch, csrc/varref_levelpipe.hip.h) against the default
path: same bits?  time per pair.   python tools/levelpipe_probe.py [small] [4k] [fast]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
L = F.lib()
what = sys.argv[1:] or ["small"]
fast = "fast" in what


def run(w, h, oppt, n=1, reps=5, seed=5):
    f0, f1 = synth_pair(h, w, seed=seed)
    a, b = torch.from_numpy(np.stack([f0] * n)).cuda(), torch.from_numpy(np.stack([f1] * n)).cuda()
    res = {}
    for lp in ("0", "1"):
        os.environ["FOTG_VR_LEVELPIPE"] = lp if lp == "0" else os.environ.get("LP_MODE", "1")
        op = F.operating_point(oppt, w, 1)
        op.fast_math = fast
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        before = L.fotg_debug_counter(b"level_pipe")
        out = ofc.calc_batch(a, b).clone()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            ofc.calc_batch(a, b)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        res[lp] = (out, ms, L.fotg_debug_counter(b"level_pipe") - before, L.fotg_ctx_counter(ofc._h, b"tile_timeouts"), L.fotg_ctx_counter(ofc._h, b"stalls"))
        ofc.close()
    if n > 1:
        print("   pairs equal to pair 0: default", [bool(torch.equal(res["0"][0][k], res["0"][0][0])) for k in range(n)],
              "level pipe", [bool(torch.equal(res["1"][0][k], res["0"][0][0])) for k in range(n)])
    same = bool(torch.equal(res["0"][0], res["1"][0]))
    d = (res["0"][0] - res["1"][0]).abs().max().item()
    print("%dx%d op-pt %d n=%d%s: default %.3f ms, level pipe %.3f ms (launches %d, timeouts %d, stalls %d)  same bits: %s  max |d| %g" %
          (w, h, oppt, n, " fast_math" if fast else "", res["0"][1], res["1"][1], res["1"][2], res["1"][3], res["1"][4], same, d), flush=True)


if "small" in what:
    run(1024, 1024, 2)
    run(640, 528, 3)
    run(1920, 1080, 3, reps=3)
    run(1024, 1024, 2, n=2)
if "multi" in what:
    run(3840, 2160, 4, n=2, reps=2)
    run(1920, 1080, 3, n=8, reps=2)
if "4k" in what:
    run(3840, 2160, 4, reps=5)
    run(3840, 2160, 4, n=4, reps=3)
