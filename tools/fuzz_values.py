#!/usr/bin/env python3
"""fuzz of the INPUT VALUES: valid configurations (operating points 1-3, gray / RGB, optical flow / depth, forward-backward merge,
tolerance mode, odd sizes), frames with a sprinkle of NaN / inf / +-1e30 / +-3e38 pixels or wholly constant / saturated, and an
`initflow` of random extremes.  Nothing may crash or hang; the flows are whatever the reference's arithmetic gives.
usage: python tools/fuzz_values.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BAD = [float("nan"), float("inf"), -float("inf"), 1e30, -1e30, 3e38, -3e38, 3e9, -3e9, 1e-40, 0.0]
ran = refused = 0
for k in range(cases):
    noc = int(rng.choice([1, 3]))
    w, h = int(rng.integers(120, 520)), int(rng.integers(100, 400))
    op = F.operating_point(int(rng.integers(1, 4)), w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 16)
    op.depth_mode = bool(rng.random() < 0.25)
    op.use_fbcon = bool(rng.random() < 0.3)
    op.fast_math = bool(rng.random() < 0.3)
    if rng.random() < 0.3:
        op.finest_scale = 0
        op.coarsest_scale = int(rng.integers(1, 4))
    nb = int(rng.integers(1, 4))
    if os.environ.get("FUZZ_VERBOSE"):
        print(k, w, h, noc, nb, op.patch_size, op.coarsest_scale, op.finest_scale, op.depth_mode, op.use_fbcon, op.fast_math, flush=True)
    try:
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=nb)
    except F.FotgError:
        refused += 1
        continue
    shape = (nb, h, w) + ((noc,) if noc > 1 else ())
    mode = int(rng.integers(0, 4))
    f0 = (torch.rand(shape, device="cuda") * 255).floor()
    f1 = torch.roll(f0, 2, 2)
    if mode == 0:                                   # a sprinkle of bad pixels
        for f in (f0, f1):
            m = torch.rand(shape, device="cuda") < float(rng.choice([1e-4, 1e-2, 0.3]))
            f[m] = float(rng.choice(BAD))
    elif mode == 1:                                 # constant / saturated frames
        f0 = torch.full(shape, float(rng.choice(BAD + [255.0, 17.0])), device="cuda"); f1 = f0.clone()
    elif mode == 2:                                 # huge dynamic range
        f0 = f0 * float(rng.choice([1e30, 1e-30, 1e36])); f1 = f1 * float(rng.choice([1e30, 1e-30, 1.0]))
    init = None
    if rng.random() < 0.6:
        sc = op.coarsest_scale + 1
        init = (torch.randn((nb, ofc.height >> sc, ofc.width >> sc, 1 if op.depth_mode else 2), device="cuda") * float(rng.choice([1.0, 50.0, 1e6, 1e30, 3e38])))
        if rng.random() < 0.5:
            init[torch.rand(init.shape, device="cuda") < 0.05] = float(rng.choice(BAD))
        init = init.contiguous()
    try:
        out = ofc.calc_batch(f0, f1, init)
        full = ofc.upsample_crop(out)
        torch.cuda.synchronize()
        ran += 1
    except F.FotgError:
        refused += 1
    if os.environ.get("FOTG_DEBUG_GUARD"):
        from flowonthego_amd._lib import lib as _l
        gv = _l().fotg_ctx_counter(ofc._h, b"guard_violations")
        assert gv <= 0, ("guard violations", gv, k)
    ofc.close()
print("fuzz_values: %d cases, %d ran, %d refused, no crash" % (cases, ran, refused))
