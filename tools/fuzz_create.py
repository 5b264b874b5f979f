#!/usr/bin/env python3
"""fuzz of fotg_create / one flow call: a valid configuration with ONE or TWO fields replaced by odd values (negative, zero or huge
sizes, patch sizes, scale ranges, iteration counts, weights, modes, batch sizes).  Every case must either be refused with a status
code or run to completion -- never crash or hang.  usage: python tools/fuzz_create.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ODD = {
    "patch_size": [0, 3, 5, 20, 64, -8, 4, 16],
    "patch_stride": [0.0, 1.0, 1.5, -0.5, 0.99, 0.01],
    "finest_scale": [0, 5, -1, 11],
    "coarsest_scale": [0, 1, 9, -1, 11],
    "grad_descent_iter": [1, 0, 128, -3, 1000],
    "min_iter": [0, 5, 2000, -7],
    "var_ref_iter": [0, 1, 5, 9, -2],
    "var_ref_sor_weight": [0.0, 2.5, -1.0, 1.0],
    "var_ref_alpha": [0.0, -1.0, 1e9, 1e-9],
    "var_ref_gamma": [0.0, -3.0, 1e9],
    "var_ref_delta": [0.0, 1e-9, -2.0],
    "dp_thresh": [0.0, -1.0, 1e9], "dr_thresh": [0.0, 2.0, -1.0], "res_thresh": [-1.0, 1e9],
    "sor_mode": [1, 2, 3, -1], "cost_func": [1, 2, 3, -1],
    "use_fbcon": [True], "depth_mode": [True], "u8_color": [1, 2, 3, -1], "fast_math": [True], "use_var_ref": [False],
    "use_mean_normalization": [False], "channels": [0, 2, 4, 3, 1],
}
refused = ran = nonfinite = 0
for k in range(cases):
    noc = int(rng.choice([1, 3]))
    w, h = int(rng.integers(100, 420)), int(rng.integers(80, 300))
    op = F.operating_point(int(rng.integers(1, 4)), w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 16)
    nb = 2
    for _ in range(int(rng.integers(1, 3))):
        what = str(rng.choice(list(ODD) + ["w", "h", "nb"]))
        if what == "w": w = int(rng.choice([0, 1, 7, 15, -5, 17]))
        elif what == "h": h = int(rng.choice([0, 1, 9, 15, -1, 33]))
        elif what == "nb": nb = int(rng.choice([0, -1, 1, 7]))
        else: setattr(op, what, type(getattr(op, what))(rng.choice(ODD[what])) if not isinstance(getattr(op, what), bool) else bool(rng.choice(ODD[what])))
    if os.environ.get("FUZZ_VERBOSE"):
        print(k, w, h, nb, {f: getattr(op, f) for f in list(ODD) + ["grad_descent_iter"]}, flush=True)
    try:
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=nb)
    except (F.FotgError, ValueError, OverflowError):
        refused += 1
        continue
    ch = op.channels
    shape = (nb, h, w) + ((ch,) if ch > 1 else ())
    f0 = (torch.rand(shape, device="cuda") * 255).floor()
    f1 = torch.roll(f0, 2, 2)
    try:
        out = ofc.calc_batch(f0, f1)
        torch.cuda.synchronize()
        ran += 1
        nonfinite += 0 if bool(torch.isfinite(out).all()) else 1
    except F.FotgError:
        refused += 1
    if os.environ.get("FOTG_DEBUG_GUARD"):
        from flowonthego_amd._lib import lib as _l
        gv = _l().fotg_ctx_counter(ofc._h, b"guard_violations")
        assert gv <= 0, ("guard violations", gv, k)
    ofc.close()
print("fuzz: %d cases, %d refused with a status code, %d ran (%d of them to a non-finite flow: degenerate weights), no crash" % (cases, refused, ran, nonfinite))
