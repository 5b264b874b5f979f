#!/usr/bin/env python3
"""debug: resident refinement pipeline vs the launch-per-iteration path on the 120 x 68 golden level, for 1..5 inner iterations"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass, VarRefClass
from conftest import load_fdf

c = load_fdf(1)["w120h68"]
im1, im2, wx, wy = c["im1"], c["im2"], c["wx"], c["wy"]
_, h, w = im1.shape
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
ps = 8
padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
res = {}
for lvl in range(0, 5):
    for mode in ("0", "1"):
        os.environ["FOTG_VR_RESIDENT"] = mode
        op = F.operating_point(2, 1024, 1)
        op.coarsest_scale = op.finest_scale = lvl
        ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8))
        flow = dev(np.stack([wx, wy], -1))[None].contiguous()
        F.lib().fotg_enable_taps(ofc._h, 1)
        VarRefClass(dev(padlvl(im1))[None], dev(padlvl(im2))[None], ofc.iparams[0], ofc.op, flow)
        st = ((w + 3) // 4) * 4
        def plane(nm):
            buf = np.zeros((1, h, st), np.float32)
            F._lib.check(F.lib().fotg_varref_plane(ofc._h, 0, nm.encode(), lvl, buf.ctypes.data))
            return buf[0, :, :w]
        res[(lvl, mode)] = {"out": flow[0].cpu().numpy(), **{nm: plane(nm) for nm in ("sh", "sv", "b1", "b2", "a11", "du", "dv", "Ix", "mask")}}
        print("lvl", lvl, "resident", mode, "timeouts", F.lib().fotg_ctx_counter(ofc._h, b"vr_res_timeouts"), "launches", F.lib().fotg_debug_counter(b"vr_resident"))
    a, b = res[(lvl, "0")], res[(lvl, "1")]
    for k in a:
        d = np.abs(a[k] - b[k])
        bad = np.argwhere(d > 0)
        print("  inner %d  %-4s max |diff| %.3g  mismatching cells %d  first %s  diag range %s" % (
            lvl + 1, k, d.max(), len(bad), bad[:3].tolist(), (int((bad[:, 0] + bad[:, 1]).min()), int((bad[:, 0] + bad[:, 1]).max())) if len(bad) and bad.shape[1] == 2 else None))
