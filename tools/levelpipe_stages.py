#!/usr/bin/env python3
"""refinement stage times of the 4K operating-point-4 pair: default path against FOTG_VR_LEVELPIPE=1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
import bench
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
lib = F.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
f0, f1 = synth_pair(2160, 3840, seed=5)
a, b = torch.from_numpy(f0).cuda()[None], torch.from_numpy(f1).cuda()[None]
for lp in ("0", os.environ.get("LP_MODE", "1")):
    os.environ["FOTG_VR_LEVELPIPE"] = lp
    op = F.operating_point(4, 3840, 1)
    op.fast_math = "fast" in sys.argv
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
    out = ofc.new_outflow(1)
    st = bench.stage_breakdown(ofc, a, b, out, lib, sp, reps=5)
    ms = bench.timed(lambda: ofc.calc_batch(a, b, None, out), torch.cuda.synchronize, 10) * 1e3
    print("LEVELPIPE=%s: %.3f ms per pair; varref" % (lp, ms), {k: round(v, 4) for k, v in st.items() if k.startswith("varref")}, flush=True)
    ofc.close()
