#!/usr/bin/env python3
"""per-wave statistics of the resident refinement kernel (a -DFOTG_RES_STATS build swapped in by tools/res_stats.sh): for pair 0,
cycles every wave of the solver workgroup waited at barriers / spun for data, and the data workers' wait for (du,dv) rows"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
f0, f1 = synth_pair(1080, 1920, seed=5)
a = torch.from_numpy(f0).cuda()[None].expand(B, -1, -1).contiguous()
b = torch.from_numpy(f1).cuda()[None].expand(B, -1, -1).contiguous()
op = F.operating_point(2, 1920, 1)
ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=B)
for _ in range(3):
    ofc.calc_batch(a, b)
torch.cuda.synchronize()
ptr = lib().fotg_ctx_counter(ofc._h, b"stamps_ptr")
st = np.zeros((4, 16, 4), np.int64)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(st.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), st.nbytes, 2)
names = {0: "solver WG"}
for role in range(4):
    print("role %d (%s)" % (role, "solver workgroup: waves 0-2 sweeps, helpers" if role == 0 else "data workgroup %d" % (role - 1)))
    for wv in range(16):
        w, t, r, sp = st[role, wv]
        if t:
            print("   wave %2d: total %8d cycles  barrier/poll wait %8d (%4.1f %%)  vmcnt wait %8d (%4.1f %%)  spin %8d" % (wv, t, w, 100.0 * w / t, r, 100.0 * r / t, sp))
