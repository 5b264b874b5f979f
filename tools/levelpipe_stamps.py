#!/usr/bin/env python3
"""time line of the roles of the level pipeline at the finest level of the 4K operating-point-4 pair (FOTG_VR_LEVELPIPE=65: stamps)"""
import os, sys, ctypes
os.environ["FOTG_VR_LEVELPIPE"] = "65"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib
f0, f1 = synth_pair(2160, 3840, seed=5)
a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
op = F.operating_point(4, 3840, 1)
ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
for _ in range(3):
    ofc.calc(a, b)
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
ptr = lib().fotg_ctx_counter(ofc._h, b"stamps_ptr")
st = np.zeros((8192, 8), np.int64)
hip.hipMemcpy(st.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), st.nbytes, 2)
st = st[st[:, 0] != 0]
t0 = st[:, 0].min()
rows = []
for r in st:
    isdata = (r[2] >> 60) & 1
    kc = (r[2] >> 40) & 0xff
    if isdata:
        rows.append((r[0], "data  k %d ty %2d" % (kc, r[2] & 0xfffff), r))
    else:
        rows.append((r[0], "tile  k %d n %d b %d" % (kc, (r[2] >> 20) & 0x3ff, (r[2] >> 30) & 0x3ff), r))
print("%d roles, whole launch %.1f us" % (len(rows), (st[:, 1].max() - t0) / 100))
if os.environ.get("FOTG_STAMPS_BRIEF"):
    # pace (us per step) of every call's first and last tile role over steps 0-512 / 512-1024 / 1024-end
    for _, name, r in sorted(rows, key=lambda x: (x[2][2] >> 40) & 0xff):
        if name.startswith("tile") and (name.endswith("n 0 b 0") or name.endswith("n 2 b 8") or name.endswith("n 0 b 4")):
            t = [(x - t0) / 100 for x in (r[4], r[5], r[6], r[1])]
            print("  %-20s first step %6.1f  pace %.3f %.3f %.3f  end %6.1f" % (name, t[0], (t[1] - t[0]) / 512, (t[2] - t[1]) / 512, (t[3] - t[2]) / 479, t[3]))
    sys.exit(0)
def where(r):
    v = int(r[7]) & 0xffffffffffffffff
    xcc, hw, nb = v >> 60, (v >> 44) & 0xffff, v & 0xfffffffffff
    # HW_ID: wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13]
    return "xcc %d cu %d.%d.%2d | waits with polls of its own %3d (own %2d below %2d top %2d data %2d)" % (
        xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, nb & 0xff, (nb >> 8) & 63, (nb >> 14) & 63, (nb >> 20) & 63, (nb >> 26) & 63)
cus = {}
for _, name, r in rows:
    v = int(r[7]) & 0xffffffffffffffff
    cus.setdefault((v >> 60, (v >> 52) & 0xff), []).append(name)          # (xcc, se | sh | cu)
print("CUs in use: %d; CUs holding more than one role: %d" % (len(cus), sum(1 for v in cus.values() if len(v) > 1)))
for _, name, r in sorted(rows, key=lambda x: (x[2][2] >> 40) & 0xff):
    if "data" in name and (r[2] & 0xfffff) % 8 not in (0, 7):
        continue
    print("  %-22s start %7.1f end %7.1f | %s | %s" % (name, (r[0] - t0) / 100, (r[1] - t0) / 100, " ".join("%7.1f" % ((x - t0) / 100) for x in r[3:7] if x), where(r)))
