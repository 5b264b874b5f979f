#!/usr/bin/env python3
"""per-wave arrival / release times at four consecutive barriers of the stage kernel (-DFOTG_STAGE_STAMPS=2 build)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flowonthego_amd as F
from flowonthego_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from flowonthego_amd.oflow import OFClass
import bench
n = int(os.environ.get("BATCH", "64"))
I0, I1 = bench.synth_batch(n, 1234, torch.device("cuda"))
op = F.operating_point(2, 1920, 1)
ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=n)
out = ofc.new_outflow(n)
lib = F.lib()
for _ in range(3):
    ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize()
ptr = lib.fotg_ctx_counter(ofc._h, b"stage_stamps_ptr")
buf = np.zeros((16 * n, 16, 8), np.uint64)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(buf.ctypes.data, C.c_void_p(ptr), buf.nbytes, 2) == 0
buf = buf.astype(np.int64)
names = {0: "solver0", 4: "solver1", 1: "solver2", 3: "loader", 2: "writer", 5: "uv/zero"}
for t in (0, 2 * n):
    b = buf[t]
    t0 = b[:, 0][b[:, 0] > 0].min()
    print("ticket %d: per wave [arrive, release] at 4 consecutive barriers, us since the first arrival" % t)
    for wv in range(16):
        r = (b[wv] - t0) / 100.0
        print("  wave %2d %-13s" % (wv, names.get(wv, "data")), " ".join("[%6.2f %6.2f]" % (r[2 * k], r[2 * k + 1]) for k in range(4)))
