#!/usr/bin/env python3
"""dump the s_memrealtime stamps of a -DFOTG_STAGE_STAMPS build: tools/stage_stamps.py tools/exp/libfotg_st0.so"""
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
inner = 5
buf = np.zeros((16 * n, 16, 8), np.uint64)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(buf.ctypes.data, C.c_void_p(ptr), buf.nbytes, 2) == 0
buf = buf[:inner * n].astype(np.int64)
t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
r0 = buf[0, 3]; r3 = buf[3 * n, 3] if inner > 3 else r0
print("   stage-0 loader stamps (I=8,24,40,56,end):", [round(us(x), 1) for x in r0[1:5]], round(us(buf[0, :, 6].max()), 1))
print("SUMMARY %s: stage-0 interval %.3f us, stage-3 interval %.3f us, stage0 end %.1f, last end %.1f" % (os.path.basename(sys.argv[1]), (r0[4] - r0[1]) / 100.0 / 48, (r3[4] - r3[1]) / 100.0 / 48, us(buf[0, :, 6].max()), us(buf[:, :, 6].max())))
if os.environ.get("BRIEF"):
    sys.exit(0)
print("level 4 stage kernel, batch", n, "-- times in us since the first workgroup started; slots: start | I=8 | I=24 | I=40 | I=56 | solver-done | end")
for stage in range(inner):
    for pair in (0, n - 1):
        t = stage * n + pair
        print("stage %d pair %d (ticket %d)" % (stage, pair, t))
        for wv, name in ((0, "solver0"), (2, "solver2"), (3, "loader"), (7, "writer"), (4, "data4"), (9, "data9")):
            r = buf[t, wv]
            print("   %-8s" % name, " ".join("%8.1f" % us(x) if x > 0 else "       -" for x in r[:7]))
starts = us(buf[:, 0, 0]); ends = us(buf[:, :, 6].max(1))
for stage in range(inner):
    sl = slice(stage * n, (stage + 1) * n)
    print("stage %d: start min/med/max %.1f %.1f %.1f   end min/med/max %.1f %.1f %.1f" % (stage, starts[sl].min(), np.median(starts[sl]), starts[sl].max(), ends[sl].min(), np.median(ends[sl]), ends[sl].max()))
