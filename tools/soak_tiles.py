#!/usr/bin/env python3
"""Soak of the tile solver / level pipeline (vr_level_pipe_kernel, the default; FOTG_VR_LEVELPIPE=0: vr_sor_tile_kernel: levels of more
than 96 rows, workgroups handing diagonals over through memory behind progress words) WHILE a second context keeps HBM saturated with pyramid launches on a stream of its own -- the
condition of the headline configuration (several batches in flight) for users of operating points 3 / 4.  Every call must
equal the result computed on an idle GPU, and no bounded wait may time out.
usage: python tools/soak_tiles.py [calls] [op-point] [width] [height] [batch]      (writes a summary line as JSON)"""
import ctypes as C, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
oppt = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
def mkstream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s
p = lambda t: C.c_void_p(t.data_ptr())
# the disturber: batch-64 1080p pyramids (1.07 GB per launch at ~6.8 TB/s) back to back on its own stream
opd = F.operating_point(2, bench.W, 1)
D = OFClass(opd, F.img_params(width=bench.W, height=bench.H, padding=opd.patch_size), max_batch=64)
J0, J1 = bench.synth_batch(64, 7, dev)
sD, sT = mkstream(), mkstream()
# the context under test
op = F.operating_point(oppt, w, 1)
if oppt == 4:
    op.grad_descent_iter = 16                  # (the LK iteration count does not matter here; keeps a call short)
T = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=nb)
prs = [synth_pair(h, w, seed=90 + k) for k in range(nb)]
I0 = torch.from_numpy(np.stack([q[0] for q in prs])).to(dev); I1 = torch.from_numpy(np.stack([q[1] for q in prs])).to(dev)
ref = T.calc_batch(I0, I1).clone()
torch.cuda.synchronize()
before = lib().fotg_debug_counter(b"sor_tiles") + lib().fotg_debug_counter(b"level_pipe")
out = T.new_outflow(nb)
bad = 0
t0 = time.perf_counter()
CH = 50
for k in range(0, calls, CH):
    for _ in range(CH):
        check(lib().fotg_pyramid_pair(D._h, 64, p(J0), p(J1), 1, sD))             # ~0.16 ms of saturated HBM each
        check(lib().fotg_pyramid_pair(D._h, 64, p(J0), p(J1), 1, sD))
        check(lib().fotg_calc_batch(T._h, nb, p(I0), p(I1), None, p(out), sT))
    assert hip.hipStreamSynchronize(sT) == 0
    if not torch.equal(out, ref):
        bad += 1
        print("calls", k, "..", k + CH, ": result differs, max abs", float((out - ref).abs().max()), flush=True)
assert hip.hipStreamSynchronize(sD) == 0
el = time.perf_counter() - t0
tiles = lib().fotg_debug_counter(b"sor_tiles") + lib().fotg_debug_counter(b"level_pipe") - before     # launches with inter-workgroup waits (FOTG_VR_LEVELPIPE=0: one per sor_coupled call)
res = {"tool": "tools/soak_tiles.py", "calls": calls, "op_point": oppt, "size": [w, h], "pairs_per_call": nb, "seconds": round(el, 1),
       "tile_or_level_pipe_launches": int(tiles), "level_pipe": os.environ.get("FOTG_VR_LEVELPIPE", "1") != "0", "mismatching_checks": bad, "checks": calls // CH,
       "tile_timeouts": int(lib().fotg_ctx_counter(T._h, b"tile_timeouts")), "stalls_reported": int(lib().fotg_ctx_counter(T._h, b"stalls")),
       "disturber": "2 x fotg_pyramid_pair(batch 64, 1080p f32) per call on a second stream (HBM saturated)"}
print(json.dumps(res))
sys.exit(1 if (bad or res["tile_timeouts"] or res["stalls_reported"] or tiles == 0) else 0)
