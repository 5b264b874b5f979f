#!/usr/bin/env python3
"""per-stage GPU time of one flow call (each stage alone between HIP events): python tools/stage_times.py [W H op-point batch channels]
default: BASELINE configs[3], one 3840x2160 pair at operating point 4"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
import bench
W, H, OP, B, NOC = (int(x) for x in (sys.argv[1:6] + ["3840", "2160", "4", "1", "1"][len(sys.argv) - 1:]))
f0, f1 = synth_pair(H, W, seed=5, noc=NOC)
op = F.operating_point(OP, W, NOC)
ofc = OFClass(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=B)
rep = (B, 1, 1) + ((1,) if NOC > 1 else ())
a = torch.from_numpy(f0).cuda()[None].repeat(*rep).contiguous(); b = torch.from_numpy(f1).cuda()[None].repeat(*rep).contiguous()
out = ofc.new_outflow(B)
for _ in range(2):
    ofc.calc_batch(a, b, None, out)
torch.cuda.synchronize()
st = bench.stage_breakdown(ofc, a, b, out, F.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream), reps=5)
print(json.dumps({"config": [W, H, OP, B, NOC], "sum_ms": round(sum(st.values()), 4), "stage_ms": {k: round(v, 4) for k, v in st.items()}}))
