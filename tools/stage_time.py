#!/usr/bin/env python3
"""time the refinement of one level (fotg_varref alone, HIP events) for a library variant: tools/stage_time.py [lib.so]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flowonthego_amd as F
from flowonthego_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from flowonthego_amd.oflow import OFClass
import bench
n = int(os.environ.get("BATCH", "64"))
I0, I1 = bench.synth_batch(n, 1234, torch.device("cuda"))
op = F.operating_point(2, 1920, 1)
ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=n)
out = ofc.new_outflow(n)
lib = F.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
st = bench.stage_breakdown(ofc, I0, I1, out, lib, sp, reps=10)
print(os.path.basename(_lib.LIB_PATH), {k: round(v * 1e3, 1) for k, v in st.items() if k.startswith("varref")}, "timeouts", lib.fotg_ctx_counter(ofc._h, b"vr_stage_timeouts"), flush=True)
