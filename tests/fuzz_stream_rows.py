"""Sweep of frame heights around the streaming solver's row range (test infrastructure, run by hand on a GPU box).
usage: python tests/fuzz_stream_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from oracle import oracle as O
from test_gpu_parity import oracle_params
rng = np.random.default_rng(3)
bad = 0
before = F.lib().fotg_debug_counter(b"sor_stream")
for k in range(14):
    w = int(rng.integers(1300, 2400)); h = int(rng.integers(1030, 1540))
    f0, f1 = synth_pair(h, w, seed=2000 + k)
    op = F.operating_point(2, w, 1)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = ofc.calc_batch(dv(np.stack([f0, f1])), dv(np.stack([f1, f0]))).cpu().numpy()
    p = oracle_params(O, op)
    a, b = O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f)
    ok = np.array_equal(out[0], O.flow(a, b, p, 0)) and np.array_equal(out[1], O.flow(b, a, p, 0))
    bad += not ok
    print(k, (w, h), "finest", out.shape[1:3], "scales", op.coarsest_scale, op.finest_scale, "OK" if ok else "MISMATCH", flush=True)
    ofc.close()
print("stream launches:", F.lib().fotg_debug_counter(b"sor_stream") - before, "mismatches:", bad)
