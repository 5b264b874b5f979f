#!/usr/bin/env python3
"""Robustness sweep (test infrastructure, run by hand on a GPU box; pytest does not collect it): random frame sizes /
operating points / channels / switches / entry points, GPU result against the oracle, bit for bit.
usage: python tests/fuzz_sizes.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from oracle import oracle as O
from test_gpu_parity import oracle_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for k in range(n):
    w, h = int(rng.integers(120, 2000)), int(rng.integers(100, 1200))
    op_point, noc = int(rng.integers(1, 5)), 1 + 2 * int(rng.integers(0, 2))
    if op_point == 4 and w * h > 700000:                      # the quality preset runs down to level 1: keep the oracle quick
        w, h = w // 2, h // 2
    if w * h * noc > 2.2e6 and rng.integers(0, 4):          # big RGB frames (slow oracle) only now and then
        noc = 1
    f0, f1 = synth_pair(h, w, seed=1000 + k, noc=noc)
    op = F.operating_point(op_point, w, noc)
    if op_point == 4:
        op.grad_descent_iter = 6                               # 128 in the preset; the solver / pyramid paths are what this sweeps
    op.cost_func = int(rng.integers(0, 3)); op.use_fbcon = bool(rng.integers(0, 4) == 0)
    if rng.integers(0, 4) == 0:                                # a custom parameter set: other patch size / overlap / iteration count
        op.patch_size = int(rng.choice([4, 8, 12, 16]))
        op.patch_stride = float(rng.choice([0.3, 0.5, 0.75]))
        op.grad_descent_iter = int(rng.integers(2, 10))
        op.use_mean_normalization = bool(rng.integers(0, 2))
        op.min_iter = int(rng.integers(0, op.grad_descent_iter + 1))
        op.res_thresh = float(rng.choice([0.0, 0.5, 2.0]))
        op.dr_thresh = float(rng.choice([0.95, 1.5]))
        op.var_ref_iter = int(rng.integers(1, 5))
        op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta = float(rng.choice([10.0, 3.0])), float(rng.choice([10.0, 0.0, 4.0])), float(rng.choice([5.0, 0.0]))
        op.var_ref_sor_weight = float(rng.choice([1.6, 1.0, 1.9]))
    op.depth_mode = bool(rng.integers(0, 3) == 0)             # stereo depth mode (one displacement channel)
    try:
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    except F.FotgError as e:
        print(k, (w, h, op_point, noc), "refused:", e); continue
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    entry = int(rng.integers(0, 5))                          # 0: calc (f32), 1: 8-bit frames, 2: batch of two (pair, swapped pair),
                                                             # 3: video entry point (f0, f1, f0), 4: initflow warm start
    if entry == 1:
        out = ofc.calc_batch_u8(dv(f0.astype(np.uint8))[None], dv(f1.astype(np.uint8))[None])[0].cpu().numpy()
        ok = np.array_equal(out, ref)
    elif entry == 2:
        ofc.close()
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        o2 = ofc.calc_batch(dv(np.stack([f0, f1])), dv(np.stack([f1, f0]))).cpu().numpy()
        out = o2[0]
        ok = np.array_equal(out, ref) and np.array_equal(o2[1], O.flow(O.pad_frame(f1, p.sc_f), O.pad_frame(f0, p.sc_f), p, 0))
    elif entry == 3:
        ofc.close()
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        o2 = ofc.calc_sequence(dv(np.stack([f0, f1, f0]))).cpu().numpy()
        out = o2[0]
        ok = np.array_equal(out, ref) and np.array_equal(o2[1], O.flow(O.pad_frame(f1, p.sc_f), O.pad_frame(f0, p.sc_f), p, 0))
    elif entry == 4:
        wp, hp, _, _ = O.padded_size(w, h, p.sc_f)
        nch = 1 if op.depth_mode else 2
        init = (rng.standard_normal((hp >> (p.sc_f + 1), wp >> (p.sc_f + 1), nch)) * 0.7).astype(np.float32)
        if op.depth_mode:
            init = -np.abs(init)
        P0, P1 = O.Pyramid(O.pad_frame(f0, p.sc_f), p.sc_f, p.ps), O.Pyramid(O.pad_frame(f1, p.sc_f), p.sc_f, p.ps)
        ref = O.flow_pyr(P0, P1, p, initflow=init)
        out = ofc.calc_batch(dv(f0)[None], dv(f1)[None], initflow=dv(init)[None])[0].cpu().numpy()
        ok = np.array_equal(out, ref)
    else:
        out = ofc.calc(dv(f0), dv(f1)).cpu().numpy()
        ok = np.array_equal(out, ref)
    bad += not ok
    print(k, (w, h, op_point, noc, op.cost_func, op.use_fbcon, op.depth_mode, "ps %d" % op.patch_size, "entry %d" % entry), "rows@finest", ref.shape[0], "OK" if ok else "MISMATCH max %g" % np.abs(out - ref).max(), flush=True)
    ofc.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
