#!/usr/bin/env python3
"""debug helper for the stage-pipelined refinement: one level through fotg_varref with the pipeline on and off, compared
bit for bit (the off path is the oracle-checked round-1 path), mismatches summarised by anti-diagonal; plus timing."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass

def run(w, h, n, stage, mins="24", reps=0, noc=1, width_for_op=None):
    os.environ["FOTG_VR_STAGE"] = stage
    os.environ["FOTG_VR_STAGE_MINS"] = mins
    op = F.operating_point(2, width_for_op or w, noc)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
    ps = [synth_pair(h, w, seed=5 + k, noc=noc) for k in range(min(n, 4))]
    I0 = torch.from_numpy(np.stack([ps[k % len(ps)][0] for k in range(n)])).cuda()
    I1 = torch.from_numpy(np.stack([ps[k % len(ps)][1] for k in range(n)])).cuda()
    out = ofc.calc_batch(I0, I1)
    torch.cuda.synchronize()
    ms = None
    if reps:
        t0 = time.perf_counter()
        for _ in range(reps):
            ofc.calc_batch(I0, I1, None, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    to = F.lib().fotg_ctx_counter(ofc._h, b"vr_stage_timeouts")
    res = out.cpu().numpy()
    ofc.close()
    return res, ms, to

for (w, h, n, noc, wop) in ((1024, 436, 1, 1, None), (1920, 1080, 2, 1, None), (328, 200, 1, 3, None), (1920, 1080, 64, 1, None)):
    a, msa, _ = run(w, h, n, "0", reps=10 if n == 64 else 0, noc=noc, width_for_op=wop)
    b, msb, to = run(w, h, n, "1", reps=10 if n == 64 else 0, noc=noc, width_for_op=wop)
    bad = (a != b).any(-1)
    print("%dx%d n=%d noc=%d: equal=%s mismatching px=%d timeouts=%s  ms off/on: %s / %s" % (w, h, n, noc, not bad.any(), int(bad.sum()), to, msa, msb), flush=True)
    if bad.any():
        k, jj, ii = np.nonzero(bad)
        diag = ii + jj
        print("   pairs", np.unique(k)[:8], "diagonals min/max", diag.min(), diag.max(), "rows min/max", jj.min(), jj.max(),
              "max abs diff", float(np.abs(a - b).max()), "nan", bool(np.isnan(b).any()))
        hist = np.bincount(diag // 4)
        print("   per chunk:", hist.tolist()[:60])
