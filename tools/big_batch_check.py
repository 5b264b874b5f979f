#!/usr/bin/env python3
"""index-width check: batches whose frame tensors exceed 2^31 and 2^32 bytes / 2^31 elements (1080p x 600 and x 1100 pairs, 4K x 80,
8-bit 1080p x 1200): the first, a middle and the LAST pair of the batch must equal the same pair computed alone.
usage: python tools/big_batch_check.py [--huge]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
def run(w, h, opp, n, u8=False, noc=1):
    op = F.operating_point(opp, w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 12)
    base = [synth_pair(h, w, seed=60 + k, noc=noc) for k in range(3)]
    dt = torch.uint8 if u8 else torch.float32
    conv = (lambda a: torch.from_numpy(a.astype(np.uint8))) if u8 else torch.from_numpy
    b0 = [conv(p[0]).cuda() for p in base]; b1 = [conv(p[1]).cuda() for p in base]
    shape = (n, h, w) + ((noc,) if noc > 1 else ())
    I0 = torch.empty(shape, dtype=dt, device="cuda"); I1 = torch.empty(shape, dtype=dt, device="cuda")
    for k in range(n):
        I0[k] = b0[k % 3]; I1[k] = b1[k % 3]
    one = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=1)
    ref = [(one.calc_batch_u8 if u8 else one.calc_batch)(b0[k][None], b1[k][None])[0].clone() for k in range(3)]
    big = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
    out = (big.calc_batch_u8 if u8 else big.calc_batch)(I0, I1)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    out = (big.calc_batch_u8 if u8 else big.calc_batch)(I0, I1, None, out)
    torch.cuda.synchronize()
    rate = n / (time.perf_counter() - t0)
    bad = [k for k in range(n) if not torch.equal(out[k], ref[k % 3])]
    print("%dx%d op-pt %d, %d pairs%s: frame tensor %.2f GB, %.2f G elements: %s" % (w, h, opp, n, " (8-bit)" if u8 else "", I0.numel() * I0.element_size() / 1e9, I0.numel() / 1e9,
                                                                              ("all pairs equal the pair alone; %.0f pairs/s in this one call" % rate) if not bad else "MISMATCH at pairs %s..." % bad[:5]), flush=True)
    big.close(); one.close()
    del I0, I1, out
    torch.cuda.empty_cache()
    return not bad
ok = True
ok &= run(1920, 1080, 2, 600)
ok &= run(1920, 1080, 2, 1100)
ok &= run(3840, 2160, 4, 80)
ok &= run(1920, 1080, 2, 1200, u8=True)
ok &= run(1920, 1080, 2, 400, noc=3)
if "--huge" in sys.argv:                      # 70 GB of frames: element indices beyond 2^33
    ok &= run(1920, 1080, 2, 4200)
sys.exit(0 if ok else 1)
