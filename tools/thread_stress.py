#!/usr/bin/env python3
"""host-thread stress: T threads, each creating / using / destroying its OWN contexts and pipes of different configurations at the
same time (the library's process-wide state: launch-attribute caches, counters, the last-HIP-error slot), every flow compared with
the result the main thread computed beforehand.  usage: python tools/thread_stress.py [threads] [rounds per thread]"""
import os, sys, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd.pipeline import FlowPipeline
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfgs = []
for k in range(8):
    noc = 3 if k % 4 == 3 else 1
    w, h = 320 + 37 * k, 200 + 23 * k
    opp = 1 + k % 3
    a, b = synth_pair(h, w, seed=k, noc=noc)
    a, b = torch.from_numpy(a).cuda()[None], torch.from_numpy(b).cuda()[None]
    op = F.operating_point(opp, w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 8)
    op.fast_math = (k % 5 == 4)
    if k % 4 == 2: op.finest_scale = max(0, op.finest_scale - 2)
    ip = F.img_params(width=w, height=h, padding=op.patch_size)
    o = OFClass(op, ip)
    ref = o.calc_batch(a, b).clone(); torch.cuda.synchronize(); o.close()
    cfgs.append((op, ip, a, b, ref))
errors = []
def worker(t):
    try:
        rng = np.random.default_rng(t)
        for r in range(R):
            op, ip, a, b, ref = cfgs[int(rng.integers(0, len(cfgs)))]
            if rng.random() < 0.5:
                o = OFClass(op, ip)
                for _ in range(int(rng.integers(1, 4))):
                    out = o.calc_batch(a, b)
                torch.cuda.synchronize()
                ok = torch.equal(out, ref)
                o.close()
            else:
                p = FlowPipeline(op, ip, max_batch=1, depth=int(rng.integers(1, 4)))
                outs = [p.submit(a, b)[1] for _ in range(int(rng.integers(1, 5)))]
                p.synchronize()
                ok = all(torch.equal(x, ref) for x in outs)
                p.close()
            if not ok:
                errors.append((t, r, "mismatch"))
    except Exception as e:
        errors.append((t, repr(e)))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for th in ths: th.start()
for th in ths: th.join()
print("thread stress: %d threads x %d rounds of create / compute / destroy, %d problems %s" % (T, R, len(errors), errors[:3]))
sys.exit(0 if not errors else 1)
