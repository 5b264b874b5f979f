import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
f0, f1 = synth_pair(2160, 3840, seed=5)
for opp in (4, 2):
    op = F.operating_point(opp, 3840, 1)
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=op.patch_size))
    a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
    for _ in range(2): ofc.calc(a, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ofc.calc(a, b)
    torch.cuda.synchronize()
    print("4K op-pt", opp, "scales", op.coarsest_scale, op.finest_scale, "ms per pair:", (time.perf_counter() - t) / 5 * 1e3)
    ofc.close()
# BASELINE configs[1]: one 1080p pair, op-pt 2 parameters, variational refinement off
g0, g1 = synth_pair(1080, 1920, seed=6)
for refine in (False, True):
    op = F.operating_point(2, 1920, 1)
    op.use_var_ref = refine
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=op.patch_size))
    a, b = torch.from_numpy(g0).cuda(), torch.from_numpy(g1).cuda()
    for _ in range(3): ofc.calc(a, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ofc.calc(a, b)
    torch.cuda.synchronize()
    print("1080p op-pt 2, refinement", refine, "ms per pair:", (time.perf_counter() - t) / 20 * 1e3)
    ofc.close()
