#!/usr/bin/env python3
"""Do an HBM-bound and a VALU-bound kernel of two different contexts run at the same time?  The pyramid of context A and the
level-4 LK of context B, each alone and both on streams of their own: python tools/corun.py"""
import os, sys, time, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import lib, check
B = 64
dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
def mkstream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s
op = F.operating_point(2, bench.W, 1)
ip = F.img_params(width=bench.W, height=bench.H, padding=op.patch_size)
I0, I1 = bench.synth_batch(B, 1, dev)
sA, sB, sC = mkstream(), mkstream(), mkstream()
A, Bc = OFClass(op, ip, max_batch=B), OFClass(op, ip, max_batch=B)
A.calc_batch(I0, I1); Bc.calc_batch(I0, I1); torch.cuda.synchronize()
p = lambda t: C.c_void_p(t.data_ptr())
def pyr(st): check(lib().fotg_pyramid_pair(A._h, B, p(I0), p(I1), 1, st))
def lk(st): check(lib().fotg_grid_optimize(Bc._h, 4, B, st))
def data(st):
    i0, s0 = Bc.level_ptr(0, 4, 0); i1, _ = Bc.level_ptr(1, 4, 0)
    check(lib().fotg_varref(Bc._h, 4, B, C.c_void_p(i0), C.c_void_p(i1), s0, p(fl), st))
fl = torch.empty((B, 68, 120, 2), device=dev)
def run(fns, n=100):
    for f, st in fns: f(st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        for f, st in fns: f(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
ta, tb = run([(pyr, sA)]), run([(lk, sB)])
tab = run([(pyr, sA), (lk, sB)])
tv = run([(data, sC)])
tav = run([(pyr, sA), (data, sC)])
tbv = run([(lk, sB), (data, sC)])
print("pyramid alone %.1f us, LK[4] alone %.1f us, both on two streams %.1f us per round (sum %.1f, max %.1f)" % (ta, tb, tab, ta + tb, max(ta, tb)))
print("refinement[4] alone %.1f us; with the pyramid %.1f us (sum %.1f); with LK[4] %.1f us (sum %.1f)" % (tv, tav, ta + tv, tbv, tb + tv))
