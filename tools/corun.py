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
def sor(st): check(lib().fotg_bench_sor_call(Bc._h, 4, B, st))
def sor5(st):
    for _ in range(5): sor(st)
def fused5(st):
    i0, s0 = Bc.level_ptr(0, 5, 0); i1, _ = Bc.level_ptr(1, 5, 0)
    check(lib().fotg_varref(Bc._h, 5, B, C.c_void_p(i0), C.c_void_p(i1), s0, p(fl5), st))
fl5 = torch.empty((B, 34, 60, 2), device=dev)
ts = run([(sor5, sC)])
tas = run([(pyr, sA), (sor5, sC)])
tbs = run([(lk, sB), (sor5, sC)])
tf = run([(fused5, sC)])
taf = run([(pyr, sA), (fused5, sC)])
tbf = run([(lk, sB), (fused5, sC)])
print("five sor_coupled calls of level 4 (the chain kernel only) alone %.1f us; with the pyramid %.1f (sum %.1f); with LK[4] %.1f (sum %.1f)" % (ts, tas, ta + ts, tbs, tb + ts))
print("fused level 5 alone %.1f us; with the pyramid %.1f (sum %.1f); with LK[4] %.1f (sum %.1f)" % (tf, taf, ta + tf, tbf, tb + tf))
def run_timed(fa, sa_, fb, sb_, n=60):
    """both streams busy back to back; per-stream mean duration of one call from events on that stream"""
    ext = lambda st: torch.cuda.ExternalStream(st.value)
    ea = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    eb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for k in range(n):
        ea[k][0].record(ext(sa_)); fa(sa_); ea[k][1].record(ext(sa_))
        eb[k][0].record(ext(sb_)); fb(sb_); eb[k][1].record(ext(sb_))
    torch.cuda.synchronize()
    da = sorted(a.elapsed_time(b) for a, b in ea[10:-10]); db = sorted(a.elapsed_time(b) for a, b in eb[10:-10])
    span = ea[0][0].elapsed_time(ea[-1][1]), eb[0][0].elapsed_time(eb[-1][1])
    print("   stream A: mean %.1f us, p10 %.1f, p90 %.1f, first-to-last %.1f us per call | stream B: mean %.1f, p10 %.1f, p90 %.1f, first-to-last %.1f us per call" % (
        sum(da) / len(da) * 1e3, da[len(da) // 10] * 1e3, da[-len(da) // 10] * 1e3, span[0] / n * 1e3, sum(db) / len(db) * 1e3, db[len(db) // 10] * 1e3, db[-len(db) // 10] * 1e3, span[1] / n * 1e3))
    return da[len(da) // 2] * 1e3, db[len(db) // 2] * 1e3
pa, sb5 = run_timed(pyr, sA, sor5, sC)
print("both streams busy: median pyramid launch %.1f us (alone %.1f), median five-sor group %.1f us (alone %.1f)" % (pa, ta, sb5, ts))
pa, fb5 = run_timed(pyr, sA, fused5, sC)
print("both streams busy: median pyramid launch %.1f us (alone %.1f), median fused level 5 %.1f us (alone %.1f)" % (pa, ta, fb5, tf))
pa, lb = run_timed(pyr, sA, lk, sB)
print("both streams busy: median pyramid launch %.1f us (alone %.1f), median LK[4] %.1f us (alone %.1f)" % (pa, ta, lb, tb))
print("refinement[4] alone %.1f us; with the pyramid %.1f us (sum %.1f); with LK[4] %.1f us (sum %.1f)" % (tv, tav, ta + tv, tbv, tb + tv))
