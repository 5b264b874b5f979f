import sys, os, ctypes as C
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch, bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd._lib import check, lib
op = F.operating_point(2, 1920, 1)
ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=64)
out = torch.randn((64, 68, 120, 2), device="cuda"); full = torch.empty((64, 1080, 1920, 2), device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = bench.HipEvents()
for _ in range(3):
    ms = ev.time_ms(lambda: check(lib().fotg_upsample_crop(ofc._h, 64, C.c_void_p(out.data_ptr()), C.c_void_p(full.data_ptr()), sp)), sp, 20)
    print("upsample_crop batch 64: %.4f ms = %.2f TB/s" % (ms, 64 * 1920 * 1080 * 8 / ms / 1e9))
