import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
dev = torch.device("cuda", 0)
op = F.operating_point(2, bench.W, 3)
B = 64
ofc = OFClass(op, F.img_params(width=bench.W, height=bench.H, padding=op.patch_size), max_batch=B)
I0, I1 = bench.synth_batch(B, 5, dev)
I0 = torch.stack([I0, I0.roll(3, 2), I0.roll(5, 1)], -1).contiguous(); I1 = torch.stack([I1, I1.roll(3, 2), I1.roll(5, 1)], -1).contiguous()
out = ofc.new_outflow(B)
for _ in range(3): ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ofc.calc_batch(I0, I1, None, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print("RGB 1080p op-pt 2 batch 64: %.3f ms/step, %.0f pairs/s" % (dt * 1e3, B / dt))
import ctypes as C
st = bench.stage_breakdown(ofc, I0, I1, out, F.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
print({k: round(v, 4) for k, v in st.items()})
