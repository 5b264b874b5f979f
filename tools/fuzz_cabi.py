#!/usr/bin/env python3
"""fuzz of the C-ABI's argument checks: every exported function (the table in flowonthego_amd/_lib.py) is called with a null handle
and then, where it takes one, with a VALID handle of its kind and every other argument drawn from {null, 0, -1, 1, huge}: the call
must return (a status code), never crash.  Device pointers are only ever null here -- a wrong non-null pointer cannot be checked.
usage: python tools/fuzz_cabi.py [calls per function] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flowonthego_amd as F
from flowonthego_amd._lib import SYMBOLS, FotgParams, lib
per = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = lib()
vp = C.c_void_p
p = FotgParams()
assert L.fotg_op_point(2, 640, 1, p) == 0
ctx, pipe, node = vp(), vp(), vp()
assert L.fotg_create(p, 320, 200, 0, 2, C.byref(ctx)) == 0
assert L.fotg_pipe_create(p, 320, 200, 0, 2, 2, C.byref(pipe)) == 0
devs = (C.c_int * 2)(0, 0)
assert L.fotg_node_create(p, 320, 200, devs, 2, 2, 2, C.byref(node)) == 0
SKIP = {"fotg_destroy", "fotg_pipe_destroy", "fotg_node_destroy", "fotg_strerror", "fotg_version", "fotg_last_hip_error", "fotg_debug_counter"}
ints = [0, -1, 1, 2, 7, 64, 1 << 20, -(1 << 31), (1 << 31) - 1]
calls = 0
keep = []
def handle_for(name):
    return pipe if name.startswith("fotg_pipe_") else node if name.startswith("fotg_node_") and name not in ("fotg_node_create", "fotg_node_shard") else ctx
def arg(t, first_handle, name):
    if t is vp:
        return vp(None)
    if t in (C.c_int, C.c_long):
        return t(int(rng.choice(ints)))
    if t is C.c_char_p:
        return C.c_char_p(rng.choice([b"", b"a11", b"nonsense", b"stamps_ptr"]))
    if hasattr(t, "_type_"):                      # POINTER(x): null, or a small scratch object
        if rng.random() < 0.5:
            return t()
        obj = (t._type_ * 8)()
        keep.append(obj)
        return C.cast(obj, t)
    return t()
for name, res, args in SYMBOLS:
    if name in SKIP or name in ("fotg_create", "fotg_pipe_create", "fotg_node_create"):
        continue
    fn = getattr(L, name)
    takes_handle = bool(args) and args[0] is vp and name not in ("fotg_gradient_magnitude", "fotg_gradient_magnitude_u8")
    for k in range(per):
        a = [arg(t, False, name) for t in args]
        if takes_handle and k >= per // 4:
            a[0] = handle_for(name)               # a valid handle, everything else odd
        if os.environ.get("FUZZ_VERBOSE"):
            print(name, k, [getattr(x, "value", "ptr") for x in a], flush=True)
        fn(*a)
        calls += 1
# creation with odd arguments (null parameter block, null out pointer, odd sizes / devices / depths)
for k in range(3 * per):
    pp = C.byref(p) if rng.random() < 0.7 else None
    out = vp()
    o = C.byref(out) if rng.random() < 0.8 else None
    i = lambda: int(rng.choice(ints))
    which = k % 3
    st = (L.fotg_create(pp, i(), i(), int(rng.choice([0, 0, -1, 9])), i(), o) if which == 0 else
          L.fotg_pipe_create(pp, i(), i(), int(rng.choice([0, 0, -1, 9])), i(), i(), o) if which == 1 else
          L.fotg_node_create(pp, i(), i(), devs if rng.random() < 0.7 else None, int(rng.choice([0, 1, 2, -1, 99])), i(), i(), o))
    if st == 0 and o is not None and out.value:
        (L.fotg_destroy if which == 0 else L.fotg_pipe_destroy if which == 1 else L.fotg_node_destroy)(out)
    calls += 1
# the handles still work
f0 = (torch.rand((2, 200, 320), device="cuda") * 255).floor(); f1 = torch.roll(f0, 2, 2)
ow, oh = C.c_int(), C.c_int()
assert L.fotg_out_size(ctx, C.byref(ow), C.byref(oh)) == 0
out = torch.empty((2, oh.value, ow.value, 2), device="cuda")
torch.cuda.synchronize()
assert L.fotg_calc_batch(ctx, 2, vp(f0.data_ptr()), vp(f1.data_ptr()), None, vp(out.data_ptr()), None) == 0
torch.cuda.synchronize()
assert bool(torch.isfinite(out).all())
for h, d in ((None, L.fotg_destroy), (None, L.fotg_pipe_destroy), (None, L.fotg_node_destroy)):
    d(vp(None))                                   # destroying a null handle is a no-op
L.fotg_destroy(ctx); L.fotg_pipe_destroy(pipe); L.fotg_node_destroy(node)
print("fuzz_cabi: %d calls with null / odd arguments over %d functions, every one returned; the handles still computed afterwards" % (calls, len(SYMBOLS)))
