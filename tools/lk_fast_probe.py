#!/usr/bin/env python3
"""fast_math (csrc/lk_fast.hip.h) against the parity mode: per-stage LK times and endpoint error of the final flow.
python tools/lk_fast_probe.py [4k] [1080p] [alley]"""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch
import bench
from conftest import synth_pair
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass

lib = F.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
what = sys.argv[1:] or ["4k", "1080p", "alley"]


def epe_stats(a, b):
    e = torch.sqrt(((a - b) ** 2).sum(-1)).flatten()
    return {"mean": float(e.mean()), "p99": float(torch.quantile(e[:: max(1, e.numel() // 4000000)], 0.99)), "max": float(e.max())}


def run(name, w, h, oppt, n, f0, f1):
    res = {}
    flows = {}
    for fast in (False, True):
        op = F.operating_point(oppt, w, 1)
        op.fast_math = fast
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        out = ofc.new_outflow(n)
        st = bench.stage_breakdown(ofc, f0, f1, out, lib, sp, reps=3)
        ms = bench.timed(lambda: ofc.calc_batch(f0, f1, None, out), torch.cuda.synchronize, 5) * 1e3
        flows[fast] = ofc.upsample_crop(ofc.calc_batch(f0, f1)).clone()
        res["fast" if fast else "exact"] = {"ms_per_step": ms, "lk_ms": {k: round(v, 4) for k, v in st.items() if k.startswith("lk[")},
                                            "lk_total": sum(v for k, v in st.items() if k.startswith("lk[")), "all_stages": sum(st.values())}
        ofc.close()
    res["epe_fast_vs_exact_fullres"] = epe_stats(flows[True], flows[False])
    print(name, json.dumps(res, indent=1))


if "4k" in what:
    f0, f1 = synth_pair(2160, 3840, seed=5)
    run("4k_op4", 3840, 2160, 4, 1, torch.from_numpy(f0).cuda()[None], torch.from_numpy(f1).cuda()[None])
if "1080p" in what:
    bench.H, bench.W = 1080, 1920
    I0, I1 = bench.synth_batch(64, 1234, torch.device("cuda"))
    run("1080p_op2_b64", 1920, 1080, 2, 64, I0, I1)
if "alley" in what:
    z = np.load(os.path.join(ROOT, "tests", "golden", "alley_1_gray.npz"))
    a, b = z["frame_0001"].astype(np.float32), z["frame_0002"].astype(np.float32)
    run("alley_op2", a.shape[1], a.shape[0], 2, 1, torch.from_numpy(a).cuda()[None], torch.from_numpy(b).cuda()[None])
