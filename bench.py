#!/usr/bin/env python3
"""bench.py -- 1080p frame-pairs/s at DIS operating point 2 (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the whole hot path (pyramid+gradients -> per-scale LK -> densify -> variational refinement)
over one batch of synthetic 1920x1080 frame pairs already resident in HBM (BASELINE.json configs[2]: batch 64,
op-pt 2 + refinement on).  Frame pairs are independent, so N ranks each process their own batch (weak scaling, no
data-path collective); value = pairs all ranks processed / max-over-ranks time.

Batches in flight: the path is latency-bound at batch 64 (the refinement keeps a quarter of the CUs busy), so consecutive
steps -- each a complete batch -- are submitted to --in-flight engine contexts on internal streams in turn
(flowonthego_amd.FlowPipeline = fotg_pipe_* of include/fotg.h) and overlap on the GPU; every window still holds exactly K
complete steps between its barriers (the window ends with a host wait for all contexts).  The same window with one batch at
a time through a single context (fotg_calc_batch) is timed too and reported as `one_batch_at_a_time`.

Extra objects in the JSON line:
  roofline      HBM roofline of the kernel that moves the path's algorithmic bytes (pyr_base_kernel: every input byte
                exactly once), duration measured live with HIP events on the launch stream
  rooflines     the same for the three kernels that hold most of a step's time (pyr_base, the fused refinement level, one
                sor_coupled call of the finest level) + a compute-roof object for the LK kernel (useful flops / fp32 peak)
  stage_ms      per-stage GPU time of one step (each stage alone between HIP events); time_dominant_stage names the
                largest
  cpu_baseline  the CPU oracle (oracle/, a scalar port of the reference's kroeger/ path) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only); value = the flow-only all-core rate
  single_pair_no_refine, config_4k_op4   BASELINE configs[1] and configs[3], each with its own roofline object(s)
  rgb_frames    the same batch as 3-channel frames (the layout the reference's src/ path feeds) + the roofline of its pyramid kernel
  redblack      the same batch with red-black SOR (value + distance from the lexicographic result)
  sequence_mode, u8_frames   the video entry point and 8-bit input
  rccl_ranks, rank_placement   how many ranks RCCL connected and which GPU (PCI bus id) each one used

--gpus N > 1 without a launcher: this process starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a
child (before touching the GPU) and relays its output and exit code; under a launcher every rank checks WORLD_SIZE == N.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP deals its streams to GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included); two busy streams on one
# queue run one after the other.  Four batches in flight need a queue each (measured: 134 k pairs/s with 4 queues, 165 k with 8).
# Read by the runtime when it is loaded, i.e. before torch is imported; an explicit setting of the caller wins.
# (16, not the 5 the headline's four slots need: HIP deals ALL streams a process creates to the queues in turn, so the pipes of the
# later legs -- node API, 4K pairs in flight -- would otherwise land two busy streams on one queue: 139 k instead of 183 k pairs/s
# for a pipe created after another one, tools/two_pipes.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, OP_POINT = 1920, 1080, 2
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak
def latest_profile(suffix):
    """newest profiles/rNN_<suffix> that exists (None if there is none)"""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for n in range(9, 0, -1):
        if os.path.exists(os.path.join(d, "r%02d_%s" % (n, suffix))):
            return "r%02d_%s" % (n, suffix)
    return None


TRAFFIC_FILE = latest_profile("pmc_traffic.json") or "r02_pmc_traffic.json"
TRAFFIC_FILE_4K = latest_profile("4k_pmc_traffic.json")
TRAFFIC_FILE_4K_TILES = latest_profile("4k_tiles_pmc_traffic.json")      # the same passes with FOTG_VR_LEVELPIPE=0 (one vr_sor_tile_kernel launch per call)
TRAFFIC_NOTE = ("L2-MISS bytes per launch (requests that left an XCD's L2: Infinity-Cache hits are INCLUDED, so this is an upper bound of the HBM "
                "traffic) from profiles/%s: separate rocprofv3 --pmc passes of this command, bytes = 2 x FETCH_SIZE + WRITE_SIZE -- the x 2 is the guide's "
                "gfx950 correction, established for 16-byte-per-lane streaming reads (pyr_base_kernel); kernels that read with dword loads (lk, densify, "
                "vr_data) may be OVERSTATED by up to 2x on the read side.  NOT measured in this run" % TRAFFIC_FILE)


def synth_batch(n, seed, device):
    """n synthetic 1080p gray pairs on the device, values on the 8-bit grid: band-limited texture (6 octaves of
    bilinearly upsampled noise) and the same texture shifted by a smooth flow (global (5,2) px + +-2 px field)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.zeros((n, 1, H, W), device=device)
    for o, gs in enumerate((8, 16, 32, 64, 128, 256)):
        noise = torch.rand((n, 1, max(2, gs * H // W), gs), generator=g).to(device) * 2 - 1
        img += torch.nn.functional.interpolate(noise, size=(H, W), mode="bilinear", align_corners=True) / (o + 1)
    mn, mx = img.amin(dim=(2, 3), keepdim=True), img.amax(dim=(2, 3), keepdim=True)
    f0 = torch.round((img - mn) / (mx - mn) * 255.0)
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    u = 5.0 + 2.0 * torch.sin(yy / H * 3.0) * torch.cos(xx / W * 2.0)
    v = 2.0 + 2.0 * torch.cos(yy / H * 2.0 + 1.0) * torch.sin(xx / W * 3.0)
    gx = ((xx - u) / (W - 1)) * 2 - 1
    gy = ((yy - v) / (H - 1)) * 2 - 1
    grid = torch.stack([gx, gy], -1)[None].expand(n, -1, -1, -1)
    f1 = torch.round(torch.nn.functional.grid_sample(f0, grid, mode="bilinear", padding_mode="border", align_corners=True))
    return f0[:, 0].contiguous(), f1[:, 0].contiguous()


class HipEvents:
    """hipEvent timing on an explicit stream (torch.cuda.Event only sees torch's current stream)"""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def time_ms(self, fn, stream, reps):
        a, b = C.c_void_p(), C.c_void_p()
        self.hip.hipEventCreate(a); self.hip.hipEventCreate(b)
        fn()
        self.hip.hipEventRecord(a, stream)
        for _ in range(reps):
            fn()
        self.hip.hipEventRecord(b, stream)
        self.hip.hipEventSynchronize(b)
        ms = C.c_float()
        self.hip.hipEventElapsedTime(ms, a, b)
        return ms.value / reps

    def time_each_ms(self, fn, stream, reps):
        """mean over `reps` launches, EACH between its own pair of events: the events are barrier packets, so no launch overlaps the
        tail of the one before it -- the duration a profiler reports for the kernel (begin to end of one dispatch).  time_ms() above
        brackets a train of back-to-back launches instead: there the next dispatch starts filling CUs while the last workgroups of
        the previous one drain, so its per-launch figure is the launch INTERVAL, a few per cent shorter than the kernel duration
        (round 5: 154-158 us against rocprofv3's 164 us for pyr_base_kernel)."""
        evs = []
        for _ in range(2 * reps):
            e = C.c_void_p()
            self.hip.hipEventCreate(e)
            evs.append(e)
        fn()
        for i in range(reps):
            self.hip.hipEventRecord(evs[2 * i], stream)
            fn()
            self.hip.hipEventRecord(evs[2 * i + 1], stream)
        self.hip.hipEventSynchronize(evs[-1])
        tot = 0.0
        for i in range(reps):
            ms = C.c_float()
            self.hip.hipEventElapsedTime(ms, evs[2 * i], evs[2 * i + 1])
            tot += ms.value
        return tot / reps


def stage_breakdown(ofc, I0, I1, out, lib, stream_ptr, reps=5):
    """per-stage GPU time of one step, each stage launched alone between HIP events on the launch stream"""
    from flowonthego_amd._lib import check
    ev = HipEvents()
    op, n, h = ofc.op, I0.shape[0], ofc._h
    p = lambda t: C.c_void_p(t.data_ptr())
    st = {}
    st["pyramid(I0,I1)"] = ev.time_ms(lambda: check(lib.fotg_pyramid_pair(h, n, p(I0), p(I1), 3, stream_ptr)), stream_ptr, reps)
    check(lib.fotg_calc_batch(h, n, p(I0), p(I1), None, p(out), stream_ptr))      # leaves every level's state valid
    for sl in range(op.coarsest_scale, op.finest_scale - 1, -1):
        i0, s0 = ofc.level_ptr(0, sl, 0)
        i1, _ = ofc.level_ptr(1, sl, 0)
        ip = ofc.iparams[sl - op.finest_scale]
        st["lk[%d]" % sl] = ev.time_ms(lambda: check(lib.fotg_grid_optimize(h, sl, n, stream_ptr)), stream_ptr, reps)
        fl = torch.empty((n, ip.height, ip.width, 2), device=I0.device)
        st["densify[%d]" % sl] = ev.time_ms(lambda: check(lib.fotg_grid_aggregate(h, sl, n, p(fl), stream_ptr)), stream_ptr, reps)
        if op.use_var_ref:
            st["varref[%d]" % sl] = ev.time_ms(lambda: check(lib.fotg_varref(h, sl, n, C.c_void_p(i0), C.c_void_p(i1), s0, p(fl), stream_ptr)), stream_ptr, reps)
    return st


def roofline(ofc, I0, I1, lib, stream_ptr, batch):
    """HBM roofline of the kernel that moves the path's algorithmic bytes: pyr_base_kernel reads every input frame
    exactly once (one launch = both frames of `batch` pairs) and writes level 4.  Algorithmic bytes per launch =
    SURVEY.md 8(d)'s 2*W*H*4 B per pair x batch (+ the level-4 pixels it writes).  Duration: HIP events around that
    kernel alone on the launch stream.  traffic: PMC-measured HBM bytes per launch from profiles/ (separate
    rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction), same workload."""
    from flowonthego_amd._lib import check
    ev = HipEvents()
    p = lambda t: C.c_void_p(t.data_ptr())
    launch = lambda: check(lib.fotg_pyramid_pair(ofc._h, batch, p(I0), p(I1), 1, stream_ptr))
    ms_train = ev.time_ms(launch, stream_ptr, 20)          # launch interval of a back-to-back train (what rounds 1-5 reported)
    ms = ev.time_each_ms(launch, stream_ptr, 20)           # duration of one launch, comparable with rocprofv3's kernel duration
    lw, lh = ofc.width >> 4, ofc.height >> 4
    alg = batch * (2 * W * H * 4 + 2 * lw * lh * 4)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            k = json.load(f)["kernels"]
        key = [x for x in k if "pyr_base_kernel<float, 1, 4, true" in x]
        if key and batch == 64:
            traffic = k[key[0]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    gbs = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "fotg::pyr_base_kernel<float,1,4,true,1> (frames -> pyramid level 4, both frames of the batch in one launch)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_semantics": "L2-miss bytes (Infinity-Cache hits included)",
            "traffic_source": TRAFFIC_NOTE, "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "ms_per_launch_back_to_back": ms_train,
            "timing": "ms_per_launch: each launch between its own pair of HIP events on the launch stream (no overlap with its neighbours: the figure "
                      "rocprofv3 reports as the kernel's duration, profiles/); ms_per_launch_back_to_back: 20 launches in a train / 20 = the launch "
                      "interval, shorter because consecutive dispatches overlap head to tail; achieved / frac use ms_per_launch",
            "measured": "one launch at a time on an otherwise idle GPU; with %s batches in flight the same kernel stretches (profiles/)" % "several"}


def roofline_dominant(ofc, lib, stream_ptr, batch, stage_ms):
    """The kernel that holds most of the step's time: one sor_coupled call of the finest level (vr_sor_stream_kernel, launched
    once per inner iteration: 5 x at level 4).  Algorithmic bytes per launch = batch x w x h x (32 B system cell read + 8 B (du,dv)
    read + 8 B written); duration: HIP events around that launch alone on the launch stream.  It is a
    dependency chain (w + h + 15 lock-stepped anti-diagonals), so the HBM roof is the nearest of the two the contract names,
    not what bounds it -- `bound_by` says so."""
    from flowonthego_amd._lib import check
    ev = HipEvents()
    lvl = ofc.op.finest_scale
    lw, lh = ofc.width >> lvl, ofc.height >> lvl
    ms = ev.time_ms(lambda: check(lib.fotg_bench_sor_call(ofc._h, lvl, batch, stream_ptr)), stream_ptr, 20)
    alg = batch * lw * lh * 48
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            k = json.load(f)["kernels"]
        key = [x for x in k if "vr_sor_stream_kernel" in x]
        if key and batch == 64:
            traffic = k[key[0]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    inner = lvl + 1
    gbs = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "bound_note": "nearest roof the contract names; the kernel is latency-bound (bound_by) and its data is L2 resident",
            "kernel": "fotg::vr_sor_stream_kernel<72,70,4,32> (one sor_coupled call = 3 lexicographic sweeps of the %dx%d level, "
                      "one workgroup per pair; %d launches per step)" % (lw, lh, inner),
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_semantics": "L2-miss bytes (Infinity-Cache hits included)", "traffic_source": TRAFFIC_NOTE,
            "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "launches_per_step": inner,
            "share_of_step": inner * ms / sum(stage_ms.values()),
            "bound_by": "dependency-chain latency: %d anti-diagonal steps in lock step (~%.0f ns each), one workgroup = one CU per pair; "
                        "the data it touches is L2 resident" % (lw + lh - 1 + 16, ms * 1e6 / (lw + lh - 1 + 16))}


def roofline_fused_level(ofc, batch, stage_ms, lvl):
    """vr_inner_fused_kernel: the whole refinement of a small level (5, 6) in ONE launch, one workgroup per pair.  Algorithmic
    bytes per launch: per pair both padded level images once (2 (w+2ps)(h+2ps) 4 B), the flow in and
    out (2 w h 8 B); everything else stays on chip.  Duration: the level's varref stage time (that stage IS the one launch)."""
    lw, lh = ofc.width >> lvl, ofc.height >> lvl
    ps = ofc.op.patch_size
    alg = batch * (2 * (lw + 2 * ps) * (lh + 2 * ps) * 4 + 2 * lw * lh * 8)
    ms = stage_ms["varref[%d]" % lvl]
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            k = json.load(f)["kernels"]
        key = [x for x in k if "vr_inner_fused_kernel" in x and ("1024>" in x) == (lw * lh > 1024)]     # (levels of > 1024 px run on 1024 threads)
        if key and batch == 64:
            traffic = k[key[0]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    gbs = alg / (ms * 1e-3) / 1e9
    inner = lvl + 1
    return {"bound": "hbm", "bound_note": "nearest roof the contract names; the kernel is bound by the solver's dependency chain and the VALU of ONE CU per pair",
            "kernel": "fotg::vr_inner_fused_kernel<1,8,32,true,true,%d> (level %d, %dx%d: set-up + %d x {data term, 3 sweeps} + w+d in one launch, one workgroup per pair)" % (1024 if lw * lh > 1024 else 512, lvl, lw, lh, inner),
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_semantics": "L2-miss bytes (Infinity-Cache hits included)", "traffic_source": TRAFFIC_NOTE,
            "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "launches_per_step": 1, "share_of_step": ms / sum(stage_ms.values())}


def roofline_lk(ofc, batch, stage_ms):
    """compute roof of lk_kernel<8,1> at the finest level: USEFUL flops (per pixel of a patch and evaluation: 7 bilinear, 2 mean,
    1 residual, 4 for the two projections, 2 for the L1 residual = 16; per patch and iteration ~30 for the 2x2 solve and the
    tests, negligible) / launch time, against the fp32 vector peak.  The kernel is VALU-issue bound at ~65 instructions per
    pixel-iteration (scalar per-patch code carried on 16 lanes, IEEE divisions, no FMA by the numerics contract)."""
    op = ofc.op
    lvl = op.finest_scale
    lw, lh = ofc.width >> lvl, ofc.height >> lvl
    steps = max(1, int(op.patch_size * (1 - op.patch_stride)))
    nop = -(-lw // steps) * -(-lh // steps)
    evals = op.grad_descent_iter + 1
    flops = batch * nop * evals * op.patch_size * op.patch_size * 16
    ms = stage_ms["lk[%d]" % lvl]
    tf = flops / (ms * 1e-3) / 1e12
    lpp8 = -(-nop // 8) * batch >= 2048 and op.patch_size == 8           # the library's automatic rule (FOTG_LK_LPP = 0): eight lanes per patch
    return {"bound": "valu", "kernel": "fotg::lk_kernel<8,1,false,true,%s> (level %d: %d patches x %d evaluations x 64 px per pair)" % ("true,8" if lpp8 else "false,16", lvl, nop, evals),
            "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "useful_flops_per_launch": flops,
            "ms_per_launch": ms, "share_of_step": sum(v for k, v in stage_ms.items() if k.startswith("lk[")) / sum(stage_ms.values()),
            "note": "useful flops only; the instruction stream is ~3-4x that (profiles/r06_pmc_valu.json: 330 VALU wave-instructions per EIGHT-patch "
                    "iteration with eight lanes per patch, 234 per four-patch iteration with sixteen); the launch retires one VALU wave-instruction per ~4.3 cycles and SIMD, where the hardware issues plain f32 / int adds "
                    "at 2.3 and selects, compares, DPP, conversions, packed f32 at 4.1-4.4 (profiles/r03_valu_issue_probe.json): about three quarters "
                    "of the issue rate of its mix"}


def timed(fn, sync, steps, warm=2):
    """wall time per call of fn (enqueue `steps` calls, one synchronisation)"""
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def synth_frame_pair(h, w, seed, device):
    """one synthetic pair of any size (same generator as synth_batch)"""
    global H, W
    keep = (H, W)
    H, W = h, w
    try:
        f0, f1 = synth_batch(1, seed, device)
    finally:
        H, W = keep
    return f0, f1


def config_4k_op4(F, OFClass, lib, local, dev, stream_ptr, fast=False, ref_flow=None, sor_mode=0):
    """BASELINE configs[3]: ONE 3840x2160 pair at operating point 4 (ps 12, stride 3, scales 7..2, 128 LK iterations, refinement on
    every level up to 960x544): ms per pair one at a time and with four pairs in flight, and the rooflines of its two dominant
    kernels, each timed alone with HIP events on the launch stream."""
    from flowonthego_amd.pipeline import FlowPipeline
    w4, h4 = 3840, 2160
    op = F.operating_point(4, w4, 1, sor_mode=sor_mode)
    op.fast_math = bool(fast)
    ip = F.img_params(width=w4, height=h4, padding=op.patch_size)
    ofc = OFClass(op, ip, max_batch=1, device=local)
    f0, f1 = synth_frame_pair(h4, w4, 77, dev)
    out = ofc.new_outflow(1)
    sync = torch.cuda.synchronize
    ms1 = timed(lambda: ofc.calc_batch(f0, f1, None, out), sync, 5) * 1e3
    st = stage_breakdown(ofc, f0, f1, out, lib, stream_ptr, reps=3)
    res = {"workload": "BASELINE configs[3]: one 3840x2160 gray f32 pair, DIS op-pt 4 (ps 12, stride 3, scales %d..%d, %d LK iterations, refinement, "
                       "%s SOR); inputs resident in HBM, output = finest-scale flow 960x544x2" % (op.coarsest_scale, op.finest_scale, op.grad_descent_iter,
                        "lexicographic" if sor_mode == 0 else "RED-BLACK (the ordering of the reference's CUDA path, src/kernels/flowUtil.cu:297-362; not the parity mode)"),
           "ms_per_pair": ms1, "value": 1e3 / ms1, "unit": "frame-pairs/s",
           "stage_ms": {k: round(v, 4) for k, v in st.items()},
           "reference_point": "19 ms on a GTX 1080 for the reference's CUDA build at this preset (/root/reference/docs/index.md:167-175; other hardware, "
                              "other numerics -- context only)"}
    # LK at the finest level: useful flops / fp32 vector peak
    lvl = op.finest_scale
    lw, lh = ofc.width >> lvl, ofc.height >> lvl
    steps = max(1, int(op.patch_size * (1 - op.patch_stride)))
    nop = -(-lw // steps) * -(-lh // steps)
    evals = op.grad_descent_iter + 1
    flops = nop * evals * op.patch_size * op.patch_size * 16
    ms = st["lk[%d]" % lvl]
    tf = flops / (ms * 1e-3) / 1e12
    lk_all = sum(v for k, v in st.items() if k.startswith("lk["))
    res["rooflines"] = [{"bound": "valu", "kernel": "fotg::%s (level %d: %d patches x %d evaluations x 144 px)" % ("lk_fast_kernel<12,1,16>" if fast else "lk_kernel<12,1,false,true,false>", lvl, nop, evals),
                         "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "useful_flops_per_launch": flops,
                         "ms_per_launch": ms, "share_of_pair": lk_all / sum(st.values()),
                         "note": ("useful flops only (16 per pixel-evaluation); tolerance mode: 91 VALU wave-instructions per four-patch iteration "
                                  "(profiles/r06_pmc_valu.json), VALU busy 87 %") if fast else
                                 ("useful flops only (16 per pixel-evaluation); the kernel is bound by the issue rate of its instruction stream (263 VALU "
                                  "wave-instructions per four-patch iteration, profiles/r06_pmc_valu.json, retired at one per ~4.8 cycles and SIMD; five instead of "
                                  "three waves per SIMD do not speed it up: docs/EXPERIMENTS.md)")}]
    # one sor_coupled call of the finest level through the tile pipeline
    try:
        if sor_mode != 0:
            raise RuntimeError("red-black: no tile pipeline")
        ev = HipEvents()
        from flowonthego_amd._lib import check
        mss = ev.time_ms(lambda: check(lib.fotg_bench_sor_call(ofc._h, lvl, 1, stream_ptr)), stream_ptr, 5)
        alg = lw * lh * 48 * op.var_ref_iter
        gbs = alg / (mss * 1e-3) / 1e9
        inner = lvl + 1
        traffic4k, src4k = None, None
        try:
            tk = {}
            for tf in (TRAFFIC_FILE_4K, TRAFFIC_FILE_4K_TILES):
                if tf:
                    with open(os.path.join(ROOT, "profiles", tf)) as f:
                        tk.update(json.load(f)["kernels"])
            cand = [v["hbm_bytes_per_launch_corrected"] for k, v in tk.items() if "vr_sor_tile_kernel" in k]
            if cand:
                traffic4k = int(max(cand))                       # (the largest tile launch = the finest level)
                src4k = ("L2-miss bytes per launch (Infinity-Cache hits included) from profiles/%s (separate rocprofv3 --pmc passes of "
                         "tools/time_4k_op4.py with FOTG_VR_LEVELPIPE=0; bytes = 2 x FETCH_SIZE + WRITE_SIZE) -- NOT measured in this run" % (TRAFFIC_FILE_4K_TILES or TRAFFIC_FILE_4K))
        except Exception:
            pass
        res["rooflines"].append({"bound": "hbm", "bound_note": "nearest roof the contract names; the kernel is a pipeline of dependency chains (bound_by)",
                                 "kernel": "fotg::vr_sor_tile_kernel<8> (one sor_coupled call = %d lexicographic sweeps of the %dx%d level as tiles of 64 rows "
                                           "x one sweep on %d workgroups; %d launches per pair at this level)" % (op.var_ref_iter, lw, lh, -(-lh // 64) * op.var_ref_iter, inner),
                                 "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic4k, "traffic_semantics": "L2-miss bytes (Infinity-Cache hits included)", "traffic_source": src4k,
                                 "algorithmic_bytes_per_launch": alg, "ms_per_launch": mss, "launches_per_pair": inner,
                                 "share_of_pair": sum(v for k, v in st.items() if k.startswith("varref[")) / sum(st.values()),
                                 "bound_by": "%d anti-diagonal steps of one wave per tile (~%.0f ns each) + the pipeline lag between tiles" %
                                             (lw + lh - 1, mss * 1e6 / (lw + lh - 1))})
    except Exception as e:
        res["rooflines"].append({"unavailable": str(e)})
    # the level pipeline (the default path of the tall levels: every inner iteration of a level -- data terms and sor_coupled calls --
    # as one launch): algorithmic bytes of the finest level = inner x (3 sweeps x 48 B per cell) + (inner - 1) x 84 B per cell for the
    # data terms (11 planes, (du,dv), the 32-byte system cell), over the stage time of that level (set-up and final w + d included)
    try:
        if sor_mode != 0:
            raise RuntimeError("red-black: no level pipeline")
        inner = lvl + 1
        alg_lp = lw * lh * (inner * op.var_ref_iter * 48 + (inner - 1) * 84)
        ms_lp = st["varref[%d]" % lvl]
        traffic_lp = None
        try:
            with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE_4K)) as f:
                tk = json.load(f)["kernels"]
            cand = [v["hbm_bytes_per_launch_corrected"] for k, v in tk.items() if "vr_level_pipe_kernel" in k]
            if cand and not fast:
                traffic_lp = int(max(cand))                      # (the largest level-pipe launch = the finest level)
        except Exception:
            pass
        res["rooflines"].append({"bound": "hbm", "bound_note": "nearest roof the contract names; the launch is a pipeline of dependency chains",
                                 "kernel": "fotg::vr_level_pipe_kernel (level %d, %dx%d: %d inner iterations x %d sweeps + %d data terms as one launch; FOTG_VR_LEVELPIPE=0 "
                                           "runs one vr_sor_tile_kernel launch per call instead)" % (lvl, lw, lh, inner, op.var_ref_iter, inner - 1),
                                 "achieved": alg_lp / (ms_lp * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_lp / (ms_lp * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": traffic_lp, "traffic_semantics": "L2-miss bytes per launch (Infinity-Cache hits included), profiles/%s, not measured in this run" % TRAFFIC_FILE_4K,
                                 "algorithmic_bytes_per_launch": alg_lp, "ms_per_launch": ms_lp,
                                 "level_pipe_launches_per_process": int(lib.fotg_debug_counter(b"level_pipe")),
                                 "bound_by": "S = %d anti-diagonal steps of the first call + the lag of %d pipeline stages behind it" % (lw + lh - 1, inner * op.var_ref_iter + inner - 1)})
    except Exception as e:
        res["rooflines"].append({"unavailable": str(e)})
    # four pairs in flight (a 4K video: consecutive pairs, one per submit)
    D = 4
    pipe = FlowPipeline(op, ip, max_batch=1, depth=D, device=local)
    outs = [pipe.new_outflow(1) for _ in range(D)]
    sync()
    k = [0]

    def sub():
        pipe.submit(f0, f1, None, outs[k[0] % D], after_current_stream=False)
        k[0] += 1
    msD = timed(sub, pipe.synchronize, 8 * D, warm=2 * D) * 1e3          # (the window starts with an empty pipe and ends with a drain)
    one = ofc.calc_batch(f0, f1).clone()
    res["fast_math"] = bool(fast)
    if ref_flow is not None:
        # endpoint error of this mode's full-resolution flow against the parity mode's (which is == the oracle: tests)
        e = torch.sqrt(((ofc.upsample_crop(one) - ref_flow) ** 2).sum(-1)).flatten()
        res["epe_vs_parity_mode_px"] = {"mean": float(e.mean()), "p99": float(torch.quantile(e[::4], 0.99)), "max": float(e.max()),
                                        "note": "full-resolution flow (3840x2160) of this mode against the parity mode's, which is bit-identical to the CPU oracle"}
    else:
        res["_full_flow"] = ofc.upsample_crop(one).clone()
    res["in_flight"] = {"batches_in_flight": D, "ms_per_pair": msD, "value": 1e3 / msD, "unit": "frame-pairs/s",
                        "same_bits_as_one_at_a_time": bool(torch.equal(outs[0], one))}
    pipe.close(); ofc.close()
    # throughput of a 4K stream: four pairs per submit, four submits in flight (16 pairs resident)
    try:
        B4 = 4
        pipe = FlowPipeline(op, ip, max_batch=B4, depth=D, device=local)
        g0, g1 = f0.expand(B4, -1, -1).contiguous(), f1.expand(B4, -1, -1).contiguous()
        outs = [pipe.new_outflow(B4) for _ in range(D)]
        sync()
        k[0] = 0

        def sub4():
            pipe.submit(g0, g1, None, outs[k[0] % D], after_current_stream=False)
            k[0] += 1
        ms4 = timed(sub4, pipe.synchronize, 4 * D, warm=D) * 1e3 / B4
        res["throughput"] = {"pairs_per_submit": B4, "submits_in_flight": D, "ms_per_pair": ms4, "value": 1e3 / ms4, "unit": "frame-pairs/s",
                             "same_bits_as_one_at_a_time": bool(all(torch.equal(outs[0][j], one[0]) for j in range(B4)))}
        pipe.close()
    except Exception as e:
        res["throughput"] = {"unavailable": str(e)}
    return res


def cpu_baseline(I0, I1, budget_s=12.0):
    """the oracle (scalar C port of the reference's kroeger/ path) on the host: first single-threaded for a few pairs, then
    frame-parallel over all host cores (one pair per thread -- the CPU analogue of frame sharding) on a bounded
    sample of the same batch.  `value` is the all-cores FLOW-ONLY rate (the quantity the reference itself times); the rate with the
    port's scalar pyramid included is reported beside it."""
    from oracle import oracle as O
    flags = O.use_native()                # -O3 -msse4 -march=native of THIS host (the travelling library has no -march)
    p = O.op_point(OP_POINT, W, 1)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota) is what the process can really use: more threads than that only
    # get throttled (the GPU boxes of this pool show 256 CPUs and a quota of 16)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except Exception:
            pass
    visible = cores
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    nb = I0.shape[0]
    host = {}

    def pair(k):
        k %= nb
        if k not in host:
            host[k] = (I0[k].cpu().numpy(), I1[k].cpu().numpy())
        return host[k]

    def one(k):
        a, b = pair(k)
        O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0)      # ctypes drops the GIL inside the C call

    t0 = time.perf_counter()
    n1 = 0
    while n1 < 3 or (time.perf_counter() - t0) < 0.2 * budget_s:
        one(n1)
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    # per-stage split of ONE pair on one thread, the oracle's stage entry points in the order of kroeger/oflow.cpp:184-337 (the
    # reference prints the same split as "TIME (Sc: ..)", oflow.cpp:303); ms per pair, comparable with stage_ms / batch
    # (three passes over the same pair, the fastest time of every stage: a single pass measured the first touch of the stage objects'
    # memory -- 50 ms instead of 7.7 for the flow of one pair on one of the driver's boxes)
    st = {}
    npatch = {}
    a0, b0 = pair(0)
    pa, pb = O.pad_frame(a0, p.sc_f), O.pad_frame(b0, p.sc_f)
    for _pass in range(3):
        cur = {}

        def lap(name, t):
            cur[name] = cur.get(name, 0.0) + (time.perf_counter() - t) * 1e3

        t = time.perf_counter()
        for fr in (pa, pb):                                   # the C call alone (the Pyramid wrapper below also copies every level)
            O.lib().dis_pyramid_free(O.lib().dis_pyramid_build(O.P(fr), fr.shape[1], fr.shape[0], 1, p.sc_f, p.ps))
        lap("pyramid(I0,I1)", t)
        P0, P1 = O.Pyramid(pa, p.sc_f, p.ps), O.Pyramid(pb, p.sc_f, p.ps)
        prev = None
        for sl in range(p.sc_f, p.sc_l - 1, -1):
            lw, lh = P0.level_wh(sl)
            t = time.perf_counter()
            g = O.Grid(lw, lh, sl, p)
            npatch[sl] = g.nop
            g.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
            if prev is not None:
                g.init_from_coarser(prev)
            g.optimize(P1.im[sl])
            lap("lk[%d]" % sl, t)
            t = time.perf_counter()
            fl = g.aggregate()
            lap("densify[%d]" % sl, t)
            t = time.perf_counter()
            prev = O.varref(P0.im[sl], P1.im[sl], lw, lh, sl, p, fl)
            lap("varref[%d]" % sl, t)
        for k, v in cur.items():
            st[k] = min(st.get(k, v), v)
    # all cores: dis_flow_many -- pthreads inside the C library, one pair per thread at a time, thread-private block caches (no
    # allocation per pair after a thread's first), every thread runs one untimed pair first.  Threads are bounded by memory:
    # a worker holds both pyramids of its pair (~90 MB at 1080p).
    try:
        import psutil
        mem_threads = max(1, int(psutil.virtual_memory().available * 0.5 / 200e6))
    except Exception:
        mem_threads = cores
    threads = max(1, min(cores, mem_threads))
    nsrc = min(nb, 8)
    H0 = np.stack([pair(k)[0] for k in range(nsrc)]); H1 = np.stack([pair(k)[1] for k in range(nsrc)])
    # a short calibration run sizes the timed ones for ~0.3 * budget_s each
    sec, _ = O.flow_many(H0, H1, p, threads, threads, True)
    nall = max(threads, min(int(threads / max(sec, 1e-3) * 0.3 * budget_s), 64 * threads))
    el, _ = O.flow_many(H0, H1, p, nall, threads, True)
    sec, _ = O.flow_many(H0, H1, p, threads, threads, False)
    nflow = max(threads, min(int(threads / max(sec, 1e-3) * 0.3 * budget_s), 256 * threads))
    el_flow, _ = O.flow_many(H0, H1, p, nflow, threads, False)
    flow_ms = sum(v for k, v in st.items() if not k.startswith("pyramid"))
    # the reference's own per-scale timing line (kroeger/oflow.cpp:303), from the port's stage split
    time_lines = ["TIME (Sc: %d, #p:%6d, pconst, pinit, poptim, cflow, tvopt, total): %8.2f %8.2f %8.2f %8.2f %8.2f -> %8.2f ms."
                  % (sl, npatch[sl], 0.0, 0.0, st["lk[%d]" % sl], st["densify[%d]" % sl], st["varref[%d]" % sl],
                     st["lk[%d]" % sl] + st["densify[%d]" % sl] + st["varref[%d]" % sl]) for sl in range(p.sc_f, p.sc_l - 1, -1)]
    return {"value": nflow / el_flow, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "value_is": "flow only on all cores -- what the reference times and prints as O.Flow Run-Time (kroeger/oflow.cpp:355-360); the pyramids "
                        "are built once per thread outside the timed region",
            "all_cores_flow_only": nflow / el_flow,
            "all_cores_with_pyramid": nall / el,
            "with_pyramid_note": "the port's pyramid is scalar C where the reference calls OpenCV's SIMD resize / Sobel / copyMakeBorder "
                                 "(kroeger/run_dense.cpp:150-175): this figure under-states the reference's CPU path and is NOT the baseline value",
            "host_cpus_visible": visible, "cgroup_cpu_quota": quota,
            "scaling_vs_single_thread": {"with_pyramid": nall / el / single, "flow_only": nflow / el_flow / (1e3 / flow_ms)},
            "single_thread": single, "single_thread_flow_only": 1e3 / flow_ms,
            "single_thread_stage_ms_per_pair": {k: round(v, 3) for k, v in st.items()}, "time_lines": time_lines,
            "sample_short": "%d flow-only runs (LK + densify + refinement, 3 scales) of the batch's 1080p pairs, one per thread on %d threads, %.1f s" % (nflow, threads, el_flow),
            "sample": "`value`: %d runs of the flow (kroeger/oflow.cpp:184-337: LK, densification, refinement of the three scales) on pairs of the "
                      "batch (every thread keeps the pyramids of one pair, built outside the timed region like the reference's O.Flow Run-Time "
                      "excludes them, kroeger/oflow.cpp:355-360), op-pt 2 + refinement, one pair per thread on %d pthreads in %.1f s (dis_flow_many: "
                      "per-thread block caches, no allocation per pair), oracle/dis_oracle.c built on this host with %s; beside it: %d pairs with "
                      "padding + both pyramids in the loop (kroeger/run_dense.cpp:130-178, scalar in the port) in %.1f s; single thread: %d pairs at "
                      "%.1f pairs/s with the pyramid, %.1f pairs/s flow only; the survey's probe of the real kroeger build (Eigen, -O3 -msse4) "
                      "measured ~130 pairs/s/core flow only on a 2.1 GHz Xeon" % (nflow, threads, el_flow, flags, nall, el, n1, single, 1e3 / flow_ms)}


def guard(res, name, fn):
    """run one informational leg; a failure there must not cost the headline line (the leg's entry says what went wrong)"""
    try:
        fn()
    except Exception as e:
        import traceback
        traceback.print_exc(file=sys.stderr)
        res.setdefault(name, {})
        res[name]["unavailable"] = "%s: %s" % (type(e).__name__, str(e)[:160])


SHORT_LINE_MAX = 6000          # bytes; the driver keeps an 8 KB tail of stdout (round 5's 23 KB line could not be parsed)


def _r(x, n=4):
    """float -> n significant digits (keeps the line short); everything else unchanged"""
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else None
    return x


def _pick(d, keys, n=4):
    return {k: _r(d[k], n) for k in keys if isinstance(d, dict) and k in d}


def short_line(res):
    """The ONE line bench.py prints on stdout: the contract's keys, one roofline, the CPU baseline and one-number summaries of the
    other legs.  Everything else (stage tables, notes, the list of rooflines, the baseline's sample prose) is in bench_detail.json /
    on stderr.  tests/test_host.py::test_bench_line_is_short holds it under SHORT_LINE_MAX bytes, also for --gpus 8."""
    s = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data"), 6)
    cfg = res.get("config", {})
    s["config"] = {"workload": "BASELINE configs[2]: batch %s synthetic 1920x1080 gray f32 pairs per GPU resident in HBM, DIS op-pt 2 + variational refinement "
                               "(SOR order: %s), output = finest-scale flow 120x68x2" % (cfg.get("batch_per_gpu"), cfg.get("sor")),
                   "global_batch": cfg.get("global_batch"), "pairs_in_flight": cfg.get("pairs_in_flight"),
                   "batches_in_flight_per_gpu": res.get("batches_in_flight"),
                   "parallelism": "frame-pair sharding x%s, no collective" % res.get("n_gpus")}
    s["one_batch_at_a_time"] = _pick(res.get("one_batch_at_a_time", {}), ("value", "ms_per_step"), 6)
    if "roofline" in res:
        s["roofline"] = _pick(res["roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                "ms_per_launch", "ms_per_launch_back_to_back", "unavailable"), 5)
        if "kernel" in s["roofline"]:
            s["roofline"]["kernel"] = s["roofline"]["kernel"].split(" (")[0]
    if "whole_path_hbm_frac" in res:
        s["whole_path_hbm_frac"] = _r(res["whole_path_hbm_frac"])
    if "time_dominant_stage" in res:
        s["time_dominant_stage"] = res["time_dominant_stage"]
    cb = res.get("cpu_baseline")
    if cb:
        s["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_thread", "single_thread_flow_only", "unavailable"))
        if "sample" in cb:
            s["cpu_baseline"]["sample"] = cb.get("sample_short", cb["sample"][:120])
        if "parity" in cb:
            s["cpu_baseline"]["parity"] = _pick(cb["parity"], ("pairs_checked", "mean_epe_px", "bit_identical"))
    # one-number summaries of the other legs
    for k in ("fast_math", "u8_frames", "u8_bgr_gray", "rgb_frames"):
        if k in res:
            s[k] = _pick(res[k], ("value", "unavailable"))
            if isinstance(res[k].get("in_flight"), dict):
                s[k]["in_flight"] = _r(res[k]["in_flight"].get("value"))
            if "epe_vs_parity_mode_px" in res[k]:
                s[k]["mean_epe_px"] = _r(res[k]["epe_vs_parity_mode_px"].get("mean"), 3)
    for k in ("single_pair_no_refine", "full_resolution_output", "sequence_mode", "node_api", "depth_mode"):
        if k in res:
            s[k] = _pick(res[k], ("value", "ms_per_pair", "unavailable"))
    if "redblack" in res:
        s["redblack"] = _pick(res["redblack"], ("value", "mean_epe_px_vs_lexicographic", "unavailable"))
    for k in ("config_4k_op4", "config_4k_op4_fast_math", "config_4k_op4_redblack"):
        if k in res:
            d = res[k]
            s[k] = _pick(d, ("ms_per_pair", "unavailable"))
            if isinstance(d.get("throughput"), dict) and "value" in d["throughput"]:
                s[k]["throughput"] = _r(d["throughput"]["value"])
            if isinstance(d.get("in_flight"), dict) and "value" in d["in_flight"]:
                s[k]["in_flight"] = _r(d["in_flight"]["value"])
            if "epe_vs_parity_mode_px" in d:
                s[k]["mean_epe_px"] = _r(d["epe_vs_parity_mode_px"].get("mean"), 3)
            for rf in d.get("rooflines", []):
                if "vr_level_pipe" in rf.get("kernel", ""):
                    s[k]["level_pipe"] = _pick(rf, ("frac", "traffic", "algorithmic_bytes_per_launch", "ms_per_launch"))
                if "lk_" in rf.get("kernel", "") and rf.get("bound") == "valu":
                    s[k]["lk_valu_frac"] = _r(rf.get("frac"), 3)
    if res.get("rccl_ranks") is not None or res.get("n_gpus", 1) > 1:
        s["rccl_ranks"] = res.get("rccl_ranks")
    if res.get("shared_gpu_test"):
        s["shared_gpu_test"] = True
    if "ms_per_step_per_rank" in res:
        pr = res["ms_per_step_per_rank"]
        s["ms_per_step_per_rank"] = {"min": _r(min(pr)), "max": _r(max(pr))}
    pl = res.get("rank_placement") or []
    s["gpus_distinct"] = len({p.get("pci_bus_id") for p in pl})
    sg = res.get("scatter_gather")
    if sg:
        s["scatter_gather"] = _pick(sg, ("scatter_plus_compute_ms", "gather_ms", "end_to_end_pairs_per_s", "gathered_flows_match_single_context", "unavailable"))
    s["detail"] = res.get("detail_file", "bench_detail.json (+ stderr)")
    return s


def emit(res):
    """detail -> bench_detail.json (gpurun_out/ when it can be created, else the working directory) and stderr; the short line ->
    stdout, last"""
    detail = json.dumps(res)
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(detail + "\n")
            res["detail_file"] = os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
            break
        except OSError:
            continue
    print("bench.py detail: " + detail, file=sys.stderr)
    line = json.dumps(short_line(res), separators=(",", ":"))
    if len(line) > SHORT_LINE_MAX:           # never print a line the driver cannot parse: drop the summaries, keep the contract
        s = short_line({k: v for k, v in res.items() if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                                "scaling", "vs_baseline", "dtype", "data", "config", "batches_in_flight", "one_batch_at_a_time",
                                                                "roofline", "cpu_baseline", "whole_path_hbm_frac", "rccl_ranks")})
        s["truncated"] = True
        line = json.dumps(s, separators=(",", ":"))
    sys.stdout.flush()
    print(line, flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def fail(msg, rc=2, **extra):
    """one JSON line on stdout (what the driver parses) + the same text on stderr, then a non-zero exit"""
    print(json.dumps(dict({"error": msg}, **extra)))
    print("bench.py: " + msg, file=sys.stderr)
    sys.exit(rc)


def self_launch(gpus, argv, device_count=None, run=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher.  It starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (never
    os.exec*: a process must not replace itself once anything may have touched the GPU), relays its stdout / stderr and returns
    its exit code.  The launcher itself makes no HIP call (torch.cuda.device_count() does not initialise the GPU on this image).
    device_count / run: injection points of tests/test_host.py."""
    import subprocess
    have = (device_count or torch.cuda.device_count)()
    if gpus > have and not (os.environ.get("FOTG_BENCH_ALLOW_SHARED_GPU") and have >= 1):
        fail("--gpus %d requested, %d GPU(s) visible on this node: refusing to run on fewer GPUs than asked for" % (gpus, have), requested_gpus=gpus, visible_gpus=have)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return (run or subprocess.run)(cmd, env=env).returncode


def pci_bus_id(device):
    """PCI bus id string of a HIP device ("0000:05:00.0") -- which physical GPU a rank sat on"""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        hip.hipDeviceGetPCIBusId.argtypes = [C.c_char_p, C.c_int, C.c_int]
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
            return buf.value.decode()
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="steps per timed window (with batches in flight a window starts with an empty pipeline and ends with a drain: short windows under-report)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per GPU per step (BASELINE configs[2]: 64)")
    ap.add_argument("--windows", type=int, default=25, help="repeats of the timed K-step window; the median window is reported")
    ap.add_argument("--in-flight", type=int, default=4, help="batches in flight per GPU: consecutive steps go to this many engine "
                    "contexts on internal streams in turn (flowonthego_amd.FlowPipeline / fotg_pipe_*); 1 = one batch at a time "
                    "through a single context (fotg_calc_batch), which is also timed and reported beside `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time the stereo depth mode")
    ap.add_argument("--scatter-gather", action="store_true", help="multi-rank runs: also time rank 0 scattering the frames of all "
                    "ranks over RCCL and gathering the flows back (SURVEY 8e); reported beside `value`, never part of it")
    ap.add_argument("--sor-mode", type=int, default=0, help="0 lexicographic (reference order, parity mode), 1 red-black")
    ap.add_argument("--fast-math", action="store_true", help="run the timed loop in the tolerance mode of the patch loop / solvers (fotg_params::fast_math; "
                    "the line says so in `metric`); default: the parity mode, with the tolerance mode reported beside it (`fast_math`)")
    a = ap.parse_args()

    if a.gpus < 1:
        fail("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (before anything touches the GPU) and relay the ranks' output and exit code
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        # `--gpus N` is a promise about the line this run prints ("n_gpus": N): a launcher that started another number of ranks
        # must not produce a plausible-looking line for the wrong N
        fail("--gpus %d but WORLD_SIZE=%d: the launcher started %d rank(s); start one rank per GPU (python -m torch.distributed.run "
             "--nproc-per-node %d ... bench.py --gpus %d) or run `python bench.py --gpus %d` and let it launch them" %
             (a.gpus, world, world, a.gpus, a.gpus, a.gpus), rank=rank)
    ndev = torch.cuda.device_count()
    # FOTG_BENCH_ALLOW_SHARED_GPU=1 (tests on a one-GPU box): the ranks of a multi-rank run share the visible GPUs and talk over gloo
    # (RCCL refuses two ranks on one device) -- the launcher, the world-size checks, the barriers and the reductions of the N > 1
    # path run for real, the line is marked "shared_gpu_test" and is NOT a scaling measurement
    shared = bool(os.environ.get("FOTG_BENCH_ALLOW_SHARED_GPU")) and world > 1
    if shared and ndev >= 1:
        local = local % ndev
    if local >= ndev:
        fail("rank %d: LOCAL_RANK %d but %d GPU(s) visible" % (rank, local, ndev), rank=rank)
    dist = world > 1 or bool(os.environ.get("FOTG_BENCH_FORCE_DIST"))      # the switch exercises the RCCL path on a 1-GPU box
    rccl_ranks = None
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if shared:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
        rdev = torch.device("cpu") if shared else torch.device("cuda", local)          # where the reduction tensors live
        # how many ranks RCCL really connected (an all-reduce of ones), and which physical GPU each one sits on
        ones = torch.ones(1, device=rdev, dtype=torch.float64)
        td.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != a.gpus or td.get_world_size() != a.gpus:
            fail("RCCL connected %d rank(s), --gpus %d" % (rccl_ranks, a.gpus), rank=rank)
        placement = [None] * world
        td.all_gather_object(placement, {"rank": rank, "local_rank": local, "pci_bus_id": pci_bus_id(local)})
        if len({p["pci_bus_id"] for p in placement}) != world and world > 1 and not os.environ.get("FOTG_BENCH_ALLOW_SHARED_GPU"):
            fail("two ranks share one GPU: %s" % placement, rank=rank)
    else:
        placement = [{"rank": 0, "local_rank": local, "pci_bus_id": pci_bus_id(local)}]
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import flowonthego_amd as F
    from flowonthego_amd.oflow import OFClass
    lib = F.lib()                               # raises if libfotg.so is missing: no fallback
    op = F.operating_point(OP_POINT, W, 1, sor_mode=a.sor_mode)
    op.fast_math = bool(a.fast_math)
    ofc = OFClass(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=a.batch, device=local)
    I0, I1 = synth_batch(a.batch, 1234 + rank, dev)
    out = ofc.new_outflow(a.batch)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # The timed loop: every step is one full batch through the whole path.  With --in-flight D > 1 step i is submitted to engine
    # context i % D (own internal stream, own frames and output buffer of that slot) and overlaps with the steps before it -- the
    # path is latency-bound at this batch size, and a server keeps several batches in flight.  Every window ends with a host
    # wait for all slots, so exactly K complete steps lie between the two barriers.
    D = max(1, a.in_flight)
    pipe = None
    if D > 1:
        from flowonthego_amd.pipeline import FlowPipeline
        pipe = FlowPipeline(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=a.batch, depth=D, device=local)
        slots = [(I0, I1, out)] + [synth_batch(a.batch, 1234 + rank + 97 * k, dev) + (ofc.new_outflow(a.batch),) for k in range(1, D)]

    def run_steps(n, pipelined):
        if pipelined:
            for i in range(n):
                f0, f1, o = slots[i % D]
                pipe.submit(f0, f1, None, o, after_current_stream=False)       # the frames are resident, nothing to wait for
            pipe.synchronize()
        else:
            for _ in range(n):
                ofc.calc_batch(I0, I1, None, out)

    local_els = []

    def window(pipelined):
        """EXACTLY a.steps steps between barrier + synchronize on both sides; max over ranks"""
        barrier()
        t0 = time.perf_counter()
        run_steps(a.steps, pipelined)
        barrier()
        el = time.perf_counter() - t0
        if pipelined or D == 1:
            local_els.append(el)
        if dist:
            t = torch.tensor([el], device=rdev, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
        return el

    run_steps(a.warmup, False)
    if pipe:
        torch.cuda.synchronize()      # the pipe's streams do not wait for torch's: slot 0 shares its buffers with the warm-up above
        run_steps(max(a.warmup, D), True)
        torch.cuda.synchronize()
        same = all(torch.equal(ofc.calc_batch(f0, f1), o) for f0, f1, o in slots)           # pipelined results = single-context results, every slot
        if not same:
            print(json.dumps({"error": "a batch through the pipe differs from the same batch through one context", "rank": rank}))
            sys.exit(1)
    # One window of K steps at this batch is ~10 ms -- too short to be robust against clock ramp and launch jitter.  The
    # K-step window is therefore repeated (every repeat is again exactly K steps between barriers); `value` / `ms_per_step`
    # are those of the MEDIAN window, the spread is reported beside them.
    els = sorted(window(pipe is not None) for _ in range(max(1, a.windows)))
    el = els[len(els) // 2]
    ms_step = el / a.steps * 1e3
    value = world * a.batch * a.steps / el
    els1 = sorted(window(False) for _ in range(max(1, min(a.windows, 9)))) if pipe else els
    el1 = els1[len(els1) // 2]

    # host side: time to ISSUE one step's launches (submit returns at once; nothing waits) -- shows whether the launch path limits
    # the in-flight rate (ms_per_step must stay above it)
    if pipe:
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(4 * D):
            f0, f1, o = slots[i % D]
            pipe.submit(f0, f1, None, o, after_current_stream=False)
        host_issue_ms = (time.perf_counter() - t0) / (4 * D) * 1e3
        pipe.synchronize()
    per_rank = None
    if dist:
        mine = torch.tensor([sorted(local_els)[len(local_els) // 2] / a.steps * 1e3], device=rdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]

    res = {"metric": "frame-pairs/sec @1080p DIS op-pt 2" + (" (%d batches of %d in flight)" % (D, a.batch) if pipe else "") + (" [fast_math: tolerance mode]" if a.fast_math else ""), "value": value, "unit": "frame-pairs/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: batch=%d synthetic 1920x1080 gray f32 pairs per GPU, DIS op-pt 2 "
                                  "(ps=8, stride 4, scales 6-5-4, 12 LK iterations) + variational refinement on, "
                                  "%s SOR; inputs resident in HBM, output = finest-scale flow 120x68x2" %
                                  (a.batch, "lexicographic (reference order)" if a.sor_mode == 0 else "red-black"),
                      "batch_per_gpu": a.batch, "sor": "lexicographic = the reference's order" if a.sor_mode == 0 else "red-black",
                      "global_batch": world * a.batch,
                      "pairs_in_flight": world * a.batch * D,
                      "pairs_in_flight_note": "`value` is measured with %d complete batches of %d pairs resident and overlapping per GPU; "
                                              "`one_batch_at_a_time` is the figure with one batch of %d resident (the one comparable with round 1)" % (D, a.batch, a.batch),
                      "parallelism": "frame-pair sharding x%d (no collective)%s" % (world, "; %d batches in flight per GPU (engine contexts on "
                                     "internal streams, fotg_pipe_*): step i runs on context i %% %d" % (D, D) if pipe else "")},
           "timed_windows": {"n": len(els), "steps_each": a.steps, "ms_per_step_median": ms_step, "ms_per_step_min": els[0] / a.steps * 1e3,
                             "ms_per_step_max": els[-1] / a.steps * 1e3},
           "batches_in_flight": D,
           "one_batch_at_a_time": {"value": world * a.batch * a.steps / el1, "unit": "frame-pairs/s", "ms_per_step": el1 / a.steps * 1e3,
                                   "note": "the same K-step window with every step through ONE context on one stream (fotg_calc_batch), "
                                           "each step starting when the previous one has finished on the GPU"}}
    if pipe:
        res["pipeline_matches_single_context"] = bool(same)
        res["pipeline_slots_compared"] = D
        res["host_issue_ms_per_step"] = host_issue_ms
        res["host_issue_note"] = "wall time of fotg_pipe_submit (23 kernel launches) per step on the issuing thread; ms_per_step above it = the GPU, not the launch path, sets the rate"
    if per_rank is not None:
        res["ms_per_step_per_rank"] = per_rank
    res["rccl_ranks"] = rccl_ranks            # ranks an all-reduce of ones counted (None: single process, RCCL not initialised)
    if dist and shared:
        res["shared_gpu_test"] = True         # ranks share GPUs and use gloo: exercises the N > 1 entry, NOT a scaling measurement
        res["metric"] += " [SHARED-GPU TEST RUN: %d ranks on %d GPU(s), gloo]" % (world, ndev)
    res["rank_placement"] = placement

    if rank == 0:
        stream_ptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if not a.no_breakdown:
            st = stage_breakdown(ofc, I0, I1, out, lib, stream_ptr)
            res["stage_ms"] = {k: round(v, 4) for k, v in st.items()}
            res["stage_ms_note"] = "each stage alone on one stream: their sum is the time of ONE batch from start to end (one_batch_at_a_time); with batches in flight the stages of different batches overlap"
            res["roofline"] = roofline(ofc, I0, I1, lib, stream_ptr, a.batch)
            res["rooflines"] = [dict(res["roofline"], share_of_step=st["pyramid(I0,I1)"] / sum(st.values()))]
            if op.use_var_ref and a.sor_mode == 0:
                try:
                    res["roofline_dominant"] = roofline_dominant(ofc, lib, stream_ptr, a.batch, st)
                    res["rooflines"].append(res["roofline_dominant"])
                except Exception as e:          # (a configuration without a stand-alone sor_coupled launch at the finest level)
                    res["roofline_dominant"] = {"unavailable": str(e)}
                res["rooflines"].append(roofline_fused_level(ofc, a.batch, st, op.finest_scale + 1))
            res["rooflines"].append(roofline_lk(ofc, a.batch, st))
            # the whole path against the same roof: pairs/s x 16 654 080 B / 8 TB/s
            res["whole_path_hbm_frac"] = value / world * (2 * W * H * 4 + 2 * 120 * 68 * 4) / 1e9 / HBM_PEAK_GBS
            res["time_dominant_stage"] = max(st, key=st.get)
            sync = torch.cuda.synchronize
            nst = max(5, min(a.steps, 20))

            def leg_fast_math():
                # the tolerance mode (fotg_params::fast_math: patch loop csrc/lk_fast.hip.h, data term csrc/varref_dataterm.inc.h) on the headline workload -- NOT `value`
                # (the default / parity mode): same steps, one batch at a time and in flight, with the endpoint error against the parity
                # mode's flows (which are bit-identical to the CPU oracle)
                opf = F.operating_point(OP_POINT, W, 1, sor_mode=a.sor_mode)
                opf.fast_math = True
                ofcf = OFClass(opf, F.img_params(width=W, height=H, padding=opf.patch_size), max_batch=a.batch, device=local)
                outf = ofcf.new_outflow(a.batch)
                tff = timed(lambda: ofcf.calc_batch(I0, I1, None, outf), torch.cuda.synchronize, a.steps)
                stf = stage_breakdown(ofcf, I0, I1, outf, lib, stream_ptr)
                ofc.calc_batch(I0, I1, None, out); ofcf.calc_batch(I0, I1, None, outf)
                ef = torch.cat([torch.sqrt(((ofcf.upsample_crop(outf[k:k + 8]) - ofc.upsample_crop(out[k:k + 8])) ** 2).sum(-1)).flatten() for k in range(0, a.batch, 8)])
                res["fast_math"] = {"value": a.batch / tff, "unit": "frame-pairs/s", "ms_per_step": tff * 1e3,
                                    "note": "fotg_params::fast_math = 1 (tolerance mode of the patch loop and of the refinement's data term), one batch at a time; `value` above is the parity mode",
                                    "lk_stage_ms": {k: round(v, 4) for k, v in stf.items() if k.startswith("lk[")},
                                    "lk_stage_ms_parity_mode": {k: round(v, 4) for k, v in st.items() if k.startswith("lk[")},
                                    "epe_vs_parity_mode_px": {"mean": float(ef.mean()), "p99": float(torch.quantile(ef[::64], 0.99)), "max": float(ef.max()),
                                                              "note": "full-resolution flows (1920x1080) of all %d pairs against the parity mode's (== the CPU oracle)" % a.batch}}
                del ef
                if pipe:
                    pipef = FlowPipeline(opf, F.img_params(width=W, height=H, padding=opf.patch_size), max_batch=a.batch, depth=D, device=local)
                    torch.cuda.synchronize()
                    k = [0]

                    def subf():
                        f0, f1, o = slots[k[0] % D]
                        pipef.submit(f0, f1, None, o, after_current_stream=False)
                        k[0] += 1
                    tfi = timed(subf, pipef.synchronize, a.steps, warm=2 * D)
                    res["fast_math"]["in_flight"] = {"value": a.batch / tfi, "unit": "frame-pairs/s", "ms_per_step": tfi * 1e3, "batches_in_flight": D}
                    pipef.close()
                ofcf.close()
                del outf
            guard(res, 'fast_math', leg_fast_math)

            def leg_u8_frames():
                # the same batch handed over as 8-bit frames (fotg_calc_batch_u8) -- NOT the headline value:
                # the reference's API takes float32 frames (src/run_dense.cpp:144-162)
                U0, U1 = I0.to(torch.uint8), I1.to(torch.uint8)
                for _ in range(2):
                    ofc.calc_batch_u8(U0, U1, None, out)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    ofc.calc_batch_u8(U0, U1, None, out)
                torch.cuda.synchronize()
                res["u8_frames"] = {"value": a.batch * a.steps / (time.perf_counter() - t1), "unit": "frame-pairs/s",
                                    "note": "same workload with uint8 input frames (exact conversion on load), one batch at a time; informational"}
                if pipe:
                    u8 = [(f0.to(torch.uint8), f1.to(torch.uint8), o) for f0, f1, o in slots]
                    torch.cuda.synchronize()
                    for i in range(2 * D):
                        pipe.submit(u8[i % D][0], u8[i % D][1], None, u8[i % D][2], after_current_stream=False)
                    pipe.synchronize()
                    t1 = time.perf_counter()
                    for i in range(a.steps):
                        pipe.submit(u8[i % D][0], u8[i % D][1], None, u8[i % D][2], after_current_stream=False)
                    pipe.synchronize()
                    res["u8_frames"]["in_flight"] = {"value": a.batch * a.steps / (time.perf_counter() - t1), "unit": "frame-pairs/s", "batches_in_flight": D,
                                                     "note": "fotg_pipe_submit_u8: the product mode for 8-bit video (SURVEY 8f row 2), %d batches in flight" % D}
                    del u8
            guard(res, 'u8_frames', leg_u8_frames)

            def leg_u8_bgr_gray():
                # three-channel 8-bit frames of a GRAY context (fotg_params::u8_color = 1: B,G,R as cv::imread delivers; SURVEY 8f row 2
                # "RGB -> gray on device", kroeger/run_dense.cpp:199-209): OpenCV's fixed-point BGR2GRAY on load in the pyramid kernel
                opc = F.operating_point(OP_POINT, W, 1, sor_mode=a.sor_mode)
                opc.u8_color = 1
                C0 = torch.stack([I0.to(torch.uint8), torch.roll(I0, 1, 2).to(torch.uint8), torch.roll(I0, 2, 1).to(torch.uint8)], -1).contiguous()
                C1 = torch.stack([I1.to(torch.uint8), torch.roll(I1, 1, 2).to(torch.uint8), torch.roll(I1, 2, 1).to(torch.uint8)], -1).contiguous()
                ofcc = OFClass(opc, F.img_params(width=W, height=H, padding=opc.patch_size), max_batch=a.batch, device=local)
                outc = ofcc.new_outflow(a.batch)
                G0 = ((C0[..., 0].to(torch.int32) * 1868 + C0[..., 1].to(torch.int32) * 9617 + C0[..., 2].to(torch.int32) * 4899 + 8192) >> 14).to(torch.uint8)
                G1 = ((C1[..., 0].to(torch.int32) * 1868 + C1[..., 1].to(torch.int32) * 9617 + C1[..., 2].to(torch.int32) * 4899 + 8192) >> 14).to(torch.uint8)
                same_c = torch.equal(ofcc.calc_batch_u8(C0, C1, None, outc), ofc.calc_batch_u8(G0, G1))
                del G0, G1
                tcg = timed(lambda: ofcc.calc_batch_u8(C0, C1, None, outc), torch.cuda.synchronize, a.steps)
                evc = HipEvents()
                from flowonthego_amd._lib import check as _chkc
                msc = evc.time_ms(lambda: _chkc(lib.fotg_pyramid_pair_u8(ofcc._h, a.batch, C.c_void_p(C0.data_ptr()), C.c_void_p(C1.data_ptr()), 1, stream_ptr)), stream_ptr, 10)
                algc = a.batch * (2 * W * H * 3 + 2 * 120 * 68 * 4)
                res["u8_bgr_gray"] = {"value": a.batch / tcg, "unit": "frame-pairs/s", "equals_gray_u8_path": bool(same_c),
                                      "note": "uint8 B,G,R frames (n x 1080 x 1920 x 3), gray on load like cv::imread(IMREAD_GRAYSCALE) (kroeger/run_dense.cpp:199-209), one batch at a time; informational",
                                      "roofline": {"bound": "hbm", "kernel": "fotg::pyr_base_kernel<unsigned char,1,4,true,3> (both frames of %d pairs)" % a.batch,
                                                   "achieved": algc / (msc * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algc / (msc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   "traffic": None, "algorithmic_bytes_per_launch": algc, "ms_per_launch": msc}}
                if pipe:
                    pipec = FlowPipeline(opc, F.img_params(width=W, height=H, padding=opc.patch_size), max_batch=a.batch, depth=D, device=local)
                    outs_c = [ofcc.new_outflow(a.batch) for _ in range(D)]
                    torch.cuda.synchronize()
                    for i in range(2 * D):
                        pipec.submit(C0, C1, None, outs_c[i % D], after_current_stream=False)
                    pipec.synchronize()
                    t1 = time.perf_counter()
                    for i in range(a.steps):
                        pipec.submit(C0, C1, None, outs_c[i % D], after_current_stream=False)
                    pipec.synchronize()
                    res["u8_bgr_gray"]["in_flight"] = {"value": a.batch * a.steps / (time.perf_counter() - t1), "unit": "frame-pairs/s", "batches_in_flight": D}
                    pipec.close()
                    del outs_c
                ofcc.close()
                del C0, C1, outc
            guard(res, 'u8_bgr_gray', leg_u8_bgr_gray)

            def leg_single_pair_no_refine():
                # BASELINE configs[1]: ONE 1080p pair, op-pt 2's patch parameters (ps 8, stride 4, 3 scales), no variational refinement:
                # the latency of a single call (informational; the headline value is configs[2])
                op1 = F.operating_point(OP_POINT, W, 1)
                op1.use_var_ref = False
                ofc1 = OFClass(op1, F.img_params(width=W, height=H, padding=op1.patch_size), max_batch=1, device=local)
                o1 = ofc1.new_outflow(1)
                for _ in range(5):
                    ofc1.calc_batch(I0[:1], I1[:1], None, o1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(100):
                    ofc1.calc_batch(I0[:1], I1[:1], None, o1)
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t1) * 10.0
                alg1 = 2 * W * H * 4 + 2 * 120 * 68 * 4
                res["single_pair_no_refine"] = {"ms_per_pair": ms1, "value": 1e3 / ms1, "unit": "frame-pairs/s",
                                                "note": "BASELINE configs[1]: one 1080p pair per call, 3 scales, no refinement; informational",
                                                "roofline": {"bound": "hbm", "kernel": "whole call (8 launches: pyramid 2, LK 3, densify 3)", "achieved": alg1 / (ms1 * 1e-3) / 1e9,
                                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                                             "algorithmic_bytes_per_launch": alg1, "ms_per_launch": ms1,
                                                             "bound_by": "latency: eight dependent launches of one pair's work (a 1080p pair fills 2 % of the chip at the coarse levels)"}}
                ofc1.close()
            guard(res, 'single_pair_no_refine', leg_single_pair_no_refine)

            def leg_full_resolution_output():
                # the reference's post-processing on the device (kroeger/run_dense.cpp:407-414: x 2^finest, bilinear upsample, crop): the
                # full-resolution flow of the batch; SURVEY 8d counts 2 W H 4 more bytes per pair for it
                full = torch.empty((a.batch, H, W, 2), device=dev)
                sync()
                ev = HipEvents()
                from flowonthego_amd._lib import check as _chk
                msu = ev.time_ms(lambda: _chk(lib.fotg_upsample_crop(ofc._h, a.batch, C.c_void_p(out.data_ptr()), C.c_void_p(full.data_ptr()), stream_ptr)), stream_ptr, 10)
                algu = a.batch * (2 * W * H * 4 + 2 * 120 * 68 * 4)
                tfu = timed(lambda: (ofc.calc_batch(I0, I1, None, out), ofc.upsample_crop(out, full)), sync, nst)
                res["full_resolution_output"] = {"value": a.batch / tfu, "unit": "frame-pairs/s", "ms_per_step": tfu * 1e3,
                                                 "note": "fotg_calc_batch + fotg_upsample_crop (the reference's post-processing on the device), one batch at a time",
                                                 "roofline": {"bound": "hbm", "kernel": "fotg::upsample_crop_kernel (120x68x2 flow -> 1920x1080x2, %d pairs)" % a.batch,
                                                              "achieved": algu / (msu * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                              "frac": algu / (msu * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                                              "algorithmic_bytes_per_launch": algu, "ms_per_launch": msu}}
                del full
            guard(res, 'full_resolution_output', leg_full_resolution_output)

            def leg_sequence_mode():
                # video mode (fotg_calc_sequence): batch + 1 consecutive frames -> batch flows, every pyramid built once
                seq = torch.cat([I0, I1[-1:]]).contiguous()
                tsq = timed(lambda: ofc.calc_sequence(seq, None, out), sync, nst)
                res["sequence_mode"] = {"value": a.batch / tsq, "unit": "frame-pairs/s", "ms_per_step": tsq * 1e3,
                                        "note": "%d consecutive f32 frames -> %d flows, one pyramid per frame (fotg_calc_sequence), one batch at a time" % (a.batch + 1, a.batch)}
                del seq
            guard(res, 'sequence_mode', leg_sequence_mode)

            def leg_redblack():
                # red-black SOR (the ordering north_star names; not reference-equivalent): same workload, own context
                if a.sor_mode == 0:
                    oprb = F.operating_point(OP_POINT, W, 1, sor_mode=1)
                    ofcrb = OFClass(oprb, F.img_params(width=W, height=H, padding=oprb.patch_size), max_batch=a.batch, device=local)
                    outrb = ofcrb.new_outflow(a.batch)
                    trb = timed(lambda: ofcrb.calc_batch(I0, I1, None, outrb), sync, nst)
                    ofc.calc_batch(I0, I1, None, out)
                    sync()
                    d = (outrb - out) * float(1 << op.finest_scale)            # full-resolution pixels
                    res["redblack"] = {"value": a.batch / trb, "unit": "frame-pairs/s", "ms_per_step": trb * 1e3,
                                       "mean_epe_px_vs_lexicographic": float(torch.sqrt((d ** 2).sum(-1)).mean().item()),
                                       "note": "FOTG_SOR_REDBLACK: red-black ordering of the same 2x2 block update, one batch at a time; the parity mode is the "
                                               "lexicographic order of the reference's sor_coupled (`value` above)"}
                    ofcrb.close()
                    del outrb
            guard(res, 'redblack', leg_redblack)

            def leg_rgb_frames():
                # the reference's own input layout: 3-channel interleaved f32 frames (src/run_dense.cpp:147), own context, same batch
                op3 = F.operating_point(OP_POINT, W, 3, sor_mode=a.sor_mode)
                ofc3 = OFClass(op3, F.img_params(width=W, height=H, padding=op3.patch_size), max_batch=a.batch, device=local)
                R0 = torch.stack([I0, I0.roll(3, 2), I0.roll(5, 1)], -1).contiguous()
                R1 = torch.stack([I1, I1.roll(3, 2), I1.roll(5, 1)], -1).contiguous()
                t3 = timed(lambda: ofc3.calc_batch(R0, R1, None, out), sync, nst)
                ev = HipEvents()
                p3 = lambda t: C.c_void_p(t.data_ptr())
                from flowonthego_amd._lib import check as _check
                ms3 = ev.time_ms(lambda: _check(lib.fotg_pyramid_pair(ofc3._h, a.batch, p3(R0), p3(R1), 1, stream_ptr)), stream_ptr, 10)
                alg3 = a.batch * (2 * W * H * 3 * 4 + 2 * 120 * 68 * 3 * 4)
                res["rgb_frames"] = {"value": a.batch / t3, "unit": "frame-pairs/s", "ms_per_step": t3 * 1e3,
                                     "note": "same workload with 3-channel interleaved f32 frames, the layout the reference's src/ path feeds (src/run_dense.cpp:147); "
                                             "3x the input bytes; one batch at a time",
                                     "whole_path_hbm_frac": a.batch / t3 * (2 * W * H * 3 * 4 + 2 * 120 * 68 * 4) / 1e9 / HBM_PEAK_GBS,
                                     "roofline": {"bound": "hbm", "kernel": "fotg::pyr_base_kernel<float,3,4,true> (RGB frames -> pyramid level 4)", "achieved": alg3 / (ms3 * 1e-3) / 1e9,
                                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                                  "algorithmic_bytes_per_launch": alg3, "ms_per_launch": ms3}}
                if pipe:
                    from flowonthego_amd.pipeline import FlowPipeline as _FP
                    pipe3 = _FP(op3, F.img_params(width=W, height=H, padding=op3.patch_size), max_batch=a.batch, depth=D, device=local)
                    o3 = [pipe3.new_outflow(a.batch) for _ in range(D)]
                    sync()
                    k3 = [0]

                    def sub3():
                        pipe3.submit(R0, R1, None, o3[k3[0] % D], after_current_stream=False)      # (the same frames for every slot: 3.2 GB per step, far beyond any cache)
                        k3[0] += 1
                    t3f = timed(sub3, pipe3.synchronize, max(nst, 4 * D), warm=2 * D)
                    res["rgb_frames"]["in_flight"] = {"value": a.batch / t3f, "unit": "frame-pairs/s", "ms_per_step": t3f * 1e3, "batches_in_flight": D,
                                                      "same_bits_as_one_at_a_time": bool(torch.equal(o3[0], ofc3.calc_batch(R0, R1)))}
                    pipe3.close()
                    del o3
                ofc3.close()
                del R0, R1
            guard(res, 'rgb_frames', leg_rgb_frames)

            def leg_node_api():
                # the one-process multi-GPU entry (fotg_node_*: one pipe + one issuing host thread per device slot) on this rank's GPU alone:
                # the same steps through the C-ABI a C++ host would use; must not be slower than the pipe it wraps
                if pipe:
                    from flowonthego_amd.node import FlowNode
                    node = FlowNode(op, F.img_params(width=W, height=H, padding=op.patch_size), devices=[local], max_batch=a.batch, depth=D)
                    sync()
                    outs_n = [[o] for (_, _, o) in slots]
                    tickets = []

                    def node_steps(n):
                        for i in range(n):
                            f0, f1, _ = slots[i % D]
                            tickets.append(node.submit(a.batch, [f0], [f1], outs_n[i % D])[0])
                            if len(tickets) > 8:
                                node.wait(tickets.pop(0))
                        node.synchronize()
                        tickets.clear()
                    node_steps(2 * D)
                    t1 = time.perf_counter()
                    node_steps(a.steps)
                    tn = (time.perf_counter() - t1) / a.steps
                    ofc.calc_batch(slots[0][0], slots[0][1], None, out)
                    sync()
                    res["node_api"] = {"value": a.batch / tn, "unit": "frame-pairs/s", "ms_per_step": tn * 1e3, "devices": [local], "batches_in_flight": D,
                                       "same_bits_as_single_context": bool(torch.equal(outs_n[0][0], out)),
                                       "note": "fotg_node_submit / fotg_node_wait (include/fotg.h) with one device slot: K steps, one window"}
                    node.close()
            guard(res, 'node_api', leg_node_api)

            def leg_config_4k_op4():
                # BASELINE configs[3]
                res["config_4k_op4"] = config_4k_op4(F, OFClass, lib, local, dev, stream_ptr)
                ref4k = res["config_4k_op4"].pop("_full_flow")
                res["config_4k_op4_fast_math"] = config_4k_op4(F, OFClass, lib, local, dev, stream_ptr, fast=True, ref_flow=ref4k)
                # the ordering north_star names (red-black at every size, src/kernels/flowUtil.cu:297-362): ms per pair + distance to the parity mode
                res["config_4k_op4_redblack"] = config_4k_op4(F, OFClass, lib, local, dev, stream_ptr, ref_flow=ref4k, sor_mode=1)
                del ref4k
            guard(res, 'config_4k_op4', leg_config_4k_op4)
        if a.extras:
            # stereo depth mode (kroeger SELECTMODE=2): same frames as a rectified pair, one displacement channel
            opd = F.operating_point(OP_POINT, W, 1, sor_mode=0)
            opd.depth_mode = True
            ofcd = OFClass(opd, F.img_params(width=W, height=H, padding=opd.patch_size), max_batch=a.batch, device=local)
            outd = ofcd.new_outflow(a.batch)
            for _ in range(2):
                ofcd.calc_batch(I0, I1, None, outd)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                ofcd.calc_batch(I0, I1, None, outd)
            torch.cuda.synchronize()
            res["depth_mode"] = {"value": a.batch * a.steps / (time.perf_counter() - t1), "unit": "frame-pairs/s",
                                 "note": "stereo depth mode (1-D displacement, RefLevelDE refinement) on the same frames; informational"}
            ofcd.close()
            del outd
        if world == 1 and not a.no_cpu_baseline:
            def leg_cpu_baseline():
                res["cpu_baseline"] = cpu_baseline(I0, I1)
                # the same leg also checks the timed batch's result against the oracle ("EPE vs kroeger CPU")
                from oracle import oracle as O
                p = O.op_point(OP_POINT, W, 1)
                ofc.calc_batch(I0, I1, None, out)
                got = out[:2].cpu().numpy()
                ref = [O.flow(O.pad_frame(I0[k].cpu().numpy(), p.sc_f), O.pad_frame(I1[k].cpu().numpy(), p.sc_f), p, a.sor_mode) for k in range(2)]
                d = np.stack(ref) - got
                res["cpu_baseline"]["parity"] = {"pairs_checked": 2, "mean_epe_px": float(np.sqrt((d ** 2).sum(-1)).mean()),
                                                 "max_abs_diff": float(np.abs(d).max()), "bit_identical": bool((d == 0).all())}
            guard(res, "cpu_baseline", leg_cpu_baseline)
    if dist and a.scatter_gather and not shared:
        # end to end with the frames starting on rank 0 and the flows ending there (RCCL over xGMI: scatter + gather only)
        # chunked, double-buffered: chunk t+1 travels (grouped ncclSend/ncclRecv) while chunk t is computed; nothing is padded
        from flowonthego_amd.shard import PipeEngine, gather_flows_exact, pipelined_scatter_compute
        if pipe is None:
            from flowonthego_amd.pipeline import FlowPipeline
            pipe = FlowPipeline(op, F.img_params(width=W, height=H, padding=op.patch_size), max_batch=a.batch, depth=4, device=local)
        G0 = G1 = None
        if rank == 0:
            G0, G1 = I0.repeat((world,) + (1,) * (I0.dim() - 1)), I1.repeat((world,) + (1,) * (I1.dim() - 1))
        chunk = max(1, a.batch // 4)
        barrier()
        t0 = time.perf_counter()
        flows, _ = pipelined_scatter_compute(G0, G1, world * a.batch, tuple(I0.shape[1:]), I0.dtype, td,
                                             PipeEngine(pipe), chunk, src=0, device=dev)     # the same FlowPipeline as the N = 1 path
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        full = gather_flows_exact(flows, world * a.batch, td, dst=0)
        barrier()
        t3 = time.perf_counter()
        ok = True
        if rank == 0:
            # every rank's shard is a copy of rank 0's batch: the gathered flows must be `world` copies of the single-context
            # result, bit for bit (a buffer re-used too early in the chunked scatter would show up here as a stale or torn frame)
            want = ofc.calc_batch(I0, I1)
            torch.cuda.synchronize()
            ok = all(torch.equal(full[r * a.batch:(r + 1) * a.batch], want) for r in range(world))
            res["scatter_gather"] = {"scatter_plus_compute_ms": (t2 - t0) * 1e3, "gather_ms": (t3 - t2) * 1e3, "chunk_pairs": chunk,
                                     "end_to_end_pairs_per_s": world * a.batch / (t3 - t0), "gathered_shape": list(full.shape),
                                     "gathered_flows_match_single_context": bool(ok)}
        flag = torch.tensor([0 if ok else 1], device=dev)
        td.all_reduce(flag)
        if int(flag.item()):
            if rank == 0:
                print("bench.py: the flows gathered after the chunked scatter differ from the single-context result", file=sys.stderr)
            td.destroy_process_group()
            sys.exit(1)
    if dist:
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        emit(res)                # the ONE stdout line, last


if __name__ == "__main__":
    main()
