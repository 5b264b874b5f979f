"""GPU parity tests: the HIP engine (through the C-ABI of libfotg.so) against the CPU oracle, the committed golden
vectors of the reference's own FDF code, and the reference's golden .flo.  Integer/bit work is compared with ==;
the engine evaluates the oracle's f32 expressions in the same order (no FMA contraction), so floats are compared
bit-for-bit as well (np.array_equal; +0 == -0)."""
import os

import numpy as np
import pytest

from conftest import load_fdf, synth_pair

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _mods():
    import flowonthego_amd as F
    from flowonthego_amd.oflow import OFClass, VarRefClass
    from oracle import oracle as O
    return F, OFClass, VarRefClass, O


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def epe(a, b):
    return np.sqrt(((a - b) ** 2).sum(-1))


def oracle_params(O, op):
    p = O.DisParams()
    p.sc_f, p.sc_l, p.ps = op.coarsest_scale, op.finest_scale, op.patch_size
    p.max_iter = p.min_iter = op.grad_descent_iter
    if 0 <= op.min_iter <= op.grad_descent_iter:
        p.min_iter = op.min_iter
    p.dp_thresh, p.dr_thresh, p.res_thresh = op.dp_thresh, op.dr_thresh, op.res_thresh
    p.patove, p.patnorm, p.noc, p.usetvref = op.patch_stride, int(op.use_mean_normalization), op.channels, int(op.use_var_ref)
    p.tv_alpha, p.tv_gamma, p.tv_delta = op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta
    p.tv_innerit, p.tv_solverit, p.tv_sor = int(getattr(op, "var_ref_inner_iter", 1)), op.var_ref_iter, op.var_ref_sor_weight
    p.costfct, p.normoutlier, p.usefbcon = op.cost_func, op.norm_outlier, int(op.use_fbcon)
    p.depth = int(op.depth_mode)
    return p


def frames(case, alley):
    if case == "alley":
        return alley["frame_0001"].astype(np.float32), alley["frame_0002"].astype(np.float32), 1
    if case == "alley_rgb":
        return alley["rgb_crop_0001"][..., ::-1].astype(np.float32), alley["rgb_crop_0002"][..., ::-1].astype(np.float32), 3
    if case == "synth_1080p":
        a, b = synth_pair(1080, 1920, seed=1234)
        return a, b, 1
    if case == "synth_odd":            # needs horizontal and vertical padding (slow load path)
        a, b = synth_pair(270, 500, seed=77)
        return a, b, 1
    if case == "synth_rgb":
        a, b = synth_pair(200, 328, seed=9, noc=3)
        return a, b, 3
    if case == "synth_rgb_lv3":        # RGB fast path with the base level at 3 (two row groups per wave) and vertical padding
        a, b = synth_pair(515, 1024, seed=1094, noc=3)
        return a, b, 3
    if case == "synth_rgb_1080p":      # RGB fast path with the base level at 4 (four row groups), partial last strip
        a, b = synth_pair(1080, 1920, seed=1095, noc=3)
        return a, b, 3
    if case == "synth_rgb_fast":       # no horizontal padding at op-pt 2 (352 = 22 * 16): the coalesced + LDS-transposed row loads,
        a, b = synth_pair(200, 352, seed=10, noc=3)        # with a partial last 256-pixel strip
        return a, b, 3
    raise KeyError(case)


@pytest.mark.parametrize("case,op_point", [("alley", 2), ("alley_rgb", 2), ("synth_1080p", 2), ("synth_odd", 2),
                                           ("synth_odd", 3), ("synth_rgb", 1), ("synth_rgb_fast", 2), ("synth_rgb_lv3", 1),
                                           ("synth_rgb_1080p", 2)])
def test_pyramid_parity(case, op_point, alley):
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    h, w = f0.shape[:2]
    op = F.operating_point(op_point, w, noc)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    ofc.ConstructImgPyramids(dev(f0)[None], dev(f1)[None])
    P0 = O.Pyramid(O.pad_frame(f0, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    P1 = O.Pyramid(O.pad_frame(f1, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    for sl in range(op.finest_scale, op.coarsest_scale + 1):
        assert np.array_equal(ofc.level(0, sl, 0)[0].cpu().numpy(), P0.im[sl]), (sl, "I0")
        assert np.array_equal(ofc.level(0, sl, 1)[0].cpu().numpy(), P0.dx[sl]), (sl, "I0x")
        assert np.array_equal(ofc.level(0, sl, 2)[0].cpu().numpy(), P0.dy[sl]), (sl, "I0y")
        assert np.array_equal(ofc.level(1, sl, 0)[0].cpu().numpy(), P1.im[sl]), (sl, "I1")


@pytest.mark.parametrize("case,op_point", [("alley", 2), ("alley_rgb", 2), ("synth_odd", 3), ("synth_rgb", 1)])
def test_patchgrid_stages_parity(case, op_point, alley):
    """InitializeGrid / InitializeFromCoarserOF / Optimize / AggregateFlowDense per scale, incl. per-iteration trace"""
    import ctypes as C
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    h, w = f0.shape[:2]
    op = F.operating_point(op_point, w, noc)
    op.use_var_ref = False
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    F.lib().fotg_enable_taps(ofc._h, 1)
    p = oracle_params(O, op)
    P0 = O.Pyramid(O.pad_frame(f0, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    P1 = O.Pyramid(O.pad_frame(f1, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    prev_o = None
    for sl in range(op.coarsest_scale, op.finest_scale - 1, -1):
        ii = sl - op.finest_scale
        g = ofc.grid[ii]
        lw, lh = P0.level_wh(sl)
        og = O.Grid(lw, lh, sl, p)
        assert (g.GetNumPatches(), g.GetNumPatchesW(), g.GetNumPatchesH()) == (og.nop, og.nopw, og.noph)
        assert np.array_equal(np.array([g.GetRefPatchPos(i) for i in range(og.nop)], np.float32), og.pt_ref)
        og.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
        g.InitializeGrid(dev(P0.im[sl])[None], dev(P0.dx[sl])[None], dev(P0.dy[sl])[None])
        g.SetTargetImage(dev(P1.im[sl])[None])
        if prev_o is not None:
            og.init_from_coarser(prev_o)
            g.InitializeFromCoarserOF(dev(prev_o)[None])
        trace = np.zeros((og.nop, op.grad_descent_iter + 1, 4), np.float32)
        F._lib.check(F.lib().fotg_grid_set_trace(ofc._h, sl, trace.ctypes.data_as(C.c_void_p)))
        otrace = og.optimize(P1.im[sl], trace=True)
        g.Optimize()
        F.lib().fotg_grid_set_trace(ofc._h, sl, None)
        st = g.read_state(0, taps=True)
        assert np.array_equal(st["tmpl"], og.tmpl) and np.array_equal(st["tdx"], og.tdx) and np.array_equal(st["tdy"], og.tdy)
        assert np.array_equal(st["hes"], og.hes)
        assert np.array_equal(st["cnt"], og.cnt)
        assert np.array_equal(trace, otrace), "per-iteration LK trace differs at scale %d" % sl
        assert np.array_equal(st["p_iter"], og.p_iter)
        assert np.array_equal(st["pweight"], og.pweight)
        fo = og.aggregate()
        fg = g.AggregateFlowDense()[0].cpu().numpy()
        assert np.array_equal(fg, fo), "densified flow differs at scale %d" % sl
        prev_o = fo


@pytest.mark.parametrize("noc", [1, 3])
def test_varref_golden_reference_vectors(noc):
    """VarRefClass against outputs of the reference's own FDF1.0.1 code (tests/golden/fdf_ref_*.npz, fdf_ref_l4_*.npz): every
    intermediate plane of the last inner iteration and the refined flow, bit for bit -- the small levels through the fused
    per-level kernel, the 120 x 68 level (1080p level 4, five inner iterations) through the set-up / data / streaming-solver
    launches"""
    F, OFClass, VarRefClass, O = _mods()
    for name, c in load_fdf(noc).items():
        im1, im2, wx, wy, lvl = c["im1"], c["im2"], c["wx"], c["wy"], int(c["lvl"])
        _, h, w = im1.shape
        op = F.operating_point(2, 1024, noc)
        op.coarsest_scale = op.finest_scale = lvl
        ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8))
        ps = 8
        padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
        flow = dev(np.stack([wx, wy], -1))[None].contiguous()
        F.lib().fotg_enable_taps(ofc._h, 1)     # solver planes of on-chip levels are only written back for taps
        VarRefClass(dev(padlvl(im1))[None], dev(padlvl(im2))[None], ofc.iparams[0], ofc.op, flow)
        out = flow[0].cpu().numpy()
        st = ((w + 3) // 4) * 4

        def plane(nm, k=1):
            buf = np.zeros((k, h, st), np.float32)
            F._lib.check(F.lib().fotg_varref_plane(ofc._h, 0, nm.encode(), lvl, buf.ctypes.data))
            return buf[:, :, :w]
        for nm in ("mask", "sh", "sv", "a11", "a12", "a22", "b1", "b2", "du", "dv"):
            ref = c[nm]
            if nm in ("a11", "a12", "a22"):
                continue            # the reference overwrites these in place with the block inverse; checked via du/dv
            assert np.array_equal(plane(nm)[0], ref), (name, nm)
        for nm in ("Ix", "Iy", "Iz", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz"):
            assert np.array_equal(plane(nm, noc), c[nm]), (name, nm)
        assert np.array_equal(out[..., 0], c["out_x"]) and np.array_equal(out[..., 1], c["out_y"]), name


@pytest.mark.parametrize("case,op_point,sor_mode", [("alley", 2, 0), ("alley", 2, 1), ("alley", 2, 2), ("synth_rgb", 2, 2), ("synth_odd", 3, 2), ("synth_rgb", 2, 1), ("synth_odd", 3, 1), ("synth_1080p", 2, 1), ("alley_rgb", 2, 0),
                                                    ("synth_odd", 3, 0), ("synth_odd", 1, 0), ("synth_rgb", 2, 0),
                                                    ("synth_rgb_fast", 2, 0), ("synth_rgb_lv3", 1, 0)])
def test_end_to_end_parity(case, op_point, sor_mode, alley):
    """OFClass::calc on original (unpadded) frames == oracle pipeline, finest-scale flow and full-resolution flow.
    sor_mode 0 = sor_coupled (the reference's default build: the parity mode), 1 = red-black (not reference-equivalent),
    2 = sor_coupled_slow_but_readable (SURVEY row a17', the reference's OpenMP-build solver in serial order; the oracle's
    restatement is pinned == against the reference library in tests/test_oracle.py)"""
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    h, w = f0.shape[:2]
    op = F.operating_point(op_point, w, noc, sor_mode=sor_mode)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    out = ofc.calc(dev(f0), dev(f1), None, None, None)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, sor_mode)
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()
    full = ofc.upsample_crop(out[None])[0].cpu().numpy()
    wp, hp, padw, padh = O.padded_size(w, h, p.sc_f)
    assert np.array_equal(full, O.upsample_crop(ref, p.sc_l, padw, padh, w, h))
    # a second call on the same object gives the same answer (state is reset per call)
    assert np.array_equal(ofc.calc(dev(f0), dev(f1)).cpu().numpy(), ref)


@pytest.mark.parametrize("shw", ["0", "1", "2", "3"])
def test_lk_shared_window_variants(shw, alley, monkeypatch):
    """the LK kernel with ONE shared LDS area for a wave's four windows (automatic for patches of >= 144 values): forced off (0)
    and on (1) for every patch size, and with some (2) / all (3) rows of every wave on the global-memory path that serves windows
    which do not fit the shared area -- the same bits as the oracle in every variant, patch sizes 8 and 12, gray and RGB,
    including levels whose patch columns are shorter than a wave (ids wrap between columns: two groups)"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_LK_SHW", shw)
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")
    for case, op_point in (("alley", 2), ("synth_rgb", 2), ("synth_odd", 3), ("synth_odd", 4)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc)
        op.grad_descent_iter = min(op.grad_descent_iter, 24)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        out = ofc.calc_batch(torch.stack([dev(f0), dev(f1)]), torch.stack([dev(f1), dev(f0)])).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out[0], O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), (shw, case, op_point)
        assert np.array_equal(out[1], O.flow(O.pad_frame(f1, p.sc_f), O.pad_frame(f0, p.sc_f), p, 0)), (shw, case, op_point)
        ofc.close()


@pytest.mark.parametrize("shw", ["-1", "2", "3"])
def test_lk_eight_lanes_per_patch(shw, alley, natural_images, monkeypatch):
    """FOTG_LK_LPP=8: the LK kernel with eight lanes per patch (eight patches per wave, two of dis_sum()'s sixteen partials per
    lane, the shared LDS area for eight windows) -- patch sizes 8 and 12, including the per-iteration traces and the patch
    state of every scale, with the global-memory path forced for some / all rows: the oracle's bits"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_LK_LPP", "8")
    monkeypatch.setenv("FOTG_LK_SHW", shw)
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")
    for case, op_point in (("alley", 2), ("synth_odd", 3), ("synth_odd", 4), ("synth_1080p", 1)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc)
        op.grad_descent_iter = min(op.grad_descent_iter, 24)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        out = ofc.calc_batch(torch.stack([dev(f0), dev(f1)]), torch.stack([dev(f1), dev(f0)])).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out[0], O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), (shw, case, op_point)
        assert np.array_equal(out[1], O.flow(O.pad_frame(f1, p.sc_f), O.pad_frame(f0, p.sc_f), p, 0)), (shw, case, op_point)
        ofc.close()


def test_redblack_without_the_fused_level_kernel(alley, monkeypatch):
    """FOTG_SOR_REDBLACK on the launch-per-stage path of every level (FOTG_VR_PATH=2: one launch per half-sweep) == the fused
    per-level kernel's LDS half-sweeps == the oracle's red-black solver"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_PATH", "2")
    for case, op_point in (("alley", 2), ("synth_rgb", 2)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc, sor_mode=1)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 1)), case


@pytest.mark.parametrize("path", ["1", "2"])
def test_solver_fallback_paths(path, alley, monkeypatch):
    """the automatic dispatch picks the fused / sweep-pipelined LDS solvers at these sizes; force the single-wave
    global-memory solver (FOTG_VR_PATH=1) and the unfused sequence set-up -> data -> LDS solver (2) and require the same
    bits.  The switches are read when the context is created."""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_PATH", path)
    for case, op_point in (("alley", 2), ("synth_rgb", 2), ("synth_odd", 3)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), (path, case)


def test_switches_are_read_at_creation_and_debug_switches_are_compiled_out(alley, monkeypatch):
    """(a) FOTG_DEBUG_NOSOR / FOTG_DEBUG_SWEEPS (timing experiments that return wrong flow) do nothing in the shipped
    library; (b) a switch set AFTER the context exists is not seen by it (nothing under fotg_calc_batch reads the
    environment)"""
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames("alley", alley)
    op = F.operating_point(2, 1024, 1)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    monkeypatch.setenv("FOTG_DEBUG_NOSOR", "1")
    monkeypatch.setenv("FOTG_DEBUG_SWEEPS", "1")
    ofc = OFClass(op, F.img_params(width=1024, height=436, padding=8))
    assert np.array_equal(ofc.calc(dev(f0), dev(f1)).cpu().numpy(), ref)
    monkeypatch.delenv("FOTG_DEBUG_NOSOR"); monkeypatch.delenv("FOTG_DEBUG_SWEEPS")
    before = F.lib().fotg_debug_counter(b"sor_stream")
    f0, f1 = synth_pair(1080, 1920, seed=9)
    op = F.operating_point(2, 1920, 1)
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8))
    monkeypatch.setenv("FOTG_VR_STREAM", "0")             # too late for this context
    ofc.calc(dev(f0), dev(f1))
    assert F.lib().fotg_debug_counter(b"sor_stream") > before


@pytest.mark.parametrize("mode", ["0", "1"])
def test_streaming_solver_kernel(mode, monkeypatch):
    """levels of 65..96 rows are solved by vr_sor_stream_kernel: (du,dv) and the system travel through LDS rings filled by
    direct-to-LDS loads, results are written back by a writer wave while the solve runs (two rows per lane, packed f32).
    FOTG_VR_STREAM=0 at context creation: the resident-D kernel instead -- same bits.  Sizes: 1080p
    (68-row level), a narrower 68-row level, a 75-row level (second LDS geometry) and an odd row count (67)"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_STREAM", mode)
    before = F.lib().fotg_debug_counter(b"sor_stream")
    sizes = ((1920, 1080), (1280, 1050), (1600, 1200), (1904, 1072 - 8))
    expect = 0
    for w, h in sizes:
        f0, f1 = synth_pair(h, w, seed=9)
        op = F.operating_point(2, 1920, 1)                   # scales 6-5-4 for all sizes
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        out = ofc.calc_batch(dev(np.stack([f0, f1])), dev(np.stack([f1, f0]))).cpu().numpy()
        p = oracle_params(O, op)
        a, b = O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f)
        assert np.array_equal(out[0], O.flow(a, b, p, 0)) and np.array_equal(out[1], O.flow(b, a, p, 0)), (w, h)
        rows = a.shape[0] >> 4
        expect += 5 if (mode == "1" and 65 <= rows <= 96) else 0     # 5 inner iterations at level 4
        ofc.close()
    assert F.lib().fotg_debug_counter(b"sor_stream") == before + expect


@pytest.mark.parametrize("levelpipe", ["1", "0"])
def test_tile_solver_pipeline(levelpipe, monkeypatch):
    """levels of more than 96 rows are relaxed by the tile pipeline (varref_tiles.hip.h): one workgroup (solver wave + writer wave)
    per (sweep, band of 64 rows), (du,dv) handed from tile to tile through global memory behind progress words -- as one launch per
    sor_coupled call (FOTG_VR_LEVELPIPE=0) or, the default, with ALL inner iterations of the level and their data terms as one pipeline
    launch (varref_levelpipe.hip.h; calls of 2..4 sweeps).  Sizes: op-pt 3 at 1080p (levels 240x136: 3 bands, 480x272: 5 bands), a
    132-row level (third band of 4 rows), op-pt 4 on a tall frame (544 rows: 9 bands), batches of two; 1, 2 and 4 sweeps.
    Bit-identical to the oracle, no wait timed out."""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_LEVELPIPE", levelpipe)
    before = F.lib().fotg_debug_counter(b"sor_tiles")
    before_lp = F.lib().fotg_debug_counter(b"level_pipe")
    for (w, h), op_point, width_for_op, sweeps in (((1920, 1080), 3, 1920, 3), ((640, 528), 3, 640, 3), ((480, 2176), 4, 3840, 3),
                                                   ((640, 528), 3, 640, 1), ((640, 528), 3, 640, 2), ((640, 528), 3, 640, 4)):
        f0, f1 = synth_pair(h, w, seed=4)
        op = F.operating_point(op_point, width_for_op, 1)
        op.var_ref_iter = sweeps
        op.grad_descent_iter = 8                               # keep the oracle quick; the solver is what is under test
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        out = ofc.calc_batch(dev(np.stack([f0, f1])), dev(np.stack([f1, f0]))).cpu().numpy()
        p = oracle_params(O, op)
        a, b = O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f)
        assert np.array_equal(out[0], O.flow(a, b, p, 0)) and np.array_equal(out[1], O.flow(b, a, p, 0)), (w, h, sweeps)
        assert F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0
        ofc.close()
    assert F.lib().fotg_debug_counter(b"sor_tiles") > before                 # (the one-sweep case runs a launch per call in both modes)
    assert (F.lib().fotg_debug_counter(b"level_pipe") > before_lp) == (levelpipe == "1")


@pytest.mark.parametrize("levelpipe", ["1", "0"])
def test_tile_solver_rgb_frames(levelpipe, monkeypatch):
    """the tile pipeline under a three-channel data term (op-pt 3, 640 x 528 RGB: levels of 132 and 264 rows), batch of two:
    bit-identical to the oracle, no wait timed out"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_LEVELPIPE", levelpipe)
    f0, f1 = synth_pair(528, 640, seed=31, noc=3)
    op = F.operating_point(3, 640, 3)
    op.grad_descent_iter = 6
    ofc = OFClass(op, F.img_params(width=640, height=528, padding=op.patch_size), max_batch=2)
    before = F.lib().fotg_debug_counter(b"level_pipe" if levelpipe == "1" else b"sor_tiles")
    out = ofc.calc_batch(dev(np.stack([f0, f1])), dev(np.stack([f1, f0]))).cpu().numpy()
    p = oracle_params(O, op)
    a, b = O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f)
    assert np.array_equal(out[0], O.flow(a, b, p, 0)) and np.array_equal(out[1], O.flow(b, a, p, 0))
    assert F.lib().fotg_debug_counter(b"level_pipe" if levelpipe == "1" else b"sor_tiles") > before and F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0
    ofc.close()


def test_tile_solver_more_tiles_than_the_chip_holds():
    """a batch whose tall levels make more tiles than the chip has room for at once (40 pairs x 5 bands x 3 sweeps = 600 workgroups
    of three waves at the 480x272 level): the later tiles are only dispatched when earlier ones have ended, the ticket order keeps
    every wait bounded, and every pair equals the result of the same pair computed alone.  No wait timed out."""
    F, OFClass, _, O = _mods()
    pairs = [synth_pair(1080, 1920, seed=20 + k) for k in range(4)]
    op = F.operating_point(3, 1920, 1)
    op.grad_descent_iter = 4
    ip = F.img_params(width=1920, height=1080, padding=op.patch_size)
    one = OFClass(op, ip, max_batch=1)
    ref = [one.calc(dev(a), dev(b)).clone() for a, b in pairs]
    one.close()
    n = 40
    ofc = OFClass(op, ip, max_batch=n)
    f0 = dev(np.stack([pairs[k % 4][0] for k in range(n)])); f1 = dev(np.stack([pairs[k % 4][1] for k in range(n)]))
    for _ in range(2):
        out = ofc.calc_batch(f0, f1)
        torch.cuda.synchronize()
        for k in range(n):
            assert torch.equal(out[k], ref[k % 4]), k
    assert F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0
    ofc.close()


def test_flow_pipeline_batches_in_flight():
    """FlowPipeline (fotg_pipe_*): consecutive batches go to `depth` engine contexts on internal streams in turn and overlap;
    every batch has the bits of OFClass.calc_batch -- different inputs and batch sizes per submit, more submits than slots,
    float and 8-bit frames, waits on the device and on the host, results of a reused slot"""
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    w, h, nb = 640, 360, 3
    op = F.operating_point(2, w, 1)
    ip = F.img_params(width=w, height=h, padding=op.patch_size)
    ofc = OFClass(op, ip, max_batch=nb)
    pipe = FlowPipeline(op, ip, max_batch=nb, depth=3)
    assert pipe.out_size() == ofc.out_size()
    batches = []
    for k in range(8):
        n = 1 + k % nb
        fr = [synth_pair(h, w, seed=40 + 3 * k + j) for j in range(n)]
        a, b = dev(np.stack([f[0] for f in fr])), dev(np.stack([f[1] for f in fr]))
        if k % 4 == 3:                                                    # 8-bit frames
            a, b = a.round().clamp(0, 255).to(torch.uint8), b.round().clamp(0, 255).to(torch.uint8)
        batches.append((a, b))
    tickets = [pipe.submit(a, b) for a, b in batches]
    assert [t for t, _ in tickets] == list(range(8))
    pipe.wait(tickets[2][0], host=True)
    pipe.wait(tickets[7][0])                                              # the current stream waits on the device
    torch.cuda.synchronize()
    pipe.synchronize()
    for (a, b), (_, out) in zip(batches, tickets):
        ref = ofc.calc_batch_u8(a, b) if a.dtype == torch.uint8 else ofc.calc_batch(a, b)
        assert torch.equal(out, ref)
    with pytest.raises(F.FotgError):
        pipe.submit(batches[0][0][:, :-1], batches[0][1][:, :-1])
    with pytest.raises(F.FotgError):
        pipe.wait(99)
    pipe.close(); ofc.close()


def test_tile_solver_more_tiles_than_cus():
    """a launch with more tiles than the GPU has CUs (24 pairs x (3 + 5 bands) x 3 sweeps = 216 and 360 tiles on 256 CUs): roles are
    dealt by ticket in dependency order, so the resident tiles never wait for one that has not started.  Every pair of the batch is
    the same pair: 24 identical results, equal to the oracle's"""
    F, OFClass, _, O = _mods()
    n = 24
    f0, f1 = synth_pair(1080, 1920, seed=6)
    op = F.operating_point(3, 1920, 1)
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=op.patch_size), max_batch=n)
    a, b = dev(f0)[None].expand(n, -1, -1).contiguous(), dev(f1)[None].expand(n, -1, -1).contiguous()
    out = ofc.calc_batch(a, b).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    for k in range(n):
        assert np.array_equal(out[k], ref), k
    assert F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0
    ofc.close()


@pytest.mark.parametrize("path,sor_mode", [("0", 0), ("tiles", 0), ("1", 0), ("0", 2)])
def test_levels_taller_than_1024_rows(path, sor_mode, monkeypatch):
    """a refined level of more than 1024 rows (beyond 16 rows per lane of the single-wave solver): the level pipeline (21 bands),
    the tile solver with one launch per call (FOTG_VR_LEVELPIPE=0), the one-workgroup-per-pair wavefront without inter-workgroup
    waits (FOTG_VR_PATH=1: what the recompute of a stalled pipeline runs), and the same for sor_coupled_slow_but_readable.  op-pt 3 on
    a narrow tall frame refines the full-resolution level (1304 rows)"""
    F, OFClass, _, O = _mods()
    monkeypatch.setenv("FOTG_VR_PATH", "0" if path == "tiles" else path)
    if path == "tiles":
        monkeypatch.setenv("FOTG_VR_LEVELPIPE", "0")
    w, h = 304, 1300
    f0, f1 = synth_pair(h, w, seed=8)
    op = F.operating_point(3, w, 1, sor_mode=sor_mode)
    assert op.finest_scale == 0
    name = b"sor_tall" if (path == "1" or sor_mode == 2) else b"level_pipe" if path == "0" else b"sor_tiles"
    before = F.lib().fotg_debug_counter(name)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, sor_mode)
    assert out.shape[0] > 1024 and np.array_equal(out, ref)
    assert F.lib().fotg_debug_counter(name) > before
    assert F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0


def test_tall_level_with_more_than_four_sweeps():
    """var_ref_iter = 6 on a level of more than 1024 rows (the reference accepts any tv_solverit): the tile pipeline runs the
    sweeps as two launches of four and two -- sequential passes over the same system, the oracle's bits"""
    F, OFClass, _, O = _mods()
    w, h = 304, 1300
    f0, f1 = synth_pair(h, w, seed=9)
    op = F.operating_point(3, w, 1)
    op.var_ref_iter = 6
    before = F.lib().fotg_debug_counter(b"sor_tiles")
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    assert p.tv_solverit == 6
    assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0))
    assert F.lib().fotg_debug_counter(b"sor_tiles") >= before + 2
    assert F.lib().fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0


@pytest.mark.parametrize("first_data", ["1", "0"])
def test_first_data_term_in_the_setup_launch(first_data, alley, monkeypatch):
    """levels refined by separate launches (FOTG_VR_PATH=2: every level): the set-up launch also builds the system of the first
    inner iteration (FOTG_VR_FIRST_DATA=1, the default) or leaves it to a data-term launch of its own (=0): same flow, gray and
    RGB, sizes with partial tiles, and the golden reference planes of the first iteration's system"""
    F, OFClass, VarRefClass, O = _mods()
    monkeypatch.setenv("FOTG_VR_FIRST_DATA", first_data)
    monkeypatch.setenv("FOTG_VR_PATH", "2")
    for case, op_point in (("alley", 2), ("synth_rgb", 2), ("synth_odd", 3)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), case


def test_default_dispatch_reaches_every_kept_solver_variant():
    """every solver variant left in the library is what SOME default configuration runs (no switch set) -- the
    geometries are named here and the library's launch counters prove the variant ran; all == the oracle.
      * fused per-level kernel with the system in GLOBAL memory: a wide, short level whose skewed system does not fit LDS beside
        (du,dv) -- 100 x 30 (a 1600 x 480 frame at scale 4): (w + h) x h x 32 B = 133 KB;
      * vr_sor_pipe_kernel ((du,dv) resident in LDS, system streamed): levels of <= 64 rows with more than 3 000 pixels, too large
        for the fused kernel -- 120 x 60 (a 1920 x 960 frame at scale 4);
      * vr_sor_stream_kernel: levels of 65..96 rows -- 120 x 68 (1080p at scale 4, the headline);
      * tile pipeline / level pipeline: levels of more than 96 rows (the 4K quality preset) -- 240 x 136 here."""
    F, OFClass, _, O = _mods()
    L = F.lib()
    for (w, h, sc_l, sc_f, counter) in ((1600, 480, 4, 5, b"fused_cglobal"), (1920, 960, 4, 5, b"sor_pipe"), (1920, 1080, 4, 5, b"sor_stream"), (960, 544, 2, 3, b"level_pipe")):
        op = F.operating_point(2, w, 1)
        op.finest_scale, op.coarsest_scale, op.grad_descent_iter = sc_l, sc_f, 6
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        f0, f1 = synth_pair(h, w, seed=61)
        before = L.fotg_debug_counter(counter)
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        assert L.fotg_debug_counter(counter) > before, counter
        p = oracle_params(O, op)
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), (w, h)
        ofc.close()


def test_lk_partial_waves(alley):
    """the LK kernel runs four patches per wave (one per row of 16 lanes): sizes whose patch count is not a multiple of four,
    incl. RGB and ps=12"""
    F, OFClass, _, O = _mods()
    for case, op_point in (("alley", 2), ("synth_odd", 2), ("synth_rgb", 2), ("synth_odd", 3)):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(op_point, w, noc)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), case


def test_4k_quality_preset(monkeypatch):
    """3840x2160 synthetic pair at op-pt 4's geometry (ps=12, stride 3, scales 7..2 = 6 levels, refinement on a 960x544 finest level:
    9 bands of the tile pipeline) with 16 instead of 128 LK iterations (the full preset runs on its named input in
    test_natural_image_4k_quality_preset; the CPU oracle needs ~10 s for it)"""
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(2160, 3840, seed=99)
    op = F.operating_point(4, 3840, 1)
    assert (op.coarsest_scale, op.finest_scale, op.grad_descent_iter) == (7, 2, 128)
    op.grad_descent_iter = 16
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=12))
    got = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    assert got.shape == (544, 960, 2)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(got, ref)


def test_uint8_frames(alley):
    """8-bit front end: fotg_calc_batch_u8 == fotg_calc_batch on the converted frames == oracle; gray with
    the fast 4-byte loads, a size that needs the padded (slow) path, and interleaved RGB"""
    F, OFClass, _, O = _mods()
    for case in ("alley", "synth_odd", "alley_rgb"):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(2, w, noc)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        u0 = torch.from_numpy(np.ascontiguousarray(f0.astype(np.uint8))).cuda()[None]
        u1 = torch.from_numpy(np.ascontiguousarray(f1.astype(np.uint8))).cuda()[None]
        got = ofc.calc_batch_u8(u0, u1)[0].cpu().numpy()
        p = oracle_params(O, op)
        assert np.array_equal(got, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), case
        assert np.array_equal(got, ofc.calc(dev(f0), dev(f1)).cpu().numpy()), case
    # gray, rows 16-byte aligned: four rows per 16-byte load (partial last strip: 352 = 256 + 96); the same frames at an
    # address that is only 4-byte aligned take the dword loads -- same bits
    f0, f1 = synth_pair(208, 352, seed=12)
    op = F.operating_point(2, 352, 1)
    ofc = OFClass(op, F.img_params(width=352, height=208, padding=op.patch_size))
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    u0 = torch.from_numpy(f0.astype(np.uint8)).cuda()[None].contiguous()
    u1 = torch.from_numpy(f1.astype(np.uint8)).cuda()[None].contiguous()
    assert np.array_equal(ofc.calc_batch_u8(u0, u1)[0].cpu().numpy(), ref)
    buf0 = torch.zeros(u0.numel() + 64, dtype=torch.uint8, device="cuda")
    buf1 = torch.zeros(u1.numel() + 64, dtype=torch.uint8, device="cuda")
    m0, m1 = buf0[4:4 + u0.numel()].view_as(u0), buf1[4:4 + u1.numel()].view_as(u1)
    m0.copy_(u0); m1.copy_(u1)
    assert m0.data_ptr() % 16 == 4
    assert np.array_equal(ofc.calc_batch_u8(m0, m1)[0].cpu().numpy(), ref)
    # 1080p (base level 4: four row groups per wave through the LDS slab), gray and RGB: same bits as the float frames
    for noc in (1, 3):
        f0, f1 = synth_pair(1080, 1920, seed=13, noc=noc)
        op = F.operating_point(2, 1920, noc)
        ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=op.patch_size))
        u0 = torch.from_numpy(f0.astype(np.uint8)).cuda()[None].contiguous()
        u1 = torch.from_numpy(f1.astype(np.uint8)).cuda()[None].contiguous()
        assert torch.equal(ofc.calc_batch_u8(u0, u1), ofc.calc_batch(dev(f0)[None], dev(f1)[None])), noc
        ofc.close()


def gray_cv(a, first=1868, third=4899):
    """OpenCV's fixed-point BGR2GRAY on (.., 3) uint8 in B,G,R byte order (first / third = the weights of byte 0 / byte 2)"""
    a = a.astype(np.int64)
    return ((a[..., 0] * first + a[..., 1] * 9617 + a[..., 2] * third + 8192) >> 14).astype(np.uint8)


def test_uint8_colour_frames_to_gray_on_load(alley):
    """SURVEY 8f row 2 "RGB -> gray on device" (kroeger/run_dense.cpp:199-209: cv::imread(.., IMREAD_GRAYSCALE) of a colour
    file): three-channel 8-bit frames through every 8-bit entry point of a GRAY context (op.u8_color = 1: B,G,R like cv::imread,
    2: R,G,B) == the gray 8-bit path on frames converted with OpenCV's fixed-point formula == the oracle on those gray frames.
    Sizes: the alley_1 colour crop (256 x 128, 16-byte aligned rows: coalesced 16-byte loads + LDS slab, base level 3), a width
    that is only 4-byte aligned (dword loads), one that needs padding (per-pixel loads) and 1080p (base level 4, partial last
    strip)."""
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    rng = np.random.default_rng(5)

    def colour_pair(h, w, seed):
        g0, g1 = synth_pair(h, w, seed=seed, noc=3)
        return g0.astype(np.uint8), g1.astype(np.uint8)

    cases = [("alley_crop", alley["rgb_crop_0001"][..., ::-1].copy(), alley["rgb_crop_0002"][..., ::-1].copy())]       # B,G,R
    cases += [("w340", *colour_pair(200, 340, 21)), ("w331_padded", *colour_pair(203, 331, 22)), ("1080p", *colour_pair(1080, 1920, 23))]
    for name, c0, c1 in cases:
        h, w = c0.shape[:2]
        for order, (first, third) in ((1, (1868, 4899)), (2, (4899, 1868))):
            g0, g1 = gray_cv(c0, first, third), gray_cv(c1, first, third)
            op = F.operating_point(2, w, 1)
            ofg = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
            ref = ofg.calc_batch_u8(torch.from_numpy(g0).cuda()[None].contiguous(), torch.from_numpy(g1).cuda()[None].contiguous())
            if name != "1080p":
                p = oracle_params(O, op)
                f0, f1 = g0.astype(np.float32), g1.astype(np.float32)
                assert np.array_equal(ref[0].cpu().numpy(), O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), name
            opc = F.operating_point(2, w, 1)
            opc.u8_color = order
            ofc = OFClass(opc, F.img_params(width=w, height=h, padding=opc.patch_size))
            u0, u1 = torch.from_numpy(c0).cuda()[None].contiguous(), torch.from_numpy(c1).cuda()[None].contiguous()
            assert torch.equal(ofc.calc_batch_u8(u0, u1), ref), (name, order)
            # the float entry point of the same context still takes gray frames
            assert torch.equal(ofc.calc_batch(dev(g0)[None], dev(g1)[None]), ref), (name, order)
            with pytest.raises(F.FotgError):
                ofc.calc_batch_u8(torch.from_numpy(g0).cuda()[None].contiguous(), torch.from_numpy(g1).cuda()[None].contiguous())      # gray bytes: wrong shape
            ofc.close(); ofg.close()
    # sequence mode and the pipe: three colour frames -> two flows
    c = [np.ascontiguousarray(alley["rgb_crop_000%d" % k][..., ::-1]) for k in (1, 2)]
    c.append(np.ascontiguousarray(np.roll(c[1], 3, axis=1)))
    g = [gray_cv(x) for x in c]
    h, w = g[0].shape
    op = F.operating_point(2, w, 1)
    ofg = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
    ref = ofg.calc_sequence(torch.from_numpy(np.stack(g)).cuda())
    opc = F.operating_point(2, w, 1)
    opc.u8_color = 1
    ofc = OFClass(opc, F.img_params(width=w, height=h, padding=opc.patch_size), max_batch=2)
    cs = torch.from_numpy(np.stack(c)).cuda()
    assert torch.equal(ofc.calc_sequence(cs), ref)
    pipe = FlowPipeline(opc, F.img_params(width=w, height=h, padding=opc.patch_size), max_batch=2, depth=2)
    t, out = pipe.submit(cs[:2].contiguous(), cs[1:].contiguous())
    pipe.wait(t, host=True)
    assert torch.equal(out, ref)
    pipe.close()
    # a colour context must be gray
    bad = F.operating_point(2, w, 3)
    bad.u8_color = 1
    with pytest.raises(F.FotgError):
        OFClass(bad, F.img_params(width=w, height=h, padding=bad.patch_size))


@pytest.mark.parametrize("cost_func", [1, 2])
def test_patch_cost_functions(cost_func, alley):
    """L1 and pseudo-Huber patch costs (kroeger/patch.cpp:238-261; the operating points use L2):
    patch grid state and the final flow against the oracle, gray and RGB"""
    F, OFClass, _, O = _mods()
    for case in ("alley", "synth_rgb"):
        f0, f1, noc = frames(case, alley)
        h, w = f0.shape[:2]
        op = F.operating_point(2, w, noc)
        op.cost_func = cost_func
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
        assert np.array_equal(out, ref), (cost_func, case)
        op.cost_func = 0
        l2 = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size)).calc(dev(f0), dev(f1)).cpu().numpy()
        assert not np.array_equal(out, l2)                     # the switch does something
        assert np.median(epe(out, l2)) < 0.5                   # ... but it is still the same flow field


@pytest.mark.parametrize("case,op_point", [("alley", 2), ("synth_rgb", 2), ("synth_odd", 3), ("alley", 1)])
def test_forward_backward_merge(case, op_point, alley):
    """usefbcon (kroeger/oflow.cpp:160-170,193-197,233-235,269-270,291-294; patchgrid.cpp:278-375): backward
    grid + refinement at every scale but the last, both densifications merge the other grid's patches at their moved
    positions -- bit-identical to the oracle, also through the sequence entry point"""
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    h, w = f0.shape[:2]
    op = F.operating_point(op_point, w, noc)
    op.use_fbcon = True
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    assert np.array_equal(ofc.calc_sequence(dev(np.stack([f0, f1])))[0].cpu().numpy(), ref)
    op.use_fbcon = False
    plain = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size)).calc(dev(f0), dev(f1)).cpu().numpy()
    assert not np.array_equal(out, plain) and np.median(epe(out, plain)) < 0.5


def test_random_sizes_sweep():
    """seeded sweep over frame sizes (odd widths / heights, tall, wide, tiny coarsest levels), channel counts and operating
    points 1-3: every size takes different padding, grid offsets, band / lane counts and solver paths -- all bit-exact"""
    F, OFClass, _, O = _mods()
    rng = np.random.default_rng(2024)
    cases = [(int(rng.integers(150, 700)), int(rng.integers(120, 420)), int(rng.integers(1, 4)), 1 + 2 * int(rng.integers(0, 2))) for _ in range(10)]
    cases += [(1280, 720, 2, 1), (333, 800, 2, 1), (900, 130, 2, 3)]
    for w, h, op_point, noc in cases:
        f0, f1 = synth_pair(h, w, seed=w + h, noc=noc)
        op = F.operating_point(op_point, w, noc)
        try:
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        except F.FotgError:
            assert min(w, h) >> op.coarsest_scale < 5, (w, h, op_point)      # refused only when the coarsest level is degenerate
            continue
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
        assert np.array_equal(out, ref), (w, h, op_point, noc, float(np.abs(out - ref).max()))
        ofc.close()


SWITCHES = {"FOTG_VR_PATH": ["0", "1", "2"], "FOTG_VR_STREAM": ["0", "1"],
            "FOTG_VR_LEVELPIPE": ["0", "1"], "FOTG_VR_FIRST_DATA": ["0", "1"], "FOTG_PYR_SPLIT": ["1", "3", "16"],
            "FOTG_LK_SHW": ["-1", "0", "1"], "FOTG_LK_LPP": ["0", "8", "16"], "FOTG_LK_LPP_MIN_WAVES": ["1", "2048"], "FOTG_LK_BANDED": ["0", "1"]}


def test_random_switch_sweep(monkeypatch):
    """every dispatch switch of the engine (read once at fotg_create; each selects among kernels that must agree: solver paths, set-up
    variants, fused-kernel shapes, pyramid launch shapes, LK window / lanes-per-patch variants) drawn at random together with the
    frame size, operating point, channels and batch: whatever path runs, the flow is the oracle's"""
    F, OFClass, _, O = _mods()
    rng = np.random.default_rng(int(os.environ.get("FOTG_TEST_SWEEP_SEED", "4242")))
    for k in range(int(os.environ.get("FOTG_TEST_SWEEP_CASES", "24"))):
        env = {name: str(rng.choice(vals)) for name, vals in SWITCHES.items() if rng.random() < 0.6}
        for name in SWITCHES:
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        noc = 1 + 2 * int(rng.integers(0, 2))
        w, h = int(rng.integers(200, 1100)), int(rng.integers(160, 700))
        op_point = int(rng.integers(1, 5))
        op = F.operating_point(op_point, w, noc)
        op.grad_descent_iter = min(op.grad_descent_iter, 10)
        if rng.random() < 0.3:
            op.finest_scale = max(0, op.finest_scale - int(rng.integers(1, 3)))      # taller finest levels
        n = int(rng.integers(1, 4))
        try:
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        except F.FotgError:
            continue
        prs = [synth_pair(h, w, seed=900 + 3 * k + j, noc=noc) for j in range(n)]
        out = ofc.calc_batch(torch.stack([dev(q[0]) for q in prs]), torch.stack([dev(q[1]) for q in prs])).cpu().numpy()
        pr = oracle_params(O, op)
        for j in (0, n - 1):
            ref = O.flow(O.pad_frame(prs[j][0], pr.sc_f), O.pad_frame(prs[j][1], pr.sc_f), pr, 0)
            assert np.array_equal(out[j], ref), (env, w, h, op_point, noc, n, j, float(np.abs(out[j] - ref).max()))
        ofc.close()


def test_random_parameter_sweep():
    """seeded sweep over the PARAMETERS kroeger's OFClass constructor takes (oflow.h:84-111), one random combination per case: patch
    size 4 / 8 / 12 / 16, overlap, scale range, iteration counts with and without early termination, the three thresholds, mean
    normalisation, cost function, forward-backward merge, refinement weights, solver iterations and relaxation weight, the three
    solver orders, gray / RGB, odd frame sizes, with and without a second pair in the batch -- every flow bit-identical to the oracle"""
    F, OFClass, _, O = _mods()
    rng = np.random.default_rng(int(os.environ.get("FOTG_TEST_SWEEP_SEED", "77")))
    done = 0
    for k in range(int(os.environ.get("FOTG_TEST_SWEEP_CASES", "90"))):
        noc = 1 + 2 * int(rng.integers(0, 2))
        w, h = int(rng.integers(160, 520)), int(rng.integers(120, 360))
        tall = bool(os.environ.get("FOTG_TEST_SWEEP_TALL"))                # (tools/sweep_hunt.sh: narrow frames of 1 100 .. 2 600 rows at full resolution)
        if tall:
            w, h = int(rng.integers(48, 160)), int(rng.integers(1100, 2600))
        op = F.operating_point(2, w, noc)
        op.patch_size = int(rng.choice([4, 8, 12, 16]))
        op.patch_stride = float(rng.choice([0.3, 0.4, 0.5, 0.65, 0.75]))
        op.finest_scale = int(rng.integers(0, 3))
        op.coarsest_scale = op.finest_scale + int(rng.integers(0, 3))
        if tall:
            op.finest_scale, op.coarsest_scale = 0, int(rng.integers(0, 3))
        op.grad_descent_iter = int(rng.integers(2, 25))
        op.min_iter = int(rng.integers(0, op.grad_descent_iter + 1)) if rng.random() < 0.5 else -1
        op.dp_thresh = float(rng.choice([0.05, 0.01, 0.2]))
        op.dr_thresh = float(rng.choice([0.95, 0.8, 0.99]))
        op.res_thresh = float(rng.choice([0.0, 0.0, 0.5, 2.0]))
        op.use_mean_normalization = bool(rng.random() < 0.7)
        op.cost_func = int(rng.choice([0, 0, 1, 2]))
        op.use_fbcon = bool(rng.random() < 0.25)
        op.use_var_ref = bool(rng.random() < 0.8)
        op.var_ref_iter = int(rng.integers(1, 6))
        op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta = float(rng.choice([10.0, 3.0, 25.0])), float(rng.choice([10.0, 0.5, 20.0])), float(rng.choice([5.0, 0.0, 12.0]))
        op.var_ref_sor_weight = float(rng.choice([1.6, 1.0, 1.25, 1.9]))
        op.sor_mode = int(rng.choice([0, 0, 0, 2, 1]))
        op.var_ref_inner_iter = int(rng.choice([1, 1, 2, 3]))                   # kroeger tv_innerit (run_dense.cpp:288, refine_variational.cpp:36)
        desc = dict(sor=op.sor_mode, inner=op.var_ref_inner_iter, w=w, h=h, noc=noc, ps=op.patch_size, ov=op.patch_stride, sc=(op.coarsest_scale, op.finest_scale), it=(op.min_iter, op.grad_descent_iter),
                    thr=(op.dp_thresh, op.dr_thresh, op.res_thresh), norm=op.use_mean_normalization, cost=op.cost_func, fb=op.use_fbcon,
                    ref=(op.use_var_ref, op.var_ref_iter, op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta, op.var_ref_sor_weight))
        try:
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        except F.FotgError:
            continue                                          # (a coarsest level too small for the patch size: refused, like test_random_sizes_sweep)
        f0, f1 = synth_pair(h, w, seed=500 + k, noc=noc)
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        pr = oracle_params(O, op)
        ref = O.flow(O.pad_frame(f0, pr.sc_f), O.pad_frame(f1, pr.sc_f), pr, op.sor_mode)
        assert np.array_equal(out, ref), (desc, float(np.abs(out - ref).max()))
        ofc.close()
        if k % 3 == 0:                                        # the same pair as the second of a batch of two (reversed pair first)
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
            both = ofc.calc_batch(torch.stack([dev(f1), dev(f0)]), torch.stack([dev(f0), dev(f1)])).cpu().numpy()
            assert np.array_equal(both[1], ref), desc
            ofc.close()
        done += 1
    assert done >= 0.6 * int(os.environ.get("FOTG_TEST_SWEEP_CASES", "90")), done


def test_random_entry_point_sweep():
    """seeded sweep over frame SIZES (the loaders' aligned / dword / padded paths, base levels, partial strips) x every front end of
    the boundary: float frames, 8-bit gray, 8-bit colour converted on load (B,G,R and R,G,B), interleaved RGB -- alone (checked against
    the oracle), in a batch with its neighbours, as a video sequence, through a pipe with batches in flight, with the device-side
    upsample: all the same bits"""
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    rng = np.random.default_rng(int(os.environ.get("FOTG_TEST_SWEEP_SEED", "123")))
    for k in range(int(os.environ.get("FOTG_TEST_SWEEP_CASES", "12"))):
        kind = ["f32", "u8", "bgr", "rgb8", "rgbf"][k % 5]               # float gray | 8-bit gray | 8-bit colour -> gray | 8-bit RGB | float RGB
        noc = 3 if kind in ("rgb8", "rgbf") else 1
        w = int(rng.integers(96, 900)) if rng.random() < 0.7 else 16 * int(rng.integers(8, 60))
        h = int(rng.integers(80, 500))
        op_point = int(rng.integers(1, 4))
        if os.environ.get("FOTG_TEST_SWEEP_BIG"):                         # (tools/sweep_hunt.sh: HD .. 4K frames, all four operating points)
            w, h, op_point = int(rng.integers(1000, 3900)), int(rng.integers(600, 2200)), int(rng.integers(1, 5))
        n = int(rng.integers(2, 5))
        op = F.operating_point(op_point, w, noc)
        op.grad_descent_iter = min(op.grad_descent_iter, 16)
        if kind == "bgr":
            op.u8_color = int(rng.integers(1, 3))
        try:
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        except F.FotgError:
            continue
        desc = (kind, w, h, op_point, n, op.u8_color)
        # n + 1 frames of a "video": frame j+1 = frame j of another seed's pair (any content will do)
        fr = [synth_pair(h, w, seed=2000 + 7 * k + j, noc=3 if kind == "bgr" else noc)[j & 1] for j in range(n + 1)]
        if kind in ("u8", "bgr", "rgb8"):
            dv = [torch.from_numpy(np.ascontiguousarray(f.astype(np.uint8))).cuda() for f in fr]
            if kind == "bgr":
                first, third = (1868, 4899) if op.u8_color == 1 else (4899, 1868)
                host = [gray_cv(f.astype(np.uint8), first, third).astype(np.float32) for f in fr]
            else:
                host = [f.astype(np.uint8).astype(np.float32) for f in fr]
            batch = lambda a, b: ofc.calc_batch_u8(torch.stack(a), torch.stack(b))
        else:
            dv = [dev(f) for f in fr]
            host = fr
            batch = lambda a, b: ofc.calc_batch(torch.stack(a), torch.stack(b))
        pr = oracle_params(O, op)
        ref0 = O.flow(O.pad_frame(host[0], pr.sc_f), O.pad_frame(host[1], pr.sc_f), pr, 0)
        one = batch(dv[:1], dv[1:2])
        assert np.array_equal(one[0].cpu().numpy(), ref0), desc
        allb = batch(dv[:-1], dv[1:]).clone()
        assert torch.equal(allb[0], one[0]), desc
        seq = ofc.calc_sequence(torch.stack(dv))
        assert torch.equal(seq, allb), desc
        wp, hp, padw, padh = O.padded_size(w, h, pr.sc_f)
        assert np.array_equal(ofc.upsample_crop(allb)[0].cpu().numpy(), O.upsample_crop(ref0, pr.sc_l, padw, padh, w, h)), desc
        # the same pairs one per submit through a pipe, all in flight
        pipe = FlowPipeline(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=1, depth=min(n, 4))
        outs = [pipe.new_outflow(1) for _ in range(n)]
        for j in range(n):
            pipe.submit(dv[j][None], dv[j + 1][None], None, outs[j])
        pipe.synchronize()
        for j in range(n):
            assert torch.equal(outs[j][0], allb[j]), (desc, j)
        pipe.close(); ofc.close()


@pytest.mark.parametrize("w,h,op_point,noc,kw", [(7680, 4320, 2, 1, {}), (72, 6000, 2, 1, {"finest_scale": 1, "coarsest_scale": 2}),
                                                 (40, 40, 2, 1, {"finest_scale": 0, "coarsest_scale": 1}),
                                                 (17, 33, 2, 3, {"finest_scale": 0, "coarsest_scale": 0, "patch_size": 4}),
                                                 (4096, 4096, 4, 1, {"grad_descent_iter": 8}), (3000, 200, 2, 3, {"finest_scale": 2, "coarsest_scale": 4}),
                                                 (64, 9000, 3, 1, {"finest_scale": 0, "coarsest_scale": 1}),
                                                 (7680, 4320, 2, 1, {"finest_scale": 0, "coarsest_scale": 1, "grad_descent_iter": 3})])
def test_extreme_shapes(w, h, op_point, noc, kw):
    """8K frames -- also refined at FULL resolution (a level of 4 320 rows) --, levels of 3 000 and 9 000 rows (the tile solver's bands
    beyond what the chip holds), square 4K at the quality preset, tiny and one-patch levels, a very wide RGB strip: bit-identical to
    the oracle.  (Limits, refused with FOTG_ERR_UNSUPPORTED: the lexicographic refinement of levels of more than 16 384 rows or
    whose skewed arrays exceed 4 GB per pair, coarsest levels of fewer than 5 rows.)"""
    F, OFClass, _, O = _mods()
    op = F.operating_point(op_point, w, noc)
    op.grad_descent_iter = min(op.grad_descent_iter, 12)
    for k, v in kw.items():
        setattr(op, k, v)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    f0, f1 = synth_pair(h, w, seed=w + h, noc=noc)
    out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(out, ref), float(np.abs(out - ref).max())
    ofc.close()
    with pytest.raises(F.FotgError):                                   # the documented limit
        big = F.operating_point(3, 64, 1)
        big.finest_scale, big.coarsest_scale = 0, 1
        OFClass(big, F.img_params(width=64, height=20000, padding=big.patch_size))


def test_sequence_mode(alley):
    """video front end: n+1 consecutive frames -> n flows with every pyramid built once; each flow
    equals the oracle on its pair (float and 8-bit frames, gray and RGB, full max_batch)"""
    F, OFClass, _, O = _mods()
    for noc, (h, w) in ((1, (272, 480)), (3, (200, 328))):
        seq = [synth_pair(h, w, seed=40 + k, noc=noc)[0] for k in range(2)]
        seq += [synth_pair(h, w, seed=40, noc=noc)[1], synth_pair(h, w, seed=41, noc=noc)[1]]
        seq = np.stack(seq)                                               # 4 frames -> 3 pairs
        op = F.operating_point(2, w, noc)
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=3)
        p = oracle_params(O, op)
        ref = [O.flow(O.pad_frame(seq[k], p.sc_f), O.pad_frame(seq[k + 1], p.sc_f), p, 0) for k in range(3)]
        got = ofc.calc_sequence(dev(seq)).cpu().numpy()
        for k in range(3):
            assert np.array_equal(got[k], ref[k]), (noc, k)
        got8 = ofc.calc_sequence(torch.from_numpy(np.ascontiguousarray(seq.astype(np.uint8))).cuda()).cpu().numpy()
        assert np.array_equal(got8, got), noc
        # an ordinary batch call afterwards is unaffected by the shifted target pointers
        assert np.array_equal(ofc.calc_batch(dev(seq[:3]), dev(seq[1:])).cpu().numpy(), got)
        with pytest.raises(F.FotgError):
            ofc.calc_sequence(dev(np.concatenate([seq, seq[:1]])))        # 4 pairs > max_batch


@pytest.mark.parametrize("ps,stride", [(4, 0.5), (16, 0.5), (16, 0.75)])
def test_custom_patch_sizes(ps, stride):
    """parameter sets beyond the four operating points (the reference's run_* binaries take the whole optparam list on
    the command line, kroeger/run_dense.cpp:200-223): patch sizes 4 and 16 -- gray and RGB, with refinement, the
    forward-backward merge and the depth variant"""
    F, OFClass, _, O = _mods()
    for noc, fb, depth in ((1, False, False), (3, False, False), (1, True, False), (1, False, True), (3, True, True)):
        f0, f1 = synth_pair(200, 328, seed=60 + ps, noc=noc)
        op = F.operating_point(2, 328, noc)
        op.patch_size, op.patch_stride, op.use_fbcon, op.depth_mode = ps, stride, fb, depth
        op.grad_descent_iter = 8
        ofc = OFClass(op, F.img_params(width=328, height=200, padding=ps))
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
        assert out.shape == ref.shape and np.array_equal(out, ref), (ps, noc, fb, depth, float(np.abs(out - ref).max()))
        ofc.close()


@pytest.mark.parametrize("noc", [1, 3])
def test_early_termination(noc):
    """min_iter < max_iter (kroeger's own parameter; every operating point sets them equal): patches stop on the update-rate
    and residual-rate tests at different iterations (patch.cpp:279-282), two or four patches share a wave -- per-iteration
    traces, iteration counts and the flow equal the oracle"""
    import ctypes as C
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(272, 480, seed=70, noc=noc)
    varied = 0
    for dp, dr in ((0.05, 0.95), (0.05, 2.0), (0.3, 1.2)):
        op = F.operating_point(2, 480, noc)
        op.grad_descent_iter, op.min_iter, op.dp_thresh, op.dr_thresh = 16, 2, dp, dr
        ofc = OFClass(op, F.img_params(width=480, height=272, padding=op.patch_size))
        F.lib().fotg_enable_taps(ofc._h, 1)
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        p = oracle_params(O, op)
        assert p.min_iter == 2 and p.max_iter == 16
        assert np.array_equal(out, O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)), (noc, dp)
        # iteration counts of the finest scale really vary
        P0 = O.Pyramid(O.pad_frame(f0, p.sc_f), p.sc_f, p.ps)
        sl = p.sc_l
        cnt = ofc.grid[0].read_state(0, taps=True)["cnt"]
        assert cnt.max() <= 16 and cnt.min() >= 1, np.bincount(cnt).tolist()
        varied += int(cnt.min() < cnt.max())
        ofc.close()
    assert varied >= 1              # with the residual-rate test relaxed the patches stop at different iterations


def test_sequence_mode_with_switches():
    """the video entry point combined with the other switches: forward-backward merge (the backward grids take their
    templates and gradients from the NEXT frame's pyramid), L1 cost, and an initflow warm start -- each flow equals the
    ordinary batch call on the same pairs"""
    F, OFClass, _, O = _mods()
    h, w = 272, 480
    seq = np.stack([synth_pair(h, w, seed=50 + k)[0] for k in range(2)] + [synth_pair(h, w, seed=50)[1]])
    for fb, cost in ((True, 0), (False, 1), (True, 2)):
        op = F.operating_point(2, w, 1)
        op.use_fbcon, op.cost_func = fb, cost
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
        p = oracle_params(O, op)
        ref = np.stack([O.flow(O.pad_frame(seq[k], p.sc_f), O.pad_frame(seq[k + 1], p.sc_f), p, 0) for k in range(2)])
        got = ofc.calc_sequence(dev(seq)).cpu().numpy()
        assert np.array_equal(got, ref), (fb, cost)
        assert np.array_equal(ofc.calc_batch(dev(seq[:2]), dev(seq[1:])).cpu().numpy(), ref), (fb, cost)
        ofc.close()
    # warm start: the same initflow through both entry points
    op = F.operating_point(2, w, 1)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
    wp, hp, _, _ = O.padded_size(w, h, op.coarsest_scale)
    init = (np.random.default_rng(3).standard_normal((2, hp >> (op.coarsest_scale + 1), wp >> (op.coarsest_scale + 1), 2)) * 0.5).astype(np.float32)
    a = ofc.calc_sequence(dev(seq), initflow=dev(init)).cpu().numpy()
    b = ofc.calc_batch(dev(seq[:2]), dev(seq[1:]), initflow=dev(init)).cpu().numpy()
    assert np.array_equal(a, b)
    p = oracle_params(O, op)
    P0 = O.Pyramid(O.pad_frame(seq[0], p.sc_f), p.sc_f, p.ps)
    P1 = O.Pyramid(O.pad_frame(seq[1], p.sc_f), p.sc_f, p.ps)
    assert np.array_equal(a[0], O.flow_pyr(P0, P1, p, initflow=init[0]))


def test_initflow_warm_start(alley):
    """initflow (kroeger/oflow.h:91, oflow.cpp:217-220; src/oflow.cpp:268-271): coarsest-scale patches start from 2 x the
    given flow, sampled nearest-neighbour like a coarser scale"""
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames("alley", alley)
    op = F.operating_point(2, 1024, 1)
    p = oracle_params(O, op)
    rng = np.random.default_rng(3)
    ih, iw = 448 >> (p.sc_f + 1), 1024 >> (p.sc_f + 1)
    init = (rng.standard_normal((ih, iw, 2)) * 0.5).astype(np.float32)
    P0 = O.Pyramid(O.pad_frame(f0, p.sc_f), p.sc_f, p.ps); P1 = O.Pyramid(O.pad_frame(f1, p.sc_f), p.sc_f, p.ps)
    ref = O.flow_pyr(P0, P1, p, initflow=init)
    ofc = OFClass(op, F.img_params(width=1024, height=436, padding=8))
    got = ofc.calc(dev(f0), dev(f1), None, dev(init)[None], None).cpu().numpy()
    assert np.array_equal(got, ref)
    assert not np.array_equal(got, ofc.calc(dev(f0), dev(f1)).cpu().numpy())      # and it does change the result


def test_golden_flo(alley, alley_golden_flow):
    """the reference's only golden output (kroeger/flows/alley_0001.flo): mean EPE 0.026 px (asserted two-sided), the same distance the
    unmodified kroeger build has to it (0.026 / 0.17 / 0.50)"""
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames("alley", alley)
    op = F.operating_point(2, 1024, 1)
    ofc = OFClass(op, F.img_params(width=1024, height=436, padding=8))
    full = ofc.upsample_crop(ofc.calc(dev(f0), dev(f1))[None])[0].cpu().numpy()
    e = epe(full, alley_golden_flow)
    # two-sided, as in tests/test_oracle.py: the distance the unmodified kroeger build has from the file
    assert 0.0255 <= e.mean() <= 0.0264 and 0.170 <= np.percentile(e, 99) <= 0.178 and 0.49 <= e.max() <= 0.51, (e.mean(), np.percentile(e, 99), e.max())


def test_batch_1080p_parity_and_independence():
    """BASELINE config: 1080p op-pt 2 + refinement, a batch; every pair equals the oracle and equals its own
    single-pair result (pairs are independent; frame-pair sharding needs no exchange)"""
    F, OFClass, _, O = _mods()
    n = 4
    pairs = [synth_pair(1080, 1920, seed=1234 + k) for k in range(n)]
    I0 = dev(np.stack([p[0] for p in pairs])); I1 = dev(np.stack([p[1] for p in pairs]))
    op = F.operating_point(2, 1920, 1)
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=n)
    out = ofc.calc_batch(I0, I1).cpu().numpy()
    assert out.shape == (n, 68, 120, 2)
    p = oracle_params(O, op)
    for k in (0, n - 1):
        ref = O.flow(O.pad_frame(pairs[k][0], p.sc_f), O.pad_frame(pairs[k][1], p.sc_f), p, 0)
        assert np.array_equal(out[k], ref)
    single = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=1)
    assert np.array_equal(single.calc(I0[2], I1[2]).cpu().numpy(), out[2])
    # recovered flow is sane: median close to the synthetic ground truth
    _, _, gt = synth_pair(1080, 1920, seed=1234, truth=True)
    full = ofc.upsample_crop(torch.from_numpy(out[:1]).cuda())[0].cpu().numpy()
    assert np.median(epe(full, gt)) < 0.5


def test_batch64_default_path_and_depth4_pipeline():
    """BASELINE configs[2] at its real size: 64 DISTINCT 1080p pairs, op-pt 2 + refinement, the default path.  One
    calc_batch of 64; every pair == its own single-pair call, 8 of them == the oracle; the same 64 pairs through
    FlowPipeline(depth=4) as 8 submits of 8 pairs and as 4 submits of the whole batch (the configuration bench.py times):
    every submit == calc_batch."""
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    n = 64
    base = [synth_pair(1080, 1920, seed=4321 + k) for k in range(8)]

    def variant(f, t):                                                    # 8 bases x 8 rigid transforms = 64 distinct pairs
        if t & 1: f = f[:, ::-1]
        if t & 2: f = f[::-1]
        if t & 4: f = np.roll(f, (37, 101), (0, 1))
        return np.ascontiguousarray(f)
    pairs = [(variant(base[k % 8][0], k // 8), variant(base[k % 8][1], k // 8)) for k in range(n)]
    assert len({p[0].tobytes()[:4096] + p[0].tobytes()[-4096:] for p in pairs}) == n
    I0 = dev(np.stack([p[0] for p in pairs])); I1 = dev(np.stack([p[1] for p in pairs]))
    op = F.operating_point(2, 1920, 1)
    ip = F.img_params(width=1920, height=1080, padding=8)
    ofc = OFClass(op, ip, max_batch=n)
    out = ofc.calc_batch(I0, I1)
    torch.cuda.synchronize()
    assert out.shape == (n, 68, 120, 2)
    single = OFClass(op, ip, max_batch=1)
    for k in range(n):
        assert torch.equal(single.calc(I0[k], I1[k]), out[k]), k
    p = oracle_params(O, op)
    outh = out.cpu().numpy()
    for k in range(0, n, 8):                                              # one pair of every transform
        kk = k + (k // 8) % 8
        ref = O.flow(O.pad_frame(pairs[kk][0], p.sc_f), O.pad_frame(pairs[kk][1], p.sc_f), p, 0)
        assert np.array_equal(outh[kk], ref), kk
    pipe = FlowPipeline(op, ip, max_batch=8, depth=4)
    tickets = [pipe.submit(I0[8 * j:8 * j + 8], I1[8 * j:8 * j + 8]) for j in range(8)]
    pipe.synchronize()
    for j, (_, o) in enumerate(tickets):
        assert torch.equal(o, out[8 * j:8 * j + 8]), j
    pipe.close()
    pipe = FlowPipeline(op, ip, max_batch=n, depth=4)
    perm = torch.arange(n - 1, -1, -1, device=I0.device)
    I0r, I1r = I0[perm].contiguous(), I1[perm].contiguous()               # a second, different batch in flight beside the first
    torch.cuda.synchronize()
    tickets = [pipe.submit(a, b) for a, b in ((I0, I1), (I0r, I1r), (I0, I1), (I0r, I1r))]
    pipe.synchronize()
    for j, (_, o) in enumerate(tickets):
        assert torch.equal(o, out if j % 2 == 0 else out[perm]), j
    # the same batch as 8-bit frames (the frames are integer valued): fotg_pipe_submit_u8 == fotg_calc_batch_u8 == the f32 result
    U0, U1 = I0.to(torch.uint8), I1.to(torch.uint8)
    torch.cuda.synchronize()
    tickets = [pipe.submit(U0, U1) for _ in range(2)]
    pipe.synchronize()
    assert torch.equal(ofc.calc_batch_u8(U0, U1), out)
    for _, o in tickets:
        assert torch.equal(o, out)
    pipe.close(); ofc.close(); single.close()


def test_bench_distributed_leg_on_one_gpu():
    """bench.py's N > 1 code path with world size 1 on this GPU (FOTG_BENCH_FORCE_DIST=1: RCCL process group, barrier, max-over-ranks
    and per-rank times, the chunked scatter through the FlowPipeline, the exact gather): runs, and the gathered flows have the
    shape of the whole batch.  (The 1 -> 8 GPU curve itself is the driver's to measure.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FOTG_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "16", "--steps", "8", "--warmup", "2", "--windows", "3",
                        "--no-cpu-baseline", "--no-breakdown", "--scatter-gather"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000                      # ONE short line (the scatter / gather totals are scalars inside it)
    short = json.loads(lines[0])
    assert short["n_gpus"] == 1 and short["value"] > 0 and short["rccl_ranks"] == 1 and short["gpus_distinct"] == 1
    assert short["scatter_gather"]["end_to_end_pairs_per_s"] > 0 and short["scatter_gather"]["gathered_flows_match_single_context"] is True
    main = json.loads([l for l in r.stderr.splitlines() if l.startswith("bench.py detail: ")][-1][len("bench.py detail: "):])
    sg = main["scatter_gather"]
    assert len(main["ms_per_step_per_rank"]) == 1 and main["pipeline_matches_single_context"]
    assert sg["gathered_shape"] == [16, 68, 120, 2] and sg["end_to_end_pairs_per_s"] > 0 and sg["gathered_flows_match_single_context"]
    assert main["rccl_ranks"] == 1 and main["rank_placement"][0]["pci_bus_id"]


def test_hoisted_division_is_the_ieee_division(tmp_path):
    """csrc/fdiv_hoist.h (the LK solve's divisions by the loop-invariant Cholesky factors: reciprocal refinement hoisted, scaling
    and fix-up steps dropped, a range guard on the quotient): tools/div_hoist_probe.hip compares the guarded fast quotient with
    x / d bit for bit on 2^30 pairs here (random mantissas / exponents, exact multiples +- 1 ulp, quotients next to the guard's
    limits, zeros); profiles/r03_div_hoist_probe.json holds a 3.4e10-pair run"""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_hoist_probe")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-w", os.path.join(root, "tools", "div_hoist_probe.hip"), "-o", exe])
    r = subprocess.run([exe, "1024"], capture_output=True, text=True, timeout=300)
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and res["mismatches"] == 0 and res["pairs"] == 4096 * 256 * 1024 and res["passed_guard"] > res["pairs"] // 4, res


def test_single_1080p_pair_no_refinement():
    """BASELINE configs[1] exactly: ONE 1920x1080 pair, patch_size 8, stride 4 (overlap 0.4), 3 pyramid levels (6-5-4), no
    variational refinement -- finest-scale flow, full-resolution flow and the patch state of every scale against the oracle"""
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(1080, 1920, seed=4321)
    op = F.operating_point(2, 1920, 1)
    op.use_var_ref = False
    assert (op.patch_size, op.steps, op.coarsest_scale, op.finest_scale, op.n_scales) == (8, 4, 6, 4, 3)
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8), max_batch=1)
    got = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    assert p.usetvref == 0
    P0 = O.Pyramid(O.pad_frame(f0, p.sc_f), p.sc_f, p.ps); P1 = O.Pyramid(O.pad_frame(f1, p.sc_f), p.sc_f, p.ps)
    ref = O.flow_pyr(P0, P1, p)
    assert got.shape == (68, 120, 2) and np.array_equal(got, ref)
    prev = None
    for sl in (6, 5, 4):                                           # patch displacements and weights of every scale
        lw, lh = P0.level_wh(sl)
        og = O.Grid(lw, lh, sl, p)
        og.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
        if prev is not None:
            og.init_from_coarser(prev)
        og.optimize(P1.im[sl])
        st = ofc.grid[sl - 4].read_state(0)
        assert og.nop == {6: 40, 5: 135, 4: 510}[sl]
        assert np.array_equal(st["p_iter"], og.p_iter), sl
        assert np.array_equal(st["pweight"], og.pweight), sl
        prev = og.aggregate()
    assert np.array_equal(prev, ref)
    full = ofc.upsample_crop(torch.from_numpy(got[None]).cuda())[0].cpu().numpy()
    assert np.array_equal(full, O.upsample_crop(ref, p.sc_l, 0, 8, 1920, 1080))
    _, _, gt = synth_pair(1080, 1920, seed=4321, truth=True)
    assert np.median(epe(full, gt)) < 0.5


def test_op4_quality_preset_small():
    """op-pt 4 (ps=12, 128 iterations, 6 scales) on a small frame: ps=12 kernels (3 pixels per lane) + deep pyramid"""
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(384, 640, seed=21)
    op = F.operating_point(4, 640, 1)
    assert (op.patch_size, op.grad_descent_iter) == (12, 128)
    ofc = OFClass(op, F.img_params(width=640, height=384, padding=12))
    got = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(got, ref)


def test_degenerate_inputs():
    """flat frames (zero gradients -> det==0 Hessian path, patch.cpp:78-82) and identical frames"""
    F, OFClass, _, O = _mods()
    op = F.operating_point(2, 512, 1)
    ofc = OFClass(op, F.img_params(width=512, height=256, padding=8))
    p = oracle_params(O, op)
    flat = np.full((256, 512), 37.0, np.float32)
    out = ofc.calc(dev(flat), dev(flat)).cpu().numpy()
    assert np.array_equal(out, O.flow(O.pad_frame(flat, p.sc_f), O.pad_frame(flat, p.sc_f), p, 0))
    assert np.all(out == 0)
    a, _ = synth_pair(256, 512, seed=3)
    out = ofc.calc(dev(a), dev(a)).cpu().numpy()
    assert np.array_equal(out, O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(a, p.sc_f), p, 0))
    big = np.roll(a, 60, axis=1)       # large motion: outlier resets / out-of-bounds starts
    out = ofc.calc(dev(a), dev(big)).cpu().numpy()
    assert np.array_equal(out, O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(big, p.sc_f), p, 0))
    # frames scaled far out of the 8-bit range: the Cholesky factors / quotients of the LK solve leave the guarded range of the
    # hoisted division (csrc/fdiv_hoist.h) and every wave takes the compiler's IEEE divisions; LK only (the refinement's
    # constants are tuned to 8-bit data), per-scale patch results and flow against the oracle
    a, b = synth_pair(256, 512, seed=5)
    opn = F.operating_point(2, 512, 1)
    opn.use_var_ref = False
    ofn = OFClass(opn, F.img_params(width=512, height=256, padding=8))
    pn = oracle_params(O, opn)
    for scale in (1e-25, 3e-13, 7e11, 1e16):
        s0, s1 = (a * np.float32(scale)).astype(np.float32), (b * np.float32(scale)).astype(np.float32)
        out = ofn.calc(dev(s0), dev(s1)).cpu().numpy()
        ref = O.flow(O.pad_frame(s0, pn.sc_f), O.pad_frame(s1, pn.sc_f), pn, 0)
        assert np.array_equal(out, ref, equal_nan=True), scale


def test_diverged_flow_stays_inside_the_images():
    """a flow that has left the float-to-int range (a relaxation weight outside (0, 2) makes the refinement diverge; a caller may
    hand in any initflow) must not move the warp's taps out of the image: found by tools/fuzz_create.py as a GPU memory fault
    (floor(x + wx) saturates at INT_MAX and x + 1 overflowed).  The reference's (int)floor is undefined there; here the taps clamp
    like any other out-of-image position and the result is whatever the arithmetic gives (non-finite or huge), without a fault"""
    F, OFClass, VarRefClass, O = _mods()
    w, h = 352, 168
    op = F.operating_point(2, 1024, 1)
    op.coarsest_scale = op.finest_scale = 0
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=8))
    im = (torch.rand((1, h + 16, w + 16, 1), device="cuda") * 255).floor()
    for val in (1e30, float("inf"), 3e9, -3e9, -1e30, float("nan")):
        flow = torch.full((1, h, w, 2), val, device="cuda")
        VarRefClass(im, im.roll(2, 2), ofc.iparams[0], ofc.op, flow)
        torch.cuda.synchronize()
    ofc.close()
    # end to end: a negative relaxation weight at every level of a four-level pyramid down to full resolution
    for noc in (1, 3):
        op = F.operating_point(3, 350, noc)
        op.finest_scale, op.coarsest_scale, op.grad_descent_iter, op.var_ref_sor_weight = 0, 3, 16, -1.0
        ofc = OFClass(op, F.img_params(width=350, height=165, padding=op.patch_size), max_batch=2)
        f0 = (torch.rand((2, 165, 350) + ((3,) if noc == 3 else ()), device="cuda") * 255).floor()
        out = ofc.calc_batch(f0, torch.roll(f0, 2, 2))
        torch.cuda.synchronize()
        assert out.shape[0] == 2
        ofc.close()


@pytest.mark.parametrize("tool,args", [("fuzz_create.py", ["120", "5"]), ("fuzz_values.py", ["60", "5"]), ("fuzz_api.py", ["8", "5"]), ("fuzz_cabi.py", ["12", "5"])])
def test_fuzz_tools_short_run(tool, args):
    """the four fuzzers of tools/ (odd configurations, poisoned inputs, random asynchronous call sequences, null / odd C-ABI
    arguments) in a short run each, in a process of their own: a crash, a hang or a flow that differs from the single-context
    result fails the test"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (tool, r.stdout[-2000:], r.stderr[-2000:])
    assert "no crash" in r.stdout or "every one returned" in r.stdout, r.stdout[-500:]


def test_errors():
    F, OFClass, _, O = _mods()
    op = F.operating_point(2, 512, 1)
    ofc = OFClass(op, F.img_params(width=512, height=256, padding=8), max_batch=2)
    with pytest.raises(F.FotgError):
        ofc.calc_batch(torch.zeros((3, 256, 512), device="cuda"), torch.zeros((3, 256, 512), device="cuda"))   # > max_batch
    with pytest.raises(F.FotgError):
        ofc.calc(torch.zeros((128, 512), device="cuda"), torch.zeros((128, 512), device="cuda"))               # wrong size
    bad = F.operating_point(2, 512, 1)
    bad.patch_size = 10
    with pytest.raises(F.FotgError):
        OFClass(bad, F.img_params(width=512, height=256, padding=10))


@pytest.mark.parametrize("noc,u8", [(1, False), (3, False), (1, True), (3, True)])
def test_gradient_magnitude_input(noc, u8):
    """the reference's SELECTCHANNEL==2 input (kroeger/run_dense.cpp:138-147): fotg_gradient_magnitude(frames) == the oracle's
    gradient magnitude of the replicate-padded frames, and the flow of an engine fed with those images == the oracle's flow
    on the same images (frames that need padding in both directions; float and 8-bit)"""
    F, OFClass, _, O = _mods()
    h, w = 270, 500
    f0, f1 = synth_pair(h, w, seed=321, noc=noc)
    op = F.operating_point(2, w, noc)
    p = oracle_params(O, op)
    g0, g1 = O.gradient_magnitude(O.pad_frame(f0, p.sc_f)), O.gradient_magnitude(O.pad_frame(f1, p.sc_f))
    fr = np.stack([f0, f1])
    t = torch.from_numpy(fr.astype(np.uint8)).cuda() if u8 else dev(fr)
    G = F.gradient_magnitude(t, op.coarsest_scale)
    got = G.cpu().numpy()
    assert got.shape == (2,) + g0.shape
    assert np.array_equal(got[0], g0) and np.array_equal(got[1], g1)
    hp, wp = g0.shape[:2]
    ofc = OFClass(op, F.img_params(width=wp, height=hp, padding=op.patch_size))
    out = ofc.calc(G[0], G[1]).cpu().numpy()
    assert np.array_equal(out, O.flow(g0, g1, p, 0))
    with pytest.raises(F.FotgError):
        F.gradient_magnitude(t.cpu(), op.coarsest_scale)


def test_context_lifecycle_and_host_threads():
    """contexts are independent objects: (1) creating and destroying them returns every byte (40 create / calc / destroy
    rounds, device memory in use afterwards = before), (2) two host threads, each with a context and a stream of its own,
    computing at the same time get the flows a single thread gets (the library's only shared state -- launch counters, the
    dynamic-LDS attribute cache -- is atomic / mutex-guarded)"""
    import threading
    F, OFClass, _, O = _mods()
    w, h = 640, 360
    op = F.operating_point(2, w, 1)
    ip = F.img_params(width=w, height=h, padding=8)
    pairs = [synth_pair(h, w, seed=40 + k) for k in range(4)]
    A = [dev(np.stack([p[0] for p in pairs[:2]])), dev(np.stack([p[0] for p in pairs[2:]]))]
    B = [dev(np.stack([p[1] for p in pairs[:2]])), dev(np.stack([p[1] for p in pairs[2:]]))]
    ofc = OFClass(op, ip, max_batch=2)
    serial = [ofc.calc_batch(A[k], B[k]).cpu().numpy() for k in range(2)]
    ofc.close()
    import gc
    gc.collect()                                  # (contexts of earlier tests that are still waiting for their finalizer)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        o = OFClass(op, ip, max_batch=2)
        o.calc_batch(A[0], B[0])
        o.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (4 << 20), (free0, torch.cuda.mem_get_info()[0])      # nothing is left behind
    results, errors = [None, None], []

    def worker(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                o = OFClass(op, ip, max_batch=2)
                out = None
                for _ in range(25):
                    out = o.calc_batch(A[k], B[k])
                st.synchronize()
                results[k] = out.cpu().numpy()
                o.close()
        except Exception as e:           # noqa: BLE001 -- reported below
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert np.array_equal(results[k], serial[k])
    pr = oracle_params(O, op)
    assert np.array_equal(serial[0][0], O.flow(O.pad_frame(pairs[0][0], pr.sc_f), O.pad_frame(pairs[0][1], pr.sc_f), pr, 0))


def test_stalled_wait_heals_on_a_level_of_more_than_1024_rows(monkeypatch):
    """the same recompute where no single wave reaches: a full-resolution level of 1 304 rows.  The first pass runs the level
    pipeline, the recompute after an injected stall the one-workgroup-per-pair wavefront (vr_sor_tall_kernel: no inter-workgroup
    waits) -- the oracle's bits, and no error for a valid call"""
    import ctypes as C
    F, OFClass, _, O = _mods()
    L = F.lib()
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")
    w, h = 304, 1300
    op = F.operating_point(3, w, 1)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    a, b = synth_pair(h, w, seed=8)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0)
    A, B = dev(a), dev(b)
    ow, oh = ofc.out_size()
    host = np.zeros((oh, ow, 2), np.float32)
    args = (ofc._h, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), None, host.ctypes.data_as(C.c_void_p))
    waits = lambda: L.fotg_debug_counter(b"sor_tiles") + L.fotg_debug_counter(b"level_pipe")
    w0, t0 = waits(), L.fotg_debug_counter(b"sor_tall")
    assert L.fotg_calc(*args) == 0 and np.array_equal(host, ref)
    w1 = waits()
    assert w1 > w0 and L.fotg_debug_counter(b"sor_tall") == t0
    L.fotg_ctx_counter(ofc._h, b"inject_stall")
    host[:] = 0
    assert L.fotg_calc(*args) == 0 and np.array_equal(host, ref)          # healed
    assert L.fotg_debug_counter(b"sor_tall") > t0 and waits() - w1 == w1 - w0     # the second pass: tall kernel, no waiting kernels
    assert L.fotg_ctx_counter(ofc._h, b"stalls") == 1
    ofc.close()


def test_stalled_wait_heals_at_the_host_sync_points(monkeypatch):
    """a bounded inter-workgroup wait that times out raises a word in pinned host memory.  The entry points that synchronise with
    the host and still have the call's inputs (fotg_calc, fotg_pipe_wait(host), fotg_pipe_sync) RECOMPUTE the batch on the solver
    path without inter-workgroup waits and succeed with the oracle's bits; asynchronous callers see the flag once through
    take_stall; fotg_pipe_wait(host_wait = 2) reports instead of recomputing.  The level sizes make the first run go through the
    tile solver (a 128-row level) and the recomputation through the single-wave kernel."""
    import ctypes as C
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    L = F.lib()
    w, h = 1024, 1024
    op = F.operating_point(2, w, 1)
    ip = F.img_params(width=w, height=h, padding=8)
    plain = OFClass(op, ip, max_batch=1)
    assert L.fotg_ctx_counter(plain._h, b"inject_stall") == -1            # the tap is dead in an ordinary context
    plain.close()
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")                             # read once at fotg_create
    ofc = OFClass(op, ip, max_batch=1)
    a, b = synth_pair(h, w, seed=5)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0)
    A, B = dev(a), dev(b)
    ow, oh = ofc.out_size()
    assert oh > 96                                                        # the finest level runs on the tile solver
    host = np.zeros((oh, ow, 2), np.float32)
    args = (ofc._h, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), None, host.ctypes.data_as(C.c_void_p))
    count = lambda: L.fotg_debug_counter(b"sor_tiles") + L.fotg_debug_counter(b"level_pipe")      # launches with inter-workgroup waits
    tiles0 = count()
    assert L.fotg_calc(*args) == 0 and L.fotg_ctx_counter(ofc._h, b"stalls") == 0
    tiles1 = count()
    assert tiles1 > tiles0 and np.array_equal(host, ref)
    L.fotg_ctx_counter(ofc._h, b"inject_stall")
    assert L.fotg_ctx_counter(ofc._h, b"stalls") == 1                     # (non-synchronising query)
    host[:] = 0
    assert L.fotg_calc(*args) == 0                                        # healed: no error for a valid call
    tiles2 = count()
    assert np.array_equal(host, ref) and L.fotg_ctx_counter(ofc._h, b"stalls") == 1
    assert tiles2 - tiles1 == tiles1 - tiles0                             # the second pass used neither the tile solver nor the level pipeline
    assert L.fotg_calc(*args) == 0 and np.array_equal(host, ref) and L.fotg_ctx_counter(ofc._h, b"stalls") == 1
    # asynchronous callers: the consuming query sees every stall once, and a stale flag is never blamed on a later call
    out = ofc.calc_batch(A[None], B[None])
    torch.cuda.synchronize()
    assert ofc.take_stall() is False
    L.fotg_ctx_counter(ofc._h, b"inject_stall")
    assert ofc.take_stall() is True and ofc.take_stall() is False and L.fotg_ctx_counter(ofc._h, b"stalls") == 2
    ofc.close()
    # a pipe with several batches in flight: the flag of a context makes every unverified batch of THAT context a suspect
    pipe = FlowPipeline(op, ip, max_batch=1, depth=2)
    outs = [pipe.new_outflow(1) for _ in range(4)]
    torch.cuda.synchronize()
    ts = [pipe.submit(A[None], B[None], None, outs[k])[0] for k in range(4)]       # tickets 0, 2 on slot 0; 1, 3 on slot 1
    torch.cuda.synchronize()                                              # (device-wide: the four batches have run)
    L.fotg_ctx_counter(pipe.context(0), b"inject_stall")
    for o in outs:
        o.zero_()                                                         # (whatever the first pass wrote: the recomputation must rewrite it)
    torch.cuda.synchronize()
    pipe.wait(ts[0], host=True)                                           # no exception: tickets 0 and 2 are recomputed
    assert np.array_equal(outs[0][0].cpu().numpy(), ref) and np.array_equal(outs[2][0].cpu().numpy(), ref)
    assert float(outs[1].abs().max()) == 0 and float(outs[3].abs().max()) == 0    # slot 1 was not touched
    pipe.wait(ts[2], host=True); pipe.wait(ts[1], host=True); pipe.wait(ts[3], host=True)
    assert L.fotg_ctx_counter(pipe.context(0), b"stalls") == 1 and L.fotg_ctx_counter(pipe.context(1), b"stalls") == 0
    # fotg_pipe_sync heals as well
    t, o = pipe.submit(A[None], B[None])
    L.fotg_ctx_counter(pipe.context(t % 2), b"inject_stall")
    pipe.synchronize()
    assert np.array_equal(o[0].cpu().numpy(), ref)
    # host_wait = 2: report, do not recompute -- and the ticket keeps its status
    t, o = pipe.submit(A[None], B[None])
    L.fotg_ctx_counter(pipe.context(t % 2), b"inject_stall")
    assert L.fotg_pipe_wait(pipe._h, t, None, 2) == 5 and L.fotg_pipe_wait(pipe._h, t, None, 2) == 5 and L.fotg_pipe_wait(pipe._h, t, None, 1) == 5
    t2, o2 = pipe.submit(A[None], B[None])
    assert L.fotg_pipe_wait(pipe._h, t2 + 1, None, 1) == 1                # FOTG_ERR_ARG: not submitted yet
    ev = C.c_void_p()
    assert L.fotg_pipe_ticket_event(pipe._h, t2 + 1, ev) == 1 and L.fotg_pipe_ticket_event(pipe._h, t2, ev) == 0
    pipe.wait(t2, host=True)
    assert np.array_equal(o2[0].cpu().numpy(), ref)
    # device-side wait: the host checks after its own synchronisation
    t, _ = pipe.submit(A[None], B[None])
    pipe.wait(t, host=False)
    torch.cuda.synchronize()
    L.fotg_ctx_counter(pipe.context(0), b"inject_stall"); L.fotg_ctx_counter(pipe.context(1), b"inject_stall")
    assert pipe.take_stalls() == 2 and pipe.take_stalls() == 0
    pipe.close()


def test_recompute_only_where_the_buffers_are_still_in_place(monkeypatch):
    """ADVICE round 5.  The self-healing host wait recomputes a suspect batch from the pointers of its submit -- legal only while the
    caller still holds the buffers.  (1) a ticket handed to a stream (wait(ticket), host=False) or out as an event is never
    recomputed: it is REPORTED (FOTG_ERR_STALL on every host wait for it) and its outflow is left alone; (2) the same for a submit
    with no_recompute (FOTG_SUBMIT_NO_RECOMPUTE: what the node uses for pulled scatter pieces); (3) a suspect older than the
    4 * depth submissions the pipe keeps arguments for is reported too (it used to be waited for as FOTG_OK), while the younger
    suspects of the same slot are recomputed."""
    import ctypes as C
    F, OFClass, _, O = _mods()
    from flowonthego_amd.pipeline import FlowPipeline
    L = F.lib()
    w, h = 1024, 1024
    op = F.operating_point(2, w, 1)
    ip = F.img_params(width=w, height=h, padding=8)
    monkeypatch.setenv("FOTG_TEST_TAPS", "1")
    a, b = synth_pair(h, w, seed=5)
    A, B = dev(a), dev(b)
    ref = OFClass(op, ip, max_batch=1).calc_batch(A[None], B[None])[0].cpu().numpy()
    STALL = 5
    pipe = FlowPipeline(op, ip, max_batch=1, depth=2)
    outs = [pipe.new_outflow(1) for _ in range(4)]
    torch.cuda.synchronize()
    # (1) ticket 0 handed to the current stream, ticket 2 (same slot) not
    ts = [pipe.submit(A[None], B[None], None, outs[k])[0] for k in range(4)]
    pipe.wait(ts[0], host=False)
    torch.cuda.synchronize()
    L.fotg_ctx_counter(pipe.context(0), b"inject_stall")
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    assert L.fotg_pipe_wait(pipe._h, ts[2], None, 1) == 0                    # ticket 2: recomputed
    assert np.array_equal(outs[2][0].cpu().numpy(), ref)
    assert float(outs[0].abs().max()) == 0                                   # ticket 0: NOT written again (the caller may have recycled it)
    assert L.fotg_pipe_wait(pipe._h, ts[0], None, 1) == STALL and L.fotg_pipe_wait(pipe._h, ts[0], None, 2) == STALL
    assert L.fotg_pipe_wait(pipe._h, ts[1], None, 1) == 0 and L.fotg_pipe_wait(pipe._h, ts[3], None, 1) == 0
    pipe.synchronize()                                                       # nothing new to report
    # ... and through the event hand-out
    t, o = pipe.submit(A[None], B[None], None, outs[0])
    ev = C.c_void_p()
    assert L.fotg_pipe_ticket_event(pipe._h, t, ev) == 0
    torch.cuda.synchronize()
    L.fotg_ctx_counter(pipe.context(t % 2), b"inject_stall")
    o.zero_(); torch.cuda.synchronize()
    with pytest.raises(F.FotgError):
        pipe.synchronize()                                                   # reported by the sync as well
    assert float(o.abs().max()) == 0 and L.fotg_pipe_wait(pipe._h, t, None, 1) == STALL
    # (2) FOTG_SUBMIT_NO_RECOMPUTE
    t, o = pipe.submit(A[None], B[None], None, outs[1], no_recompute=True)
    torch.cuda.synchronize()
    assert np.array_equal(o[0].cpu().numpy(), ref)                           # (an ordinary submit otherwise)
    L.fotg_ctx_counter(pipe.context(t % 2), b"inject_stall")
    o.zero_(); torch.cuda.synchronize()
    assert L.fotg_pipe_wait(pipe._h, t, None, 1) == STALL and float(o.abs().max()) == 0
    assert L.fotg_pipe_submit_ex(pipe._h, 1, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), 0, None, C.c_void_p(o.data_ptr()), C.c_void_p(-1), 2, None) == 1   # unknown flag
    pipe.close()
    # (3) more than 4 * depth submissions outstanding when the flag is found
    pipe = FlowPipeline(op, ip, max_batch=1, depth=2)
    outs = [pipe.new_outflow(1) for _ in range(10)]
    torch.cuda.synchronize()
    ts = [pipe.submit(A[None], B[None], None, outs[k])[0] for k in range(10)]        # ring = 8 tickets: 0 and 1 have dropped out
    torch.cuda.synchronize()
    L.fotg_ctx_counter(pipe.context(0), b"inject_stall")
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    assert L.fotg_pipe_wait(pipe._h, ts[0], None, 1) == STALL                # nothing is known about it any more: a suspect for good
    assert float(outs[0].abs().max()) == 0
    for k in (2, 4, 6, 8):                                                   # the slot's younger suspects were recomputed by that wait
        assert np.array_equal(outs[k][0].cpu().numpy(), ref), k
        assert L.fotg_pipe_wait(pipe._h, ts[k], None, 1) == 0
    for k in (1, 3, 5, 7, 9):                                                # the other slot: no flag, no suspects -- also the one older than the ring
        assert L.fotg_pipe_wait(pipe._h, ts[k], None, 1) == 0 and float(outs[k].abs().max()) == 0
    assert L.fotg_pipe_wait(pipe._h, ts[0], None, 1) == STALL
    pipe.close()


def test_tv_innerit_on_every_refinement_path(monkeypatch):
    """VERDICT round 5, missing #3: kroeger's run_dense reads tv_innerit from argv (run_dense.cpp:288) and RefLevelOF multiplies it into
    the inner count (refine_variational.cpp:36: tv_innerit * (level + 1)).  Both shims expose it (opt_params::var_ref_inner_iter) and
    every dispatch of the refinement honours it, bit for bit against the oracle: the fused small levels, the streaming solver, the
    level pipeline of tall levels (inner <= FOTG_LP_KMAX = 8: one launch), its fall-back to one tile-solver launch per call when
    inner > FOTG_LP_KMAX, the single-wave path, red-black and sor_coupled_slow_but_readable"""
    import subprocess
    F, OFClass, _, O = _mods()
    L = F.lib()
    w, h = 640, 1088                      # scales 1..5: 320 x 544, 160 x 272, 80 x 136 (tile pipeline: more than 96 rows), 40 x 68 and 20 x 34 (fused)
    f0, f1 = synth_pair(h, w, seed=31)
    for innerit, sor, lp_expected in ((2, 0, True), (3, 0, True), (5, 0, False), (2, 1, None), (2, 2, None)):
        op = F.operating_point(2, w, 1, sor_mode=sor)
        op.finest_scale, op.coarsest_scale, op.grad_descent_iter = 1, 5, 6
        op.var_ref_inner_iter = innerit
        assert op.to_c().tv_innerit == innerit
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        before = L.fotg_debug_counter(b"level_pipe")
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        lp = L.fotg_debug_counter(b"level_pipe") - before
        p = oracle_params(O, op)
        assert p.tv_innerit == innerit
        ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, sor)
        assert np.array_equal(out, ref), (innerit, sor, float(np.abs(out - ref).max()))
        if lp_expected is True:
            # inner = innerit * (l + 1) at the tall scales 1, 2, 3: 4, 6, 8 with innerit = 2 (three pipeline launches); 6, 9, 12 with
            # innerit = 3: scales 2 and 3 exceed FOTG_LP_KMAX and fall back while scale 1 still runs as a pipeline
            assert lp == (3 if innerit == 2 else 1), (innerit, lp)
        if lp_expected is False:
            assert lp == 0                                                  # 10 and 15 inner iterations: no level fits one launch
        assert L.fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0
        ofc.close()
    # a different count really is a different flow (the parameter is not ignored on either side)
    op1 = F.operating_point(2, w, 1, sor_mode=2); op1.finest_scale, op1.coarsest_scale, op1.grad_descent_iter = 1, 5, 6
    o1 = OFClass(op1, F.img_params(width=w, height=h, padding=8))
    assert not np.array_equal(o1.calc(dev(f0), dev(f1)).cpu().numpy(), out)
    o1.close()
    # the single-wave path (what a recompute after a stall runs)
    monkeypatch.setenv("FOTG_VR_PATH", "1")
    op = F.operating_point(2, w, 1); op.finest_scale, op.coarsest_scale, op.grad_descent_iter, op.var_ref_inner_iter = 1, 5, 6, 2
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=8))
    p = oracle_params(O, op)
    assert np.array_equal(ofc.calc(dev(f0), dev(f1)).cpu().numpy(), O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0))
    ofc.close()


def test_cpp_shim_run_dense_example(tmp_path):
    """examples/run_dense_min.cpp: raw frames -> OFClass::calc through the C++ shim -> device-side upsample + crop -> .flo,
    compared bit for bit with the oracle's full-resolution flow (the shape of src/run_dense.cpp:120-305)"""
    import subprocess
    from test_host import _build_example
    F, OFClass, _, O = _mods()
    from flowonthego_amd.flo import read_flo
    exe = _build_example(tmp_path)
    for noc, (h, w) in ((1, (272, 480)), (3, (200, 328))):
        f0, f1 = synth_pair(h, w, seed=77, noc=noc)
        p0, p1, out = (str(tmp_path / n) for n in ("f0.raw", "f1.raw", "out.flo"))
        f0.astype(np.float32).tofile(p0)
        f1.astype(np.float32).tofile(p1)
        r = subprocess.run([exe, p0, p1, str(w), str(h), str(noc), out, "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(read_flo(out), O.full_flow(f0, f1, op=2)), noc
        # kroeger's tv_innerit argument (run_dense.cpp:288) through the C++ shim's opt_params::var_ref_inner_iter
        r = subprocess.run([exe, p0, p1, str(w), str(h), str(noc), out, "2", "innerit=2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        pr = O.op_point(2, w, noc)
        pr.tv_innerit = 2
        got = read_flo(out)
        assert np.array_equal(got, O.full_flow(f0, f1, params=pr)) and not np.array_equal(got, O.full_flow(f0, f1, op=2)), noc


def test_cpp_video_pipeline_example(tmp_path):
    """examples/video_pipeline.cpp: consecutive frame pairs of a clip submitted to OFC::FlowPipeline (include/fotg/pipeline.h)
    behind their uploads, three pairs in flight -- every flow equals the oracle's for that pair"""
    import subprocess
    from test_host import _build_example
    F, OFClass, _, O = _mods()
    exe = _build_example(tmp_path, "video_pipeline")
    h, w, n = 272, 480, 6
    fr = [synth_pair(h, w, seed=90 + k)[0] for k in range(n)]
    src, out = str(tmp_path / "frames.raw"), str(tmp_path / "flows.raw")
    np.stack(fr).astype(np.float32).tofile(src)
    r = subprocess.run([exe, src, str(w), str(h), str(n), out, "3", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    op = F.operating_point(2, w, 1)
    p = oracle_params(O, op)
    ow, oh = O.padded_size(w, h, p.sc_f)[0] >> p.sc_l, O.padded_size(w, h, p.sc_f)[1] >> p.sc_l
    flows = np.fromfile(out, np.float32).reshape(n - 1, oh, ow, 2)
    for k in range(n - 1):
        assert np.array_equal(flows[k], O.flow(O.pad_frame(fr[k], p.sc_f), O.pad_frame(fr[k + 1], p.sc_f), p, 0)), k


def test_cpp_shim_reference_scale_loop(tmp_path):
    """examples/oflow_scale_loop.cpp: the reference's scale loop written with the shim classes, grids and VarRefClass constructed
    with the reference's own constructor signatures (src/oflow.cpp:101, :332; the context comes from the registry), equals
    OFClass::calc (exit code 0) and the oracle; with verbosity 2 the reference's TIME lines come out (src/oflow.cpp:343, :356)"""
    import subprocess
    from test_host import _build_example
    F, OFClass, _, O = _mods()
    exe = _build_example(tmp_path, "oflow_scale_loop")
    h, w = 272, 480
    f0, f1 = synth_pair(h, w, seed=78)
    p0, p1, out = (str(tmp_path / n) for n in ("f0.raw", "f1.raw", "out.raw"))
    f0.astype(np.float32).tofile(p0)
    f1.astype(np.float32).tofile(p1)
    r = subprocess.run([exe, p0, p1, str(w), str(h), "1", out, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "scale loop by hand == OFClass::calc" in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("TIME (Sc:")]
    assert len(lines) == 3 and "TIME (O.Flow Run-Time   ) (ms):" in r.stdout and "[optiTime]" in r.stdout, r.stdout
    p = O.op_point(2, w, 1)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    got = np.fromfile(out, np.float32).reshape(ref.shape)
    assert np.array_equal(got, ref)


def test_verbosity_prints_the_reference_timing_lines(capfd):
    """op.verbosity (src/oflow.cpp:246-365, kroeger/oflow.cpp:298-360): 1 -> "TIME (O.Flow Run-Time   ) (ms): ..", 2 -> also one
    "TIME (Sc: l, #p: n, pconst, pinit, poptim, cflow, tvopt, total): .." line per scale, coarsest first, with GPU times of
    the stages; PatGridClass.printTimings (src/patchgrid.cpp:334-345) prints the same numbers; the flow is unchanged"""
    import re
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(272, 480, seed=5)
    op = F.operating_point(2, 480, 1)
    quiet = OFClass(op, F.img_params(width=480, height=272, padding=8)).calc(dev(f0), dev(f1)).cpu().numpy()
    capfd.readouterr()
    op.verbosity = 2
    ofc = OFClass(op, F.img_params(width=480, height=272, padding=8))
    out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
    txt = capfd.readouterr().out
    assert np.array_equal(out, quiet)
    rows = re.findall(r"TIME \(Sc: (\d+), #p:\s*(\d+), pconst, pinit, poptim, cflow, tvopt, total\):\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+) ->\s+([\d.]+) ms\.", txt)
    assert [int(r[0]) for r in rows] == list(range(ofc.op.coarsest_scale, ofc.op.finest_scale - 1, -1)), txt
    assert [int(r[1]) for r in rows] == [g.GetNumPatches() for g in reversed(ofc.grid)]
    for r in rows:
        t = [float(x) for x in r[2:]]
        assert t[0] == 0 and t[1] == 0 and t[2] > 0 and t[4] > 0 and abs(t[2] + t[3] + t[4] - t[5]) < 0.02
    assert re.search(r"TIME \(O\.Flow Run-Time   \) \(ms\): [\d.e+-]+", txt)
    tt = ofc.grid[0].printTimings()
    assert "[optiTime]" in capfd.readouterr().out and tt[2] > 0
    op.verbosity = 1
    ofc1 = OFClass(op, F.img_params(width=480, height=272, padding=8))
    ofc1.calc(dev(f0), dev(f1))
    txt = capfd.readouterr().out
    assert "TIME (O.Flow Run-Time" in txt and "TIME (Sc:" not in txt


def test_natural_images_1080p(natural_images):
    """SURVEY 8(d) inputs for C2 / C3: the reference's images/road_HD.jpg (1920x1080) and a copy rolled by (2, 5) -- op-pt 2 with
    and without refinement against the oracle bit for bit, the recovered flow is the shift; and a 1080p crop of
    images/yosemite_4k.jpg with a sub-pixel warp"""
    F, OFClass, _, O = _mods()
    a = natural_images["road_HD"].astype(np.float32)
    b = np.roll(a, (2, 5), axis=(0, 1))
    for refine in (True, False):
        op = F.operating_point(2, 1920, 1)
        op.use_var_ref = refine
        ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8))
        out = ofc.calc(dev(a), dev(b)).cpu().numpy()
        p = oracle_params(O, op)
        ref = O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0)
        assert np.array_equal(out, ref), refine
        full = ofc.upsample_crop(torch.from_numpy(out[None]).cuda())[0].cpu().numpy()
        med = np.median(full[100:-100, 100:-100].reshape(-1, 2), axis=0)
        assert abs(med[0] - 5) < 0.6 and abs(med[1] - 2) < 0.6, med
        ofc.close()
    y = natural_images["yosemite_4k"][540:1620, 960:2880].astype(np.float32)
    y1 = np.round(0.5 * (np.roll(y, (1, 3), axis=(0, 1)) + np.roll(y, (1, 4), axis=(0, 1))))      # shift (3.5, 1), 8-bit grid
    op = F.operating_point(2, 1920, 1)
    ofc = OFClass(op, F.img_params(width=1920, height=1080, padding=8))
    out = ofc.calc(dev(y), dev(y1)).cpu().numpy()
    p = oracle_params(O, op)
    assert np.array_equal(out, O.flow(O.pad_frame(y, p.sc_f), O.pad_frame(y1, p.sc_f), p, 0))


def test_natural_image_4k_quality_preset(natural_images):
    """BASELINE configs[3] on its named input: images/yosemite_4k.jpg (3840x2160) and a shifted copy, op-pt 4
    (ps 12, stride 3, scales 7..2, 128 LK iterations, refinement on 960x544) -- bit-identical to the oracle"""
    F, OFClass, _, O = _mods()
    a = natural_images["yosemite_4k"].astype(np.float32)
    b = np.roll(a, (3, 7), axis=(0, 1))
    op = F.operating_point(4, 3840, 1)
    ofc = OFClass(op, F.img_params(width=3840, height=2160, padding=12))
    got = ofc.calc(dev(a), dev(b)).cpu().numpy()
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0)
    assert got.shape == (544, 960, 2) and np.array_equal(got, ref)
    full = ofc.upsample_crop(torch.from_numpy(got[None]).cuda())[0].cpu().numpy()
    med = np.median(full[200:-200, 200:-200].reshape(-1, 2), axis=0)
    assert abs(med[0] - 7) < 0.6 and abs(med[1] - 3) < 0.6, med


def test_golden_flo_on_the_coarse_grid(alley, alley_golden_flow):
    """the reference's golden kroeger/flows/alley_0001.flo compared where the engine computes: on the 128x56 finest-scale grid.
    The .flo is the coarse flow x 8, bilinearly upsampled (src coordinate (d + 0.5) / 8 - 0.5) and cropped by 6 rows
    (kroeger/run_dense.cpp:407-414); along each axis the mean of the two full-resolution pixels around a coarse sample k is
    c_k + (c_{k-1} - 2 c_k + c_{k+1}) / 32, a tridiagonal system that is solved for the coarse field (interior samples: the
    crop removes the pixels around the first and last coarse row).  Reported in coarse-grid pixels (the unit the engine's
    output has) and full-resolution pixels (x 8); the residual is the golden file's own (pin 2)."""
    import scipy.linalg
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames("alley", alley)
    op = F.operating_point(2, 1024, 1)
    ofc = OFClass(op, F.img_params(width=1024, height=436, padding=8))
    coarse = ofc.calc(dev(f0), dev(f1)).cpu().numpy()                    # (56, 128, 2), rows of the padded 448-row frame
    g = alley_golden_flow.astype(np.float64)                              # (436, 1024, 2), full-resolution px

    def deinterp(m):                                                      # m[k] = c_k + (c_{k-1} - 2 c_k + c_{k+1}) / 32 along axis 0, replicate ends
        n = m.shape[0]
        ab = np.zeros((3, n)); ab[0, 1:] = 1 / 32; ab[1, :] = 1 - 2 / 32; ab[2, :-1] = 1 / 32
        ab[1, 0] += 1 / 32; ab[1, -1] += 1 / 32
        return scipy.linalg.solve_banded((1, 1), ab, m)
    # columns: coarse sample k sits between full-resolution columns 8k+3 and 8k+4 (no horizontal crop)
    cols = 0.5 * (g[:, 3::8] + g[:, 4::8])                                # (436, 128, 2)
    cols = np.moveaxis(deinterp(np.moveaxis(cols, 1, 0)), 0, 1)
    # rows: coarse row j sits between padded rows 8j+3, 8j+4 = cropped rows 8j-3, 8j-2: j = 1 .. 54
    rows = 0.5 * (cols[5:429:8] + cols[6:430:8])                          # j = 1 .. 53 -> (53, 128, 2)
    assert rows.shape[0] == 53
    gold_coarse = deinterp(rows) / 8.0                                    # (edge rows of this system are approximate: dropped below)
    ours = coarse[1:54].astype(np.float64)
    e = np.sqrt(((ours - gold_coarse)[2:-2] ** 2).sum(-1))
    print("coarse-grid EPE vs alley_0001.flo: mean %.5f px (coarse) = %.4f px full resolution, p99 %.4f, max %.4f (coarse px)"
          % (e.mean(), 8 * e.mean(), np.percentile(e, 99), e.max()))
    assert e.mean() < 0.03 / 8 * 1.5 and np.percentile(e, 99) < 0.2 / 8 * 1.5


def test_level_pipeline_equals_the_launch_per_stage_path(monkeypatch):
    """the default refinement of tall levels -- every inner iteration of the level, data terms included, as ONE pipeline launch
    (varref_levelpipe.hip.h) -- against one launch per stage (FOTG_VR_LEVELPIPE=0): the same bits for a single 4K-class pair at the
    quality preset (six scales, levels of 136 / 272 / 544 rows with 5 / 4 / 3 inner iterations), for batches that make more
    workgroups than the chip holds at once (the ticket order keeps every wait bounded), for RGB and in the tolerance mode; no wait
    timed out; and the pair alone equals the oracle"""
    F, OFClass, _, O = _mods()
    L = F.lib()

    def run(w, h, op_point, frames, n, lp, noc=1, fast=False):
        monkeypatch.setenv("FOTG_VR_LEVELPIPE", lp)
        op = F.operating_point(op_point, w, noc)
        op.grad_descent_iter = min(op.grad_descent_iter, 12)              # (keeps the oracle quick; the refinement is what is under test)
        op.fast_math = fast
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        before = L.fotg_debug_counter(b"level_pipe")
        out = ofc.calc_batch(*frames).clone()
        torch.cuda.synchronize()
        assert L.fotg_ctx_counter(ofc._h, b"tile_timeouts") == 0 and L.fotg_ctx_counter(ofc._h, b"stalls") == 0
        launched = L.fotg_debug_counter(b"level_pipe") - before
        ofc.close()
        return out, launched, op

    for (w, h, op_point, n, noc, fast) in ((2048, 1152, 4, 1, 1, False), (1920, 1080, 3, 12, 1, False), (640, 528, 3, 2, 3, False), (2048, 1152, 4, 1, 1, True)):
        pairs = [synth_pair(h, w, seed=900 + k, noc=noc) for k in range(min(n, 3))]
        f0 = dev(np.stack([pairs[k % len(pairs)][0] for k in range(n)]))
        f1 = dev(np.stack([pairs[k % len(pairs)][1] for k in range(n)]))
        a, la, op = run(w, h, op_point, (f0, f1), n, "1", noc, fast)
        b, lb, _ = run(w, h, op_point, (f0, f1), n, "0", noc, fast)
        assert la > 0 and lb == 0, (w, h, la, lb)
        assert torch.equal(a, b), (w, h, op_point, n, noc, fast)
        if n == 1 and not fast:
            p = oracle_params(O, op)
            ref = O.flow(O.pad_frame(pairs[0][0], p.sc_f), O.pad_frame(pairs[0][1], p.sc_f), p, 0)
            assert np.array_equal(a[0].cpu().numpy(), ref)
