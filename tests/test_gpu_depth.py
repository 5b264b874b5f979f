"""GPU parity tests of the stereo depth mode (the reference's SELECTMODE=2 build, run_DE_*): one horizontal displacement
per patch / pixel, sign-clamped by camera side; refinement = RefLevelDE.  Same bar as test_gpu_parity.py: bit-for-bit
against the CPU oracle, whose depth restatement is pinned on the reference's own FDF code (tests/test_oracle.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_fdf, synth_pair
from test_gpu_parity import _mods, dev, oracle_params

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def stereo_pair(h, w, seed=5, noc=1, sign=-1.0):
    """rectified synthetic pair: second view = first one displaced horizontally by sign * (4 +- 2) px"""
    f0, _ = synth_pair(h, w, seed=seed, noc=noc)
    f0 = f0.reshape(h, w, noc)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = sign * (4.0 + 2.0 * np.sin(yy / h * 3.0) * np.cos(xx / w * 2.0))
    sx = np.clip(xx - d, 0, w - 1.001)
    x0 = sx.astype(int)
    ax = (sx - x0)[..., None]
    f1 = np.round(f0[yy.astype(int), x0] * (1 - ax) + f0[yy.astype(int), x0 + 1] * ax).astype(np.float32)
    if noc == 1:
        return f0[..., 0], f1[..., 0], d
    return f0, f1, d


def depth_op(F, op_point, w, noc):
    op = F.operating_point(op_point, w, noc)
    op.depth_mode = True
    return op


@pytest.mark.parametrize("noc,op_point,camlr", [(1, 2, 0), (1, 2, 1), (3, 2, 0), (1, 3, 0)])
def test_depth_patchgrid_stages(noc, op_point, camlr):
    """InitializeGrid / InitializeFromCoarserOF / Optimize / AggregateFlowDense per scale with the 1-D parameter:
    scalar Hessian, sign clamp by camera side, one-channel densification -- incl. the per-iteration trace"""
    import ctypes as C
    F, OFClass, _, O = _mods()
    f0, f1, _ = stereo_pair(200, 328, seed=9, noc=noc, sign=-1.0 if camlr == 0 else 1.0)
    h, w = f0.shape[:2]
    op = depth_op(F, op_point, w, noc)
    op.use_var_ref = False
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    F.lib().fotg_enable_taps(ofc._h, 1)
    p = oracle_params(O, op)
    P0 = O.Pyramid(O.pad_frame(f0, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    P1 = O.Pyramid(O.pad_frame(f1, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    prev_o = None
    moved = 0
    for sl in range(op.coarsest_scale, op.finest_scale - 1, -1):
        g = ofc.grid[sl - op.finest_scale]
        lw, lh = P0.level_wh(sl)
        og = O.Grid(lw, lh, sl, p, camlr=camlr)
        og.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
        g.InitializeGrid(dev(P0.im[sl])[None], dev(P0.dx[sl])[None], dev(P0.dy[sl])[None])
        g.SetTargetImage(dev(P1.im[sl])[None])
        g.SetCamera(camlr)
        if prev_o is not None:
            og.init_from_coarser(prev_o)
            g.InitializeFromCoarserOF(dev(prev_o)[None])
        trace = np.zeros((og.nop, op.grad_descent_iter + 1, 4), np.float32)
        F._lib.check(F.lib().fotg_grid_set_trace(ofc._h, sl, trace.ctypes.data_as(C.c_void_p)))
        otrace = og.optimize(P1.im[sl], trace=True)
        g.Optimize()
        F.lib().fotg_grid_set_trace(ofc._h, sl, None)
        st = g.read_state(0, taps=True)
        assert np.array_equal(st["hes"], og.hes) and (og.hes[:, 1:] == 0).all()
        assert np.array_equal(st["cnt"], og.cnt)
        assert np.array_equal(trace, otrace), "per-iteration LK trace differs at scale %d" % sl
        assert np.array_equal(st["p_iter"], og.p_iter) and (og.p_iter[:, 1] == 0).all()
        assert ((og.p_iter[:, 0] <= 0).all() if camlr == 0 else (og.p_iter[:, 0] >= 0).all())
        assert np.array_equal(st["pweight"], og.pweight)
        moved += int((og.p_iter[:, 0] != 0).sum())
        fo = og.aggregate()
        fg = g.AggregateFlowDense()[0].cpu().numpy()
        assert fg.shape == fo.shape == (lh, lw, 1)
        assert np.array_equal(fg, fo), "densified displacement differs at scale %d" % sl
        prev_o = fo
    assert moved > 10


@pytest.mark.parametrize("noc", [1, 3])
def test_depth_varref_golden_reference_vectors(noc):
    """RefLevelDE on the GPU against the outputs of the reference's own compute_data_DE / sor_coupled_slow_but_readable_DE
    chain (tests/golden/fdf_ref_depth_*.npz): system planes of the last inner iteration and the refined displacement"""
    F, OFClass, VarRefClass, O = _mods()
    z = np.load(os.path.join(GOLDEN, "fdf_ref_depth_%s.npz" % ("gray" if noc == 1 else "rgb")))
    for name, c in load_fdf(noc, level4=False).items():
        im1, im2, wx, lvl = c["im1"], c["im2"], c["wx"], int(c["lvl"])
        _, h, w = im1.shape
        op = depth_op(F, 2, 1024, noc)
        op.coarsest_scale = op.finest_scale = lvl
        ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8))
        ps = 8
        padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
        st = ((w + 3) // 4) * 4

        def plane(nm):
            buf = np.zeros((h, st), np.float32)
            F._lib.check(F.lib().fotg_varref_plane(ofc._h, 0, nm.encode(), lvl, buf.ctypes.data))
            return buf[:, :w]
        F.lib().fotg_enable_taps(ofc._h, 1)     # the solver planes of levels refined on chip are only written back for taps
        for camlr in (0, 1):
            w0 = (-np.abs(wx) if camlr == 0 else np.abs(wx)).astype(np.float32)
            flow = dev(w0[..., None])[None].contiguous()
            ofc.grid[0].SetCamera(camlr)
            VarRefClass(dev(padlvl(im1))[None], dev(padlvl(im2))[None], ofc.iparams[0], ofc.op, flow)
            for nm in ("sh", "sv", "a11", "b1", "du"):
                assert np.array_equal(plane(nm), z["%s/%s_de%d" % (name, nm, camlr)]), (name, nm, camlr)
            assert np.array_equal(flow[0, ..., 0].cpu().numpy(), z["%s/out_de%d" % (name, camlr)]), (name, camlr)


@pytest.mark.parametrize("path", ["0", "2"])
@pytest.mark.parametrize("w,h,solverit", [(37, 19, 3), (120, 68, 3), (64, 40, 2), (200, 110, 3), (200, 110, 2), (300, 170, 1), (300, 170, 3),
                                          (20, 400, 3), (60, 400, 3), (24, 1500, 3), (40, 2100, 2)])
def test_depth_varref_sizes(w, h, solverit, path, monkeypatch):
    """level sizes with stride padding; the three solver residencies (everything in LDS up to 8192 cells, du alone in LDS
    up to 128 KiB, global memory beyond: 300x170), one wave group per sweep (up to 341 rows) or all sweeps in each thread
    (the 400-row cases), more rows than a workgroup has threads (1 500 / 2 100: rows looped, one sweep per launch), other sweep counts"""
    F, OFClass, VarRefClass, O = _mods()
    monkeypatch.setenv("FOTG_VR_PATH", path)        # 0: one launch per level where it applies (<= 8192 cells, 3 sweeps); 2: launch per stage
    lvl = 2
    f0, f1, _ = stereo_pair(h, w, seed=w)
    rng = np.random.default_rng(w * 100 + h)
    wx = -np.abs(0.7 + 0.3 * rng.standard_normal((h, w))).astype(np.float32)
    op = depth_op(F, 2, 1024, 1)
    op.coarsest_scale = op.finest_scale = lvl
    op.var_ref_iter = solverit
    n = 2
    ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8), max_batch=n)
    pad = lambda a: np.pad(a, ((8, 8), (8, 8)), mode="edge")[..., None]
    p = oracle_params(O, op)
    ref = O.varref_depth(pad(f0), pad(f1), w, h, lvl, p, wx[..., None], 0)
    flow = dev(np.stack([wx[..., None]] * n))
    VarRefClass(dev(np.stack([pad(f0)] * n)), dev(np.stack([pad(f1)] * n)), ofc.iparams[0], ofc.op, flow)
    got = flow.cpu().numpy()
    assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref)
    assert (ref != wx[..., None]).mean() > 0.5


@pytest.mark.parametrize("noc,op_point,fb", [(1, 2, False), (3, 2, False), (1, 3, False), (1, 2, True), (1, 1, False)])
def test_depth_end_to_end(noc, op_point, fb):
    """OFClass::calc in depth mode == oracle pipeline: finest-scale displacement (one channel), full-resolution upsample,
    batch independence, repeatability; with the forward-backward merge the backward grid is the right camera"""
    F, OFClass, _, O = _mods()
    f0, f1, d = stereo_pair(270, 500, seed=21, noc=noc)
    h, w = f0.shape[:2]
    op = depth_op(F, op_point, w, noc)
    op.use_fbcon = fb
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=3)
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert ref.shape[-1] == 1 and (ref <= 0).all()
    out = ofc.calc(dev(f0), dev(f1))
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()
    full = ofc.upsample_crop(out[None])[0].cpu().numpy()
    wp, hp, padw, padh = O.padded_size(w, h, p.sc_f)
    assert np.array_equal(full, O.upsample_crop(ref, p.sc_l, padw, padh, w, h))
    if op.use_var_ref and not fb:
        assert np.median(np.abs(full[20:-20, 20:-20, 0] - d[20:-20, 20:-20])) < 0.25
    # batch of 3: the pair, the swapped pair (true disparity positive -> clamped), the pair again
    I0 = torch.stack([dev(f0), dev(f1), dev(f0)])
    I1 = torch.stack([dev(f1), dev(f0), dev(f1)])
    gb = ofc.calc_batch(I0, I1).cpu().numpy()
    assert np.array_equal(gb[0], ref) and np.array_equal(gb[2], ref)
    assert np.array_equal(gb[1], O.flow(O.pad_frame(f1, p.sc_f), O.pad_frame(f0, p.sc_f), p, 0))


def test_depth_random_parameter_sweep():
    """the stereo depth mode over random parameter combinations (patch size, overlap, scales, iteration counts, early termination,
    thresholds, mean normalisation, cost function, refinement weights / iterations, gray / RGB, odd sizes): bit-identical to the oracle"""
    F, OFClass, _, O = _mods()
    rng = np.random.default_rng(int(os.environ.get("FOTG_TEST_SWEEP_SEED", "31")))
    done = 0
    for k in range(int(os.environ.get("FOTG_TEST_SWEEP_CASES", "36"))):
        noc = 1 + 2 * int(rng.integers(0, 2))
        w, h = int(rng.integers(200, 520)), int(rng.integers(140, 340))
        tall = bool(os.environ.get("FOTG_TEST_SWEEP_TALL"))
        if tall:
            w, h = int(rng.integers(48, 160)), int(rng.integers(1100, 2600))
        op = depth_op(F, 2, w, noc)
        op.patch_size = int(rng.choice([4, 8, 12, 16]))
        op.patch_stride = float(rng.choice([0.3, 0.4, 0.5, 0.65, 0.75]))
        op.finest_scale = int(rng.integers(0, 3))
        op.coarsest_scale = op.finest_scale + int(rng.integers(0, 3))
        if tall:
            op.finest_scale, op.coarsest_scale = 0, int(rng.integers(0, 3))
        op.grad_descent_iter = int(rng.integers(2, 25))
        op.min_iter = int(rng.integers(0, op.grad_descent_iter + 1)) if rng.random() < 0.5 else -1
        op.dp_thresh, op.dr_thresh = float(rng.choice([0.05, 0.01, 0.2])), float(rng.choice([0.95, 0.8]))
        op.res_thresh = float(rng.choice([0.0, 0.0, 0.5]))
        op.use_mean_normalization = bool(rng.random() < 0.7)
        op.cost_func = int(rng.choice([0, 0, 1, 2]))
        op.use_var_ref = bool(rng.random() < 0.8)
        op.var_ref_iter = int(rng.integers(1, 5))
        op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta = float(rng.choice([10.0, 3.0, 25.0])), float(rng.choice([10.0, 0.5, 20.0])), float(rng.choice([5.0, 0.0, 12.0]))
        op.var_ref_sor_weight = float(rng.choice([1.6, 1.0, 1.9]))
        op.var_ref_inner_iter = int(rng.choice([1, 1, 2, 3]))                   # kroeger tv_innerit (run_dense.cpp:288)
        desc = dict(inner=op.var_ref_inner_iter, w=w, h=h, noc=noc, ps=op.patch_size, ov=op.patch_stride, sc=(op.coarsest_scale, op.finest_scale), it=(op.min_iter, op.grad_descent_iter),
                    thr=(op.dp_thresh, op.dr_thresh, op.res_thresh), norm=op.use_mean_normalization, cost=op.cost_func,
                    ref=(op.use_var_ref, op.var_ref_iter, op.var_ref_alpha, op.var_ref_gamma, op.var_ref_delta, op.var_ref_sor_weight))
        try:
            ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        except F.FotgError:
            continue
        f0, f1, _ = stereo_pair(h, w, seed=300 + k, noc=noc)
        out = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
        pr = oracle_params(O, op)
        ref = O.flow(O.pad_frame(f0, pr.sc_f), O.pad_frame(f1, pr.sc_f), pr, 0)
        assert np.array_equal(out, ref), (desc, float(np.abs(out - ref).max()))
        ofc.close()
        done += 1
    assert done >= 0.6 * int(os.environ.get("FOTG_TEST_SWEEP_CASES", "36")), done


def test_depth_tall_frame_end_to_end():
    """a portrait stereo pair refined at FULL resolution (a level of 1 400 rows: beyond one thread per row) == the oracle"""
    F, OFClass, _, O = _mods()
    w, h = 300, 1400
    f0, f1, _ = stereo_pair(h, w, seed=77)
    op = depth_op(F, 2, w, 1)
    op.finest_scale, op.coarsest_scale, op.grad_descent_iter = 0, 2, 6
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    p = oracle_params(O, op)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(ofc.calc(dev(f0), dev(f1)).cpu().numpy(), ref)
    ofc.close()


def test_depth_unsupported_combinations():
    F, OFClass, _, O = _mods()
    op = depth_op(F, 2, 640, 1)
    op.sor_mode = 1                      # red-black ordering exists only for the coupled optical-flow system
    with pytest.raises(F.FotgError):
        OFClass(op, F.img_params(width=640, height=360, padding=8))


def test_depth_u8_and_sequence_entry_points():
    """the other front-ends of the same path in depth mode: 8-bit frames and the video entry point (one pyramid per frame)"""
    F, OFClass, _, O = _mods()
    f0, f1, _ = stereo_pair(270, 500, seed=33)
    f2 = np.roll(f1, -3, axis=1)
    h, w = f0.shape
    op = depth_op(F, 2, w, 1)
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=2)
    I0 = torch.stack([dev(f0), dev(f1)])
    I1 = torch.stack([dev(f1), dev(f2)])
    ref = ofc.calc_batch(I0, I1).cpu().numpy()
    p = oracle_params(O, op)
    assert np.array_equal(ref[0], O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0))
    assert np.array_equal(ofc.calc_batch_u8(I0.to(torch.uint8), I1.to(torch.uint8)).cpu().numpy(), ref)
    seq = torch.stack([dev(f0), dev(f1), dev(f2)])
    assert np.array_equal(ofc.calc_sequence(seq).cpu().numpy(), ref)
    assert np.array_equal(ofc.calc_sequence(seq.to(torch.uint8)).cpu().numpy(), ref)


def test_cpp_shim_run_dense_example_depth(tmp_path):
    """examples/run_dense_min.cpp with the `depth` switch = the reference's run_DE_* binary: rectified pair in, PFM out
    (SavePFMFile layout), compared bit for bit with the oracle's full-resolution displacement"""
    import subprocess
    from test_host import _build_example
    F, OFClass, _, O = _mods()
    from flowonthego_amd.flo import read_pfm
    exe = _build_example(tmp_path)
    f0, f1, _ = stereo_pair(272, 480, seed=77)
    p0, p1, out = (str(tmp_path / n) for n in ("f0.raw", "f1.raw", "out.pfm"))
    f0.astype(np.float32).tofile(p0)
    f1.astype(np.float32).tofile(p1)
    r = subprocess.run([exe, p0, p1, "480", "272", "1", out, "2", "depth"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    p = O.op_point(2, 480, 1)
    p.depth = 1
    assert np.array_equal(read_pfm(out), O.full_flow(f0, f1, params=p)[..., 0])
