"""fast_math (fotg_params::fast_math): the tolerance mode of the patch loop (csrc/lk_fast.hip.h), of the refinement's data term
(csrc/varref_dataterm.inc.h) and of the tall-level solvers' cell update.  NOT bit-identical to the oracle by
design -- the north star's bound is a mean endpoint error of 1e-3 px against the kroeger CPU result.  These tests state the bound
(mean EPE <= 1e-3 px on the full-resolution flow) on BASELINE configs[0]-[3] and print mean / p99 / max; the parity mode
(tests/test_gpu_parity.py, ==) is untouched by the switch."""
import numpy as np
import pytest

from conftest import synth_pair
from test_gpu_parity import _mods, dev, epe, frames, load_fdf, oracle_params

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOL_MEAN = 1e-3          # px, full-resolution flow: BASELINE.json north_star "EPE within 1e-3 of kroeger CPU"


def stats(a, b):
    e = epe(a, b)
    return float(e.mean()), float(np.percentile(e, 99)), float(e.max())


def fast_and_exact(F, OFClass, f0, f1, op_point, noc=1, refine=None, n=1, **kw):
    h, w = f0.shape[1:3] if n > 1 else f0.shape[:2]
    outs = []
    for fast in (False, True):
        op = F.operating_point(op_point, w, noc)
        if refine is not None:
            op.use_var_ref = refine
        for k, v in kw.items():
            setattr(op, k, v)
        op.fast_math = fast
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size), max_batch=n)
        a, b = (dev(f0), dev(f1)) if n > 1 else (dev(f0)[None], dev(f1)[None])
        lo = ofc.calc_batch(a, b)
        outs.append((lo.cpu().numpy(), ofc.upsample_crop(lo).cpu().numpy()))
        ofc.close()
    return outs


def test_fast_math_alley_against_the_oracle(alley, alley_golden_flow):
    """configs[0]: Sintel alley_1, op-pt 2 -- fast mode against the ORACLE itself (full-resolution flow), and the two-sided bound
    against the reference's golden alley_0001.flo still holds"""
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames("alley", alley)
    h, w = f0.shape
    (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, f0, f1, 2)
    p = O.op_point(2, w, 1)
    ref = O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0)
    assert np.array_equal(ex_lo[0], ref)                       # (the parity mode, for reference)
    wp, hp, padw, padh = O.padded_size(w, h, p.sc_f)
    ref_full = O.upsample_crop(ref, p.sc_l, padw, padh, w, h)
    mean, p99, mx = stats(fa_full[0], ref_full)
    print("alley_1 op-pt 2, fast_math vs oracle, full resolution: mean %.3g  p99 %.3g  max %.3g px" % (mean, p99, mx))
    assert mean <= TOL_MEAN
    g_mean, g_p99, g_max = stats(fa_full[0], alley_golden_flow)
    assert 0.0250 <= g_mean <= 0.0270 and 0.165 <= g_p99 <= 0.183, (g_mean, g_p99, g_max)


@pytest.mark.parametrize("refine", [False, True])
def test_fast_math_single_1080p_pair(refine, natural_images):
    """configs[1] (one 1080p pair, no refinement) and the same with refinement: road_HD + a shifted copy, and a synthetic pair"""
    F, OFClass, _, O = _mods()
    a = natural_images["road_HD"].astype(np.float32)
    b = np.roll(a, (2, 5), axis=(0, 1))
    for name, (f0, f1) in (("road_HD", (a, b)), ("synthetic", synth_pair(1080, 1920, seed=1234))):
        (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, f0, f1, 2, refine=refine)
        mean, p99, mx = stats(fa_full[0], ex_full[0])
        print("1080p %s op-pt 2 refine=%s, fast_math vs parity mode (== oracle), full resolution: mean %.3g  p99 %.3g  max %.3g px" % (name, refine, mean, p99, mx))
        assert mean <= TOL_MEAN, name


def test_fast_math_batch_1080p():
    """configs[2] in small: a batch of 8 synthetic 1080p pairs, op-pt 2 + refinement (the eight-lanes-per-patch kernel is chosen by
    the launch size: forced here through a batch that reaches the automatic threshold at level 4)"""
    F, OFClass, _, O = _mods()
    pairs = [synth_pair(1080, 1920, seed=300 + k) for k in range(8)]
    f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, f0, f1, 2, n=8)
    worst = 0.0
    for k in range(8):
        mean, p99, mx = stats(fa_full[k], ex_full[k])
        worst = max(worst, mean)
        assert mean <= TOL_MEAN, k
    print("batch of 8 x 1080p op-pt 2, fast_math vs parity mode: worst mean %.3g px" % worst)
    # every pair of the batch = the same pair alone (independence)
    (_, _), (one_lo, _) = fast_and_exact(F, OFClass, f0[3], f1[3], 2)
    assert np.array_equal(one_lo[0], fa_lo[3])


def test_fast_math_4k_quality_preset(natural_images):
    """configs[3]: yosemite_4k + a shifted copy, op-pt 4 (ps 12, 128 iterations, six scales)"""
    F, OFClass, _, O = _mods()
    a = natural_images["yosemite_4k"].astype(np.float32)
    b = np.roll(a, (3, -4), axis=(0, 1))
    (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, a, b, 4)
    mean, p99, mx = stats(fa_full[0], ex_full[0])
    print("yosemite_4k op-pt 4, fast_math vs parity mode (== oracle), full resolution: mean %.3g  p99 %.3g  max %.3g px" % (mean, p99, mx))
    assert mean <= TOL_MEAN


@pytest.mark.parametrize("case,op_point", [("alley", 2), ("synth_odd", 3), ("alley", 1)])
def test_fast_math_patch_stage_against_the_oracle(case, op_point, alley):
    """the patch stage alone, level by level on the ORACLE's inputs: p_iter and patch weights of the fast kernel against the
    oracle's (no refinement in between: the levels' errors are not damped)"""
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    h, w = f0.shape[:2]
    op = F.operating_point(op_point, w, noc)
    op.use_var_ref = False
    op.fast_math = True
    ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
    p = oracle_params(O, op)
    P0 = O.Pyramid(O.pad_frame(f0, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    P1 = O.Pyramid(O.pad_frame(f1, op.coarsest_scale), op.coarsest_scale, op.patch_size)
    prev_o = None
    for sl in range(op.coarsest_scale, op.finest_scale - 1, -1):
        g = ofc.grid[sl - op.finest_scale]
        lw, lh = P0.level_wh(sl)
        og = O.Grid(lw, lh, sl, p)
        og.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
        g.InitializeGrid(dev(P0.im[sl])[None], dev(P0.dx[sl])[None], dev(P0.dy[sl])[None])
        g.SetTargetImage(dev(P1.im[sl])[None])
        if prev_o is not None:
            og.init_from_coarser(prev_o)
            g.InitializeFromCoarserOF(dev(prev_o)[None])
        og.optimize(P1.im[sl])
        g.Optimize()
        st = g.read_state(0)
        d = np.sqrt(((st["p_iter"] - og.p_iter) ** 2).sum(-1))
        dw = np.abs(st["pweight"] - og.pweight)
        print("%s op-pt %d scale %d: |p - p_oracle| mean %.3g max %.3g px; |w - w_oracle| mean %.3g max %.3g" % (case, op_point, sl, d.mean(), d.max(), dw.mean(), dw.max()))
        assert d.mean() <= 2e-4 and np.percentile(d, 99) <= 2e-3, sl
        assert dw.mean() <= 1e-3, sl
        prev_o = og.aggregate()


def test_fast_math_falls_back_to_the_exact_kernel(alley):
    """configurations the fast patch kernel does not cover run the exact one with fast_math set: other cost functions, early
    termination (min_iter < max_iter) -- the patch stage (refinement off) is bit-identical to the parity mode; with the refinement,
    whose data term and solvers have a tolerance-mode arithmetic of their own, the flows agree within the tolerance"""
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames("alley", alley)
    for kw in ({"cost_func": 1}, {"min_iter": 3}):
        (ex_lo, _), (fa_lo, _) = fast_and_exact(F, OFClass, f0, f1, 2, refine=False, **kw)
        assert np.array_equal(ex_lo, fa_lo), kw
        (_, ex_full), (_, fa_full) = fast_and_exact(F, OFClass, f0, f1, 2, **kw)
        mean, p99, mx = stats(fa_full[0], ex_full[0])
        print("%s with refinement, fast_math vs parity mode: mean %.3g  p99 %.3g  max %.3g px" % (kw, mean, p99, mx))
        assert mean <= TOL_MEAN


@pytest.mark.parametrize("case,op_point,kw", [("alley_rgb", 2, {}), ("synth_rgb", 3, {}), ("alley", 2, {"patch_size": 4, "patch_stride": 0.5}),
                                               ("alley", 2, {"patch_size": 16, "patch_stride": 0.5}), ("alley", 2, {"use_mean_normalization": False}),
                                               ("alley", 2, {"use_fbcon": True})])
def test_fast_math_other_configurations(case, op_point, kw, alley):
    """RGB patches, patch sizes 4 and 16, no mean normalisation, the forward-backward merge"""
    F, OFClass, _, O = _mods()
    f0, f1, noc = frames(case, alley)
    (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, f0, f1, op_point, noc=noc, **kw)
    mean, p99, mx = stats(fa_full[0], ex_full[0])
    print("%s op-pt %d %s, fast_math vs parity mode: mean %.3g  p99 %.3g  max %.3g px" % (case, op_point, kw, mean, p99, mx))
    assert mean <= TOL_MEAN


@pytest.mark.parametrize("case,op_point", [("alley", 2), ("synth_odd", 3)])
def test_fast_math_small_window_and_level_image_paths_agree(case, op_point, alley, monkeypatch):
    """gray 8 x 8 / 12 x 12 patches stage a window of radius 2 around the start; an evaluation outside it reads the level image.
    Both paths deliver the same values: forcing EVERY evaluation through the level image (FOTG_LK_SHW=3 with FOTG_TEST_TAPS=1) gives
    the same bits, and so does the kernel that stages the whole reachable region (FOTG_LK_FAST_R=0)."""
    F, OFClass, _, O = _mods()
    f0, f1, _ = frames(case, alley)
    h, w = f0.shape
    outs = []
    for env in ({}, {"FOTG_TEST_TAPS": "1", "FOTG_LK_SHW": "3"}, {"FOTG_LK_FAST_R": "0"}):
        for k in ("FOTG_TEST_TAPS", "FOTG_LK_SHW", "FOTG_LK_FAST_R"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        op = F.operating_point(op_point, w, 1)
        op.fast_math = True
        ofc = OFClass(op, F.img_params(width=w, height=h, padding=op.patch_size))
        outs.append(ofc.calc(dev(f0), dev(f1)).cpu().numpy())
        ofc.close()
    assert np.array_equal(outs[0], outs[1]), "window path != level-image path"
    assert np.array_equal(outs[0], outs[2]), "radius-2 window != whole-region window"


def test_fast_math_large_motion_leaves_the_small_window():
    """a pair whose patches move further than the staged radius (no coarser level to start from: one scale only, 4 px shift):
    evaluations leave the window for real, same bits as the whole-region kernel, flow within tolerance of the parity mode"""
    import os
    F, OFClass, _, O = _mods()
    f0, f1 = synth_pair(256, 384, seed=31, shift=(4.0, -3.0))
    res = {}
    for r in ("2", "0"):
        os.environ["FOTG_LK_FAST_R"] = r
        try:
            for fast in (True, False):
                op = F.operating_point(2, 384, 1)
                op.coarsest_scale = op.finest_scale = 1
                op.fast_math = fast
                ofc = OFClass(op, F.img_params(width=384, height=256, padding=op.patch_size))
                res[(r, fast)] = ofc.calc(dev(f0), dev(f1)).cpu().numpy()
                ofc.close()
        finally:
            del os.environ["FOTG_LK_FAST_R"]
    assert np.array_equal(res[("2", True)], res[("0", True)])
    assert float(np.abs(res[("0", False)]).mean()) > 0.5                       # the patches really moved
    assert epe(res[("2", True)], res[("0", False)]).mean() <= TOL_MEAN


@pytest.mark.parametrize("noc", [1, 3])
def test_fast_math_refinement_against_the_reference_fdf_vectors(noc):
    """the refinement alone in the tolerance mode (v_rcp / v_rsq in compute_data / compute_smoothness / the block inverse, fused
    multiply-adds in the solvers of levels of more than 64 rows) against the outputs of the reference's own FDF1.0.1 code
    (tests/golden/fdf_ref_*.npz; the parity mode equals them bit for bit, tests/test_gpu_parity.py): the refined flow and the
    right-hand sides of the last inner iteration"""
    F, OFClass, VarRefClass, O = _mods()
    for name, c in load_fdf(noc).items():
        im1, im2, wx, wy, lvl = c["im1"], c["im2"], c["wx"], c["wy"], int(c["lvl"])
        _, h, w = im1.shape
        op = F.operating_point(2, 1024, noc)
        op.coarsest_scale = op.finest_scale = lvl
        op.fast_math = True
        ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8))
        padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((8, 8), (8, 8), (0, 0)), mode="edge")
        flow = dev(np.stack([wx, wy], -1))[None].contiguous()
        F.lib().fotg_enable_taps(ofc._h, 1)
        VarRefClass(dev(padlvl(im1))[None], dev(padlvl(im2))[None], ofc.iparams[0], ofc.op, flow)
        out = flow[0].cpu().numpy()
        ref = np.stack([c["out_x"], c["out_y"]], -1)
        e = np.sqrt(((out - ref) ** 2).sum(-1))
        st = ((w + 3) // 4) * 4
        b1 = np.zeros((1, h, st), np.float32)
        F._lib.check(F.lib().fotg_varref_plane(ofc._h, 0, b"b1", lvl, b1.ctypes.data))
        rel = np.abs(b1[0, :, :w] - c["b1"]).max() / max(1e-30, np.abs(c["b1"]).max())
        print("%s (%d x %d, level %d): refined flow vs the reference's, mean %.3g  max %.3g px (flow magnitude %.3g); b1 max rel. diff %.3g"
              % (name, w, h, lvl, e.mean(), e.max(), np.abs(ref).mean(), rel))
        assert e.mean() <= 1e-4 and e.max() <= 1e-2, name          # (at the level's own resolution)
        assert rel <= 1e-3, name          # (b1 is a difference of terms of opposite sign)


def test_fast_math_random_parameter_sweep():
    """the tolerance mode over random parameter combinations (patch size, overlap, scales, iteration counts, thresholds, mean
    normalisation, refinement weights and solver iterations, gray / RGB): mean endpoint error of the full-resolution flow against
    the parity mode (== the oracle: tests/test_gpu_parity.py::test_random_parameter_sweep) within the north star's bound"""
    F, OFClass, _, O = _mods()
    rng = np.random.default_rng(99)
    worst, done = (0.0, None), 0
    for k in range(40):
        noc = 1 + 2 * int(rng.integers(0, 2))
        w, h = int(rng.integers(200, 640)), int(rng.integers(160, 420))
        kw = dict(patch_size=int(rng.choice([4, 8, 12, 16])), patch_stride=float(rng.choice([0.3, 0.4, 0.5, 0.65, 0.75])),
                  grad_descent_iter=int(rng.integers(4, 33)), use_mean_normalization=bool(rng.random() < 0.8),
                  var_ref_iter=int(rng.integers(1, 5)), var_ref_alpha=float(rng.choice([10.0, 3.0, 25.0])),
                  var_ref_gamma=float(rng.choice([10.0, 0.5, 20.0])), var_ref_delta=float(rng.choice([5.0, 0.0, 12.0])),
                  var_ref_sor_weight=float(rng.choice([1.6, 1.0, 1.9])))
        kw["finest_scale"] = int(rng.integers(0, 3))
        kw["coarsest_scale"] = kw["finest_scale"] + int(rng.integers(1, 3))
        f0, f1 = synth_pair(h, w, seed=700 + k, noc=noc)
        try:
            (ex_lo, ex_full), (fa_lo, fa_full) = fast_and_exact(F, OFClass, f0, f1, 2, noc=noc, refine=bool(rng.random() < 0.8), **kw)
        except F.FotgError:
            continue
        mean, p99, mx = stats(fa_full[0], ex_full[0])
        if mean > worst[0]:
            worst = (mean, (w, h, noc, kw))
        assert mean <= TOL_MEAN, (mean, p99, mx, w, h, noc, kw)
        done += 1
    print("fast_math over %d random parameter combinations: worst mean EPE %.3g px at %s" % (done, worst[0], worst[1]))
    assert done >= 25, done
