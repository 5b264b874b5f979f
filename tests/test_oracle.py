"""CPU tests of the oracle itself: pinned against (1) the reference's golden .flo, (2) golden vectors
produced by the reference's own FDF code (tests/golden/fdf_ref_*.npz), (3) oracle/_ref live when present."""
import numpy as np
import pytest

from conftest import load_fdf, synth_pair
from oracle import oracle as O
from oracle import fdf_ref as R


def epe(a, b):
    return np.sqrt(((a - b) ** 2).sum(-1))


def test_op_points():
    # kroeger/run_dense.cpp:225-268; SURVEY.md section 8 table
    p = O.op_point(2, 1024)
    assert (p.sc_f, p.sc_l, p.ps, p.max_iter, p.usetvref) == (5, 3, 8, 12, 1)
    p = O.op_point(2, 1920)
    assert (p.sc_f, p.sc_l) == (6, 4)
    p = O.op_point(4, 3840)
    assert (p.sc_f, p.sc_l, p.ps, p.max_iter) == (7, 2, 12, 128)
    p = O.op_point(1, 1024)
    assert (p.usetvref, p.max_iter) == (0, 16)
    assert O.padded_size(1920, 1080, 6) == (1920, 1088, 0, 8)
    assert O.padded_size(1024, 436, 5) == (1024, 448, 0, 12)


def test_grid_geometry():
    # patch counts SURVEY.md section 8: 1080p op-2 -> 40/135/510, Sintel -> 32/112/448
    p = O.op_point(2, 1920)
    for (w, h), n in (((30, 17), 40), ((60, 34), 135), ((120, 68), 510)):
        g = O.Grid(w, h, 4, p)
        assert g.nop == n and g.steps == 4
    p = O.op_point(4, 3840)
    g = O.Grid(960, 544, 2, p)
    assert g.nop == 58240 and g.steps == 3


def test_golden_flo(alley, alley_golden_flow):
    """the reference's only golden result: kroeger/flows/alley_0001.flo (run_OF_INT, op-pt 2).
    The survey's build of the unmodified kroeger sources reproduces it to mean 0.026 / p99 0.17 /
    max 0.50 px; the oracle must land in the same place."""
    fl = O.full_flow(alley["frame_0001"].astype(np.float32), alley["frame_0002"].astype(np.float32), op=2)
    e = epe(fl, alley_golden_flow)
    assert fl.shape == alley_golden_flow.shape == (436, 1024, 2)
    # two-sided: the LK / pyramid / densify half of the oracle is pinned to the reference through this file only, at the distance
    # the unmodified kroeger build has from it (mean 0.026 / p99 0.17 / max 0.50 px) -- a drift of the restatement in
    # either direction moves these figures
    assert 0.0255 <= e.mean() <= 0.0264, e.mean()
    assert 0.170 <= np.percentile(e, 99) <= 0.178, np.percentile(e, 99)
    assert 0.49 <= e.max() <= 0.51, e.max()


def _oracle_lk_with_reference_fdf(f0, f1, slow_solver):
    """the oracle's pyramid + LK + densification, with the refinement of every level done by the LIVE reference FDF library
    (oracle/_ref: kroeger/FDF1.0.1 compiled unmodified) sequenced as VarRefClass::RefLevelOF does -> full-resolution flow"""
    p = O.op_point(2, f0.shape[1], 1)
    h, w = f0.shape
    wp, hp, padw, padh = O.padded_size(w, h, p.sc_f)
    P0, P1 = O.Pyramid(O.pad_frame(f0, p.sc_f), p.sc_f, p.ps), O.Pyramid(O.pad_frame(f1, p.sc_f), p.sc_f, p.ps)
    ref = R.FdfRef(1)
    prev = None
    for sl in range(p.sc_f, p.sc_l - 1, -1):
        lw, lh = P0.level_wh(sl)
        g = O.Grid(lw, lh, sl, p)
        g.init(P0.im[sl], P0.dx[sl], P0.dy[sl])
        if prev is not None:
            g.init_from_coarser(prev)
        g.optimize(P1.im[sl])
        fl = g.aggregate()
        crop = lambda a: np.ascontiguousarray(a[p.ps:p.ps + lh, p.ps:p.ps + lw, 0])[None]
        ox, oy = ref.ref_level_of(crop(P0.im[sl]), crop(P1.im[sl]), fl[..., 0].copy(), fl[..., 1].copy(), sl, slow_solver=slow_solver)
        prev = np.stack([ox, oy], -1)
    return O.upsample_crop(prev, p.sc_l, padw, padh, w, h)


@pytest.mark.skipif(not R.available(1), reason="oracle/_ref not built (reference tree absent)")
def test_lk_half_with_live_reference_refinement(alley):
    """a second, independent agreement of the LK half with the real kroeger build: with the live
    reference FDF library refining every level, the oracle's pyramid + LK + densification give O.full_flow bit for bit; and
    with the reference's sor_coupled_slow_but_readable instead, the result moves by 6.2e-3 px -- the value SURVEY 8a's probe of
    the real kroeger build measured for that solver swap"""
    f0, f1 = alley["frame_0001"].astype(np.float32), alley["frame_0002"].astype(np.float32)
    want = O.full_flow(f0, f1, op=2)
    assert np.array_equal(_oracle_lk_with_reference_fdf(f0, f1, False), want)
    slow = _oracle_lk_with_reference_fdf(f0, f1, True)
    d = epe(slow, want).mean()
    assert 5.2e-3 <= d <= 7.2e-3, d
    # the oracle's own restatement of that solver (sor_mode 2, SURVEY row a17') gives the same flow as the reference's
    assert np.array_equal(O.full_flow(f0, f1, op=2, sor_mode=2), slow)


@pytest.mark.parametrize("noc", [1, 3])
def test_fdf_golden_vectors(noc):
    """oracle FDF restatement == outputs of the reference's own FDF1.0.1 C code, bit for bit"""
    L = O.lib()
    f = np.float32
    for name, c in load_fdf(noc).items():
        im1, im2, wx, wy, lvl = c["im1"], c["im2"], c["wx"], c["wy"], int(c["lvl"])
        _, h, w = im1.shape
        st = L.dis_stride(w)

        def pl(a):          # (h,w) or (noc,h,w) -> stride-padded planar
            a = a.reshape(-1, h, w)
            o = np.zeros((a.shape[0], h, st), f)
            o[:, :, :w] = a
            return o
        I1, I2, WX, WY = pl(im1), pl(im2), pl(wx), pl(wy)
        w2, mask = np.zeros_like(I1), np.zeros((1, h, st), f)
        L.dis_image_warp(O.P(w2), O.P(mask), O.P(I2), O.P(WX), O.P(WY), w, h, noc)
        assert np.array_equal(w2[:, :, :w], c["w2"]) and np.array_equal(mask[0, :, :w], c["mask"])
        D = {k: np.zeros_like(I1) for k in ("Ix", "Iy", "Iz", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz")}
        L.dis_get_derivatives(O.P(I1), O.P(w2), w, h, noc, *[O.P(D[k]) for k in ("Ix", "Iy", "Iz", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz")])
        for k in D:
            assert np.array_equal(D[k][:, :, :w], c[k]), (name, k)
        # whole chain through dis_varref on a padded level image
        ps = 8
        p = O.op_point(2, 1024, noc)

        def padlvl(a):
            return np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
        fl = np.stack([wx, wy], -1)
        out = O.varref(padlvl(im1), padlvl(im2), w, h, lvl, p, fl)
        assert np.array_equal(out[..., 0], c["out_x"]) and np.array_equal(out[..., 1], c["out_y"]), name


@pytest.mark.skipif(not R.available(1), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("noc,w,h", [(1, 30, 17), (1, 37, 19), (3, 33, 18), (1, 120, 68)])
def test_varref_vs_live_reference(noc, w, h):
    """random smooth inputs through the live reference FDF library vs the oracle, incl. widths that are
    not multiples of 4 (stride padding) -- bit-exact"""
    rng = np.random.default_rng(w * 100 + h)
    f0, f1 = synth_pair(h, w, seed=w, noc=noc, shift=(0.7, -0.4))
    im1 = f0.reshape(h, w, noc).transpose(2, 0, 1).copy()
    im2 = f1.reshape(h, w, noc).transpose(2, 0, 1).copy()
    wx = (0.7 + 0.3 * rng.standard_normal((h, w))).astype(np.float32)
    wy = (-0.4 + 0.3 * rng.standard_normal((h, w))).astype(np.float32)
    ox, oy = R.FdfRef(noc).ref_level_of(im1, im2, wx, wy, 3)
    p = O.op_point(2, 1024, noc)
    ps = 8
    padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
    out = O.varref(padlvl(im1), padlvl(im2), w, h, 3, p, np.stack([wx, wy], -1))
    assert np.array_equal(out[..., 0], ox) and np.array_equal(out[..., 1], oy)
    # SURVEY row a17': sor_coupled_slow_but_readable (solver.c:19-72), the reference's OpenMP-build solver run serially
    sx, sy = R.FdfRef(noc).ref_level_of(im1, im2, wx, wy, 3, slow_solver=True)
    out = O.varref(padlvl(im1), padlvl(im2), w, h, 3, p, np.stack([wx, wy], -1), sor_mode=2)
    assert np.array_equal(out[..., 0], sx) and np.array_equal(out[..., 1], sy)
    assert not np.array_equal(sx, ox)


def test_synthetic_flow_recovers_shift():
    f0, f1, gt = synth_pair(272, 480, seed=5, truth=True)
    fl = O.full_flow(f0, f1, op=2)
    assert np.median(epe(fl, gt)) < 0.5


def test_patch_cost_functions_numpy_restatement():
    """costfct 1 / 2 (kroeger/patch.cpp:238-261): the oracle's L1 and pseudo-Huber error images, checked through the
    patch weights it reports against a numpy restatement of the two formulas applied to the L2 run's first error image
    (at iteration 0 all three see the same difference image), and through the recovered flow"""
    f0, f1, gt = synth_pair(272, 480, seed=7, truth=True)
    p = O.op_point(2, 480, 1)
    p.max_iter = p.min_iter = 0                    # OptimizeStart only: pweight = |transformed difference image|
    a, b = O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f)
    P0, P1 = O.Pyramid(a, p.sc_f, p.ps), O.Pyramid(b, p.sc_f, p.ps)
    l = p.sc_l
    pw = {}
    for cf in (0, 1, 2):
        p.costfct = cf
        g = O.Grid(*P0.level_wh(l), l, p)
        g.init(P0.im[l], P0.dx[l], P0.dy[l])
        g.optimize(P1.im[l])
        pw[cf] = g.pweight
    d = pw[0]                                      # |d| (sign does not matter for the weights)
    assert np.array_equal(pw[1], np.sqrt(d))
    bsq = np.float32(25.0)
    hub = np.sqrt((np.sqrt(np.float32(1.0) + (d * d) / bsq) - np.float32(1.0)) * (bsq * np.float32(2.0)))
    assert np.array_equal(pw[2], hub.astype(np.float32))
    for cf in (1, 2):
        q = O.op_point(2, 480, 1)
        q.costfct = cf
        fl = O.upsample_crop(O.flow(a, b, q, 0), q.sc_l, *O.padded_size(480, 272, q.sc_f)[2:], 480, 272)
        assert np.median(epe(fl, gt)) < 0.5


def test_forward_backward_merge_python_restatement():
    """usefbcon: the oracle's dis_grid_aggregate_fb against a line-by-line Python restatement of the scatter loops of
    kroeger/patchgrid.cpp:213-275 (own patches) and :278-375 (complementary grid at its moved positions, bilinear taps,
    reversed flow) on a small level, with both grids filled by real LK runs"""
    f0, f1 = synth_pair(96, 160, seed=11)
    p = O.op_point(2, 160, 1)
    p.sc_f = p.sc_l = 1
    a, b = O.pad_frame(f0, 1), O.pad_frame(f1, 1)
    P0, P1 = O.Pyramid(a, 1, p.ps), O.Pyramid(b, 1, p.ps)
    w, h = P0.level_wh(1)
    g, gb = O.Grid(w, h, 1, p), O.Grid(w, h, 1, p)
    g.init(P0.im[1], P0.dx[1], P0.dy[1]); g.optimize(P1.im[1])
    gb.init(P1.im[1], P1.dx[1], P1.dy[1]); gb.optimize(P0.im[1])
    got = np.zeros((h, w, 2), np.float32)
    O.lib().dis_grid_aggregate_fb(g.ptr, gb.ptr, O.C.byref(p), O.P(got))
    f = np.float32
    flow, we = np.zeros((h, w, 2), f), np.zeros((h, w), f)
    ps, lb, ub = p.ps, -p.ps // 2, p.ps // 2 - 1
    for ip in range(g.nop):                                           # :223-272
        fl, pw, ref = g.p_iter[ip], g.pweight[ip].reshape(ps, ps), g.pt_ref[ip]
        for y in range(lb, ub + 1):
            for x in range(lb, ub + 1):
                yt, xt = int(y + ref[1]), int(x + ref[0])
                if 0 <= xt < w and 0 <= yt < h:
                    absw = f(1.0) / max(f(2.0), pw[y - lb, x - lb])
                    we[yt, xt] += absw
                    flow[yt, xt] += fl * absw
    for ip in range(gb.nop):                                          # :286-374
        fl, pw = gb.p_iter[ip], gb.pweight[ip].reshape(ps, ps)
        rp = gb.pt_ref[ip] + fl
        pos = (int(np.ceil(float(rp[0]) + .00001)), int(np.ceil(float(rp[1]) + .00001)), int(np.floor(rp[0])), int(np.floor(rp[1])))
        r0, r1 = rp[0] - f(pos[2]), rp[1] - f(pos[3])
        wb = (r0 * r1, (f(1) - r0) * r1, r0 * (f(1) - r1), (f(1) - r0) * (f(1) - r1))
        for y in range(lb, ub + 1):
            for x in range(lb, ub + 1):
                yt, xt = y + pos[1], x + pos[0]
                if xt >= 1 and yt >= 1 and xt < w - 1 and yt < h - 1:
                    absw = f(1.0) / max(f(2.0), pw[y - lb, x - lb])
                    fn = fl * absw
                    for k, (yy, xx) in enumerate(((yt, xt), (yt, xt - 1), (yt - 1, xt), (yt - 1, xt - 1))):
                        we[yy, xx] += wb[k] * absw
                        flow[yy, xx] -= wb[k] * fn
    nz = we > 0
    flow[nz] /= we[nz][:, None]
    assert np.array_equal(got, flow)
    plain = g.aggregate()
    assert not np.array_equal(got, plain) and np.median(epe(got, plain)) < 0.3


def test_forward_backward_merge_recovers_flow():
    f0, f1, gt = synth_pair(272, 480, seed=5, truth=True)
    p = O.op_point(2, 480, 1)
    p.usefbcon = 1
    fl = O.upsample_crop(O.flow(O.pad_frame(f0, p.sc_f), O.pad_frame(f1, p.sc_f), p, 0), p.sc_l, *O.padded_size(480, 272, p.sc_f)[2:], 480, 272)
    assert np.median(epe(fl, gt)) < 0.5


def test_redblack_is_not_reference(alley):
    a, b = alley["frame_0001"].astype(np.float32), alley["frame_0002"].astype(np.float32)
    e = epe(O.full_flow(a, b, op=2), O.full_flow(a, b, op=2, sor_mode=1))
    assert 0.01 < e.mean() < 0.2      # red-black moves the result by ~0.068 px


def test_dis_sum_order():
    """definition D1: 16 partial sums (pixel q -> partial q % 16, ascending), then the balanced tree xor 8, 4, 2, 1"""
    for n, noc in ((64, 1), (144, 1), (432, 3), (16, 1)):
        v = (np.random.default_rng(3 + n).standard_normal(n) * 100).astype(np.float32)
        part = np.zeros(16, np.float32)
        used = np.zeros(16, bool)
        for e in range(n):
            l = (e // noc) % 16
            part[l] = part[l] + v[e] if used[l] else v[e]
            used[l] = True
        k = 8
        while k >= 1:
            part = part + part[np.arange(16) ^ k]
            k >>= 1
        assert O.lib().dis_sum(O.P(v), n, noc) == part[0]


# ---------------------------------------------------------------------------------------------------------------
# stereo depth (SELECTMODE 2)
# ---------------------------------------------------------------------------------------------------------------
def depth_params(op, width, noc=1):
    p = O.op_point(op, width, noc)
    p.depth = 1
    return p


def _padlvl(a, ps=8):
    return np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")


@pytest.mark.parametrize("noc", [1, 3])
def test_depth_fdf_golden_vectors(noc):
    """oracle RefLevelDE restatement == outputs of the reference's own compute_data_DE / sor_coupled_slow_but_readable_DE
    chain (tests/golden/fdf_ref_depth_*.npz, made by tests/golden/make_golden.py), bit for bit, both camera sides"""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "fdf_ref_depth_%s.npz" % ("gray" if noc == 1 else "rgb")))
    p = depth_params(2, 1024, noc)
    for name, c in load_fdf(noc, level4=False).items():
        im1, im2, wx, lvl = c["im1"], c["im2"], c["wx"], int(c["lvl"])
        _, h, w = im1.shape
        for camlr in (0, 1):
            w0 = (-np.abs(wx) if camlr == 0 else np.abs(wx)).astype(np.float32)
            out = O.varref_depth(_padlvl(im1), _padlvl(im2), w, h, lvl, p, w0[..., None], camlr)
            assert np.array_equal(out[..., 0], z["%s/out_de%d" % (name, camlr)]), (name, camlr)
            assert (out <= 0).all() if camlr == 0 else (out >= 0).all()


@pytest.mark.skipif(not R.available(1), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("noc,w,h,camlr", [(1, 30, 17, 0), (1, 37, 19, 1), (3, 33, 18, 0), (1, 120, 68, 0)])
def test_depth_varref_vs_live_reference(noc, w, h, camlr):
    rng = np.random.default_rng(w * 100 + h)
    f0, f1 = synth_pair(h, w, seed=w, noc=noc, shift=(-0.7 if camlr == 0 else 0.7, 0.0))
    im1 = f0.reshape(h, w, noc).transpose(2, 0, 1).copy()
    im2 = f1.reshape(h, w, noc).transpose(2, 0, 1).copy()
    wx = (0.7 + 0.3 * rng.standard_normal((h, w))).astype(np.float32)
    wx = -np.abs(wx) if camlr == 0 else np.abs(wx)
    dump = {}
    ox = R.FdfRef(noc).ref_level_de(im1, im2, wx, 3, camlr=camlr, dump=dump)
    p = depth_params(2, 1024, noc)
    out = O.varref_depth(_padlvl(im1), _padlvl(im2), w, h, 3, p, wx[..., None], camlr)
    assert np.array_equal(out[..., 0], ox)


def _d1_sum(v, noc=1):
    """definition D1 in numpy: 16 partial sums (pixel q -> partial q % 16, ascending), then the tree xor 8, 4, 2, 1"""
    v = np.asarray(v, np.float32)
    part = np.zeros(16, np.float32)
    used = np.zeros(16, bool)
    for e in range(len(v)):
        l = (e // noc) % 16
        part[l] = part[l] + v[e] if used[l] else v[e]
        used[l] = True
    k = 8
    while k >= 1:
        part = part + part[np.arange(16) ^ k]
        k >>= 1
    return part[0]


@pytest.mark.parametrize("camlr", [0, 1])
def test_depth_lk_numpy_restatement(camlr):
    """the depth-specific pieces of the patch optimisation (kroeger/patch.cpp, SELECTMODE 2), written again in numpy from
    the reference text and compared with the oracle bit for bit after ONE iteration from an integer start position:
    scalar Hessian sum(Tx^2) with the 1e-10 guard (:83-87), one projection sum(Tx * pdiff) (:181), the 1x1 LLT solve
    (L = sqrt(H), y = d / L, x = y / L, :184), p -= x (:186), the sign clamp by camera side (:188-193), y untouched
    (:218-220), reset to p_in when the patch leaves the valid region or moves more than ps/2 (:199-208).  The template,
    its gradient and the query patch at an integer position are plain pixel reads (shared with the flow mode, pinned
    through the golden .flo)"""
    h, w, ps = 136, 240, 8
    f0, f1 = synth_pair(h, w, seed=21)
    f1 = np.roll(f0, -1 if camlr == 0 else 1, axis=1)              # a one-pixel disparity of the admissible sign
    p = depth_params(2, w)
    p.sc_f = p.sc_l = 0
    p.max_iter = p.min_iter = 1
    P0, P1 = O.Pyramid(O.pad_frame(f0, 0), 0, ps), O.Pyramid(O.pad_frame(f1, 0), 0, ps)
    g = O.Grid(*P0.level_wh(0), 0, p, camlr=camlr)
    g.init(P0.im[0], P0.dx[0], P0.dy[0])
    tr = g.optimize(P1.im[0], trace=True)
    I1 = P1.im[0][..., 0]
    lb, ubw, ubh = -ps / 2.0, float(w + ps // 2 - 2), float(h + ps // 2 - 2)
    T, Tx, ref = g.tmpl, g.tdx, g.pt_ref
    nv = ps * ps
    checked = moved = 0
    for ip in range(0, g.nop, 7):
        rx, ry = float(ref[ip, 0]), float(ref[ip, 1])
        if rx < lb or ry < lb or rx > ubw or ry > ubh:
            continue                                                 # never started (oracle definition D2)
        H = _d1_sum(Tx[ip] * Tx[ip])
        if H == 0:
            H = np.float32(np.float64(H) + 1e-10)
        x0, y0 = int(rx) + ps - ps // 2, int(ry) + ps - ps // 2     # padded coordinates of the patch's first pixel
        q = I1[y0:y0 + ps, x0:x0 + ps].reshape(-1).astype(np.float32)
        q = q - np.float32(_d1_sum(q) / np.float32(nv))             # mean normalisation (patch.cpp:330-331 / :397-398)
        pdiff = q - T[ip]
        d = _d1_sum(Tx[ip] * pdiff)
        L = np.sqrt(np.float32(H))
        x = np.float32(np.float32(d / L) / L)
        pn = np.float32(0.0) - x
        pn = min(pn, np.float32(0.0)) if camlr == 0 else max(pn, np.float32(0.0))
        px = np.float32(rx) + pn
        if abs(float(np.float32(rx) - px)) > ps / 2 or px < lb or px > ubw:
            pn = np.float32(0.0)                                     # reset to p_in
        assert tr[ip, 1, 0] == pn and tr[ip, 1, 1] == 0 and g.p_iter[ip, 0] == pn, ip
        checked += 1
        moved += pn != 0
    assert checked > 50 and moved > 20


def test_depth_recovers_disparity():
    """whole depth pipeline on a synthetic rectified pair: the second view is the first shifted left, so the disparity is
    negative (left camera, camlr 0) and the clamp p <= 0 never binds at the solution"""
    h, w = 272, 480
    f0, _ = synth_pair(h, w, seed=5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = -(4.0 + 2.0 * np.sin(yy / h * 3.0) * np.cos(xx / w * 2.0))       # disparity field (<= -2)
    sx = np.clip(xx - d, 0, w - 1.001)
    x0 = sx.astype(int)
    ax = sx - x0
    f1 = np.round(f0[yy.astype(int), x0] * (1 - ax) + f0[yy.astype(int), x0 + 1] * ax).astype(np.float32)
    p = depth_params(2, w)
    fl = O.full_flow(f0, f1, params=p)
    assert fl.shape == (h, w, 1) and (fl <= 0).all()
    err = np.abs(fl[20:-20, 20:-20, 0] - d[20:-20, 20:-20])
    assert np.median(err) < 0.15 and err.mean() < 0.4
    # forward-backward merge variant runs and agrees roughly
    p.usefbcon = 1
    fl2 = O.full_flow(f0, f1, params=p)
    assert np.abs(fl2 - fl).mean() < 0.3


def test_reduction_order_does_not_move_the_flow(alley, natural_images):
    """What the reference leaves to Eigen (order of the per-patch sums, patch.cpp:74-77,178-179,278,330-331) and to OpenCV
    (order of the additions of the 2x2 mean, run_dense.cpp:150) is a DEFINITION in the oracle (D1, D4 in dis_oracle.c).  The
    claim "within 1e-3 px of kroeger" needs those definitions not to matter: run the whole pipeline (op-pt 2, refinement on)
    with four summation orders -- the oracle's D1, a scalar loop, Eigen-style 4-float packet accumulators, pairwise -- and four
    2x2-mean orders, and bound the movement of the full-resolution flow between ANY two of them.  Measured:
    sums: mean <= 1.1e-5 px, max <= 7.1e-4 px; 2x2 mean on non-integer input: mean <= 1.2e-5, p99 <= 1.3e-4, max 1.8e-3 px at
    one pixel of road_HD; on 8-bit valued input (every reference input) the 2x2-mean orders give identical bits."""
    import itertools
    road = natural_images["road_HD"].astype(np.float32)
    cases = {"alley_1": (alley["frame_0001"].astype(np.float32), alley["frame_0002"].astype(np.float32)),
             "road_HD": (road, np.roll(road, (2, 5), (0, 1)))}
    try:
        for name, (f0, f1) in cases.items():
            res = []
            for o in range(4):
                O.set_sum_order(o)
                res.append(O.full_flow(f0, f1, 2))
            O.set_sum_order(0)
            assert not np.array_equal(res[0], res[1]), "the switch must actually change the order"
            for i, j in itertools.combinations(range(4), 2):
                e = epe(res[i], res[j])
                assert e.mean() <= 1e-4 and e.max() <= 1e-3, (name, "sum order", i, j, e.mean(), e.max())
            # 2x2 mean: exact on 8-bit valued frames whatever the order ...
            ref = res[0]
            for o in range(1, 4):
                O.set_mean_order(o)
                assert np.array_equal(O.full_flow(f0, f1, 2), ref), (name, "2x2 mean order on integer input", o)
            # ... and a rounding-level effect on non-integer frames
            g0, g1 = (f0 * 0.731 + 0.123).astype(np.float32), (f1 * 0.731 + 0.123).astype(np.float32)
            res = []
            for o in range(4):
                O.set_mean_order(o)
                res.append(O.full_flow(g0, g1, 2))
            O.set_mean_order(0)
            assert not np.array_equal(res[0], res[1])
            for i, j in itertools.combinations(range(4), 2):
                e = epe(res[i], res[j])
                assert e.mean() <= 1e-4 and np.percentile(e, 99) <= 1e-3 and e.max() <= 5e-3, (name, "2x2 mean order", i, j, e.mean(), e.max())
    finally:
        O.set_sum_order(0)
        O.set_mean_order(0)


def test_flow_many_threads_equal_single_calls():
    """dis_flow_many (bench.py's all-cores cpu_baseline leg: pthreads, thread-private block caches) returns the bits of dis_flow"""
    prs = [synth_pair(270, 480, seed=300 + k) for k in range(3)]
    I0, I1 = np.stack([p[0] for p in prs]), np.stack([p[1] for p in prs])
    p = O.op_point(2, 480)
    ref = [O.flow(O.pad_frame(a, p.sc_f), O.pad_frame(b, p.sc_f), p, 0) for a, b in prs]
    for with_pyr in (True, False):
        sec, out = O.flow_many(I0, I1, p, 7, 3, with_pyr, True)
        assert sec > 0 and out.shape[0] == 3
        for k in range(3):
            assert np.array_equal(out[k], ref[k]), (with_pyr, k)


def test_gradient_magnitude_input_restatement():
    """SELECTCHANNEL==2 front end (kroeger/run_dense.cpp:138-147): the oracle's gradient magnitude against an independent numpy
    restatement of cv::Sobel(ksize 1, BORDER_DEFAULT) + mul + add + cv::sqrt, gray and RGB, odd sizes; a constant frame gives 0"""
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    for shape in ((7, 9), (2, 2), (33, 18, 3), (5, 64, 3)):
        a = (rng.random(shape) * 255).astype(np.float32)
        pw = ((1, 1), (1, 1)) + (((0, 0),) if a.ndim == 3 else ())
        p = np.pad(a, pw, mode="reflect")                       # REFLECT_101: -1 -> 1, n -> n-2
        dx = p[1:-1, 2:] - p[1:-1, :-2]
        dy = p[2:, 1:-1] - p[:-2, 1:-1]
        ref = np.sqrt(dx * dx + dy * dy, dtype=np.float32)
        assert np.array_equal(O.gradient_magnitude(a), ref), shape
    assert not O.gradient_magnitude(np.full((6, 5), 77.0, np.float32)).any()
