"""ctypes front-end of the CPU oracle (oracle/libdis_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (flowonthego_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)


class DisParams(C.Structure):
    """mirror of struct dis_params (oracle/dis_oracle.h) == optparam of kroeger/oflow.h:33-76"""
    _fields_ = [("sc_f", C.c_int), ("sc_l", C.c_int), ("ps", C.c_int), ("max_iter", C.c_int),
                ("min_iter", C.c_int), ("dp_thresh", C.c_float), ("dr_thresh", C.c_float),
                ("res_thresh", C.c_float), ("patove", C.c_float), ("patnorm", C.c_int),
                ("noc", C.c_int), ("usetvref", C.c_int), ("tv_alpha", C.c_float),
                ("tv_gamma", C.c_float), ("tv_delta", C.c_float), ("tv_innerit", C.c_int),
                ("tv_solverit", C.c_int), ("tv_sor", C.c_float),
                ("costfct", C.c_int), ("normoutlier", C.c_float), ("usefbcon", C.c_int), ("depth", C.c_int)]


class DisPyramid(C.Structure):
    _fields_ = [("nlev", C.c_int), ("noc", C.c_int), ("ps", C.c_int), ("w0", C.c_int), ("h0", C.c_int),
                ("im", C.POINTER(f32p)), ("dx", C.POINTER(f32p)), ("dy", C.POINTER(f32p))]


class DisGrid(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("ps", C.c_int), ("noc", C.c_int), ("steps", C.c_int),
                ("nopw", C.c_int), ("noph", C.c_int), ("nop", C.c_int), ("pad", C.c_int),
                ("tmp_w", C.c_int), ("lvl", C.c_int), ("lb", C.c_float), ("ubw", C.c_float),
                ("ubh", C.c_float), ("pt_ref", f32p), ("p_init", f32p), ("tmpl", f32p), ("tdx", f32p),
                ("tdy", f32p), ("hes", f32p), ("p_iter", f32p), ("pweight", f32p),
                ("cnt", C.POINTER(C.c_int)), ("depth", C.c_int), ("camlr", C.c_int)]


def build(force=False):
    """compile oracle/libdis_oracle.so (and oracle/_ref when the reference tree is present)"""
    so = os.path.join(_HERE, "libdis_oracle.so")
    src = os.path.join(_HERE, "dis_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdis_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/kroeger/FDF1.0.1"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_SO_NAME = "libdis_oracle.so"


def use_native():
    """bench.py's cpu_baseline leg only: rebuild the port with -march=native ON the box being timed (`make native`) and switch
    to it.  Same source, same -ffp-contract=off: same bits, only the instruction selection differs.  Returns the flags line."""
    global _LIB, _SO_NAME
    try:
        subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return "gcc -O3 -msse4 -ffp-contract=off (native rebuild failed)"
    _SO_NAME, _LIB = "libdis_oracle_native.so", None
    return "gcc -O3 -msse4 -march=native -ffp-contract=off"


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, _SO_NAME)
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.dis_pyramid_build.restype = C.POINTER(DisPyramid)
        L.dis_grid_new.restype = C.POINTER(DisGrid)
        L.dis_sum.restype = C.c_float
        L.dis_sum.argtypes = [f32p, C.c_int, C.c_int]
        L.dis_compute_smoothness.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_float]
        L.dis_compute_data.argtypes = [f32p] * 16 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        L.dis_sor_coupled.argtypes = [f32p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_float]
        L.dis_sor_coupled_redblack.argtypes = L.dis_sor_coupled.argtypes
        L.dis_sor_coupled_slow.argtypes = L.dis_sor_coupled.argtypes
        _LIB = L
    return _LIB


def P(a):
    return a.ctypes.data_as(f32p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def op_point(op, width_org, noc=1):
    p = DisParams()
    lib().dis_op_point(int(op), int(width_org), int(noc), C.byref(p))
    return p


def padded_size(w, h, sc_f):
    wp, hp, pw, ph = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().dis_padded_size(w, h, sc_f, C.byref(wp), C.byref(hp), C.byref(pw), C.byref(ph))
    return wp.value, hp.value, pw.value, ph.value


def pad_frame(img, sc_f):
    """img: (h, w[, noc]) float32 -> padded (hp, wp[, noc]) (kroeger/run_dense.cpp:298-311)"""
    img = f32(img)
    h, w = img.shape[:2]
    noc = 1 if img.ndim == 2 else img.shape[2]
    wp, hp, _, _ = padded_size(w, h, sc_f)
    out = np.empty((hp, wp) + (() if img.ndim == 2 else (noc,)), np.float32)
    lib().dis_pad_frame(P(img), w, h, noc, sc_f, P(out))
    return out


def gradient_magnitude(img):
    """(h, w[, noc]) float32 padded frame -> its gradient magnitude image (kroeger/run_dense.cpp:138-147, SELECTCHANNEL==2)"""
    img = f32(img)
    h, w = img.shape[:2]
    noc = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    lib().dis_gradient_magnitude(P(img), w, h, noc, P(out))
    return out


class Pyramid:
    """dis_pyramid_build wrapper; levels as numpy copies: im[l], dx[l], dy[l] of shape
    (h_l+2ps, w_l+2ps, noc)"""

    def __init__(self, img, sc_f, ps):
        img = f32(img)
        hp, wp = img.shape[:2]
        self.noc = 1 if img.ndim == 2 else img.shape[2]
        self.ptr = lib().dis_pyramid_build(P(img), wp, hp, self.noc, sc_f, ps)
        self.ps, self.sc_f, self.w0, self.h0 = ps, sc_f, wp, hp
        self.im, self.dx, self.dy = [], [], []
        for l in range(sc_f + 1):
            shp = ((hp >> l) + 2 * ps, (wp >> l) + 2 * ps, self.noc)
            n = shp[0] * shp[1] * shp[2]
            for lst, src in ((self.im, self.ptr.contents.im), (self.dx, self.ptr.contents.dx),
                             (self.dy, self.ptr.contents.dy)):
                lst.append(np.ctypeslib.as_array(src[l], shape=(n,)).reshape(shp).copy())

    def level_wh(self, l):
        return self.w0 >> l, self.h0 >> l

    def __del__(self):
        try:
            lib().dis_pyramid_free(self.ptr)
        except Exception:
            pass


class Grid:
    def __init__(self, w, h, lvl, params, camlr=0):
        self.params = params
        self.ptr = lib().dis_grid_new(w, h, lvl, C.byref(params))
        self.ptr.contents.camlr = camlr
        self.np = 1 if params.depth else 2
        g = self.ptr.contents
        self.nop, self.nopw, self.noph, self.steps = g.nop, g.nopw, g.noph, g.steps
        self.nv = params.ps * params.ps * params.noc
        self.w, self.h = w, h

    def _arr(self, ptr, shape, dtype=np.float32):
        return np.ctypeslib.as_array(ptr, shape=shape)

    def init(self, I0, I0x, I0y):
        self._keep = (f32(I0), f32(I0x), f32(I0y))
        lib().dis_grid_init(self.ptr, C.byref(self.params), *[P(a) for a in self._keep])

    def init_from_coarser(self, flow_prev):
        fp = f32(flow_prev)
        lib().dis_grid_init_from_coarser(self.ptr, P(fp))

    def optimize(self, I1, trace=False):
        I1 = f32(I1)
        tr = None
        if trace:
            tr = np.zeros((self.nop, self.params.max_iter + 1, 4), np.float32)
        lib().dis_grid_optimize(self.ptr, C.byref(self.params), P(I1), P(tr) if trace else None)
        return tr

    def aggregate(self):
        out = np.zeros((self.h, self.w, self.np), np.float32)
        lib().dis_grid_aggregate(self.ptr, C.byref(self.params), P(out))
        return out

    def aggregate_fb(self, cg):
        out = np.zeros((self.h, self.w, self.np), np.float32)
        lib().dis_grid_aggregate_fb(self.ptr, cg.ptr, C.byref(self.params), P(out))
        return out

    @property
    def pt_ref(self): return self._arr(self.ptr.contents.pt_ref, (self.nop, 2)).copy()
    @property
    def p_init(self): return self._arr(self.ptr.contents.p_init, (self.nop, 2)).copy()
    @property
    def p_iter(self): return self._arr(self.ptr.contents.p_iter, (self.nop, 2)).copy()
    @property
    def tmpl(self): return self._arr(self.ptr.contents.tmpl, (self.nop, self.nv)).copy()
    @property
    def tdx(self): return self._arr(self.ptr.contents.tdx, (self.nop, self.nv)).copy()
    @property
    def tdy(self): return self._arr(self.ptr.contents.tdy, (self.nop, self.nv)).copy()
    @property
    def hes(self): return self._arr(self.ptr.contents.hes, (self.nop, 3)).copy()
    @property
    def pweight(self): return self._arr(self.ptr.contents.pweight, (self.nop, self.nv)).copy()
    @property
    def cnt(self): return np.ctypeslib.as_array(self.ptr.contents.cnt, shape=(self.nop,)).copy()

    def __del__(self):
        try:
            lib().dis_grid_free(self.ptr)
        except Exception:
            pass


def varref(I0_lvl, I1_lvl, w, h, lvl, params, flow, sor_mode=0):
    flow = f32(flow).copy()
    a, b = f32(I0_lvl), f32(I1_lvl)
    lib().dis_varref(P(a), P(b), w, h, lvl, C.byref(params), P(flow), int(sor_mode))
    return flow


def varref_depth(I0_lvl, I1_lvl, w, h, lvl, params, flow, camlr=0):
    """RefLevelDE on a (h, w, 1) displacement field"""
    flow = f32(flow).copy()
    a, b = f32(I0_lvl), f32(I1_lvl)
    lib().dis_varref_depth(P(a), P(b), w, h, lvl, C.byref(params), P(flow), int(camlr))
    return flow


def flow_pyr(P0, P1, params, sor_mode=0, dump=False, initflow=None):
    """OFClass ctor on prebuilt pyramids -> finest-scale flow (h_l, w_l, 2) [+ per-level dump list]
    initflow: optional (h/2^(sc_f+1), w/2^(sc_f+1), 2) warm start (kroeger/oflow.cpp:217-220)"""
    w, h = P0.level_wh(params.sc_l)
    nch = 1 if params.depth else 2
    if initflow is not None:
        initflow = f32(initflow)
    out = np.zeros((h, w, nch), np.float32)
    d = None
    if dump:
        tot = sum(2 * nch * (P0.w0 >> l) * (P0.h0 >> l) for l in range(params.sc_l, params.sc_f + 1))
        d = np.zeros(tot, np.float32)
    lib().dis_flow_pyr(P0.ptr, P1.ptr, C.byref(params), P(initflow) if initflow is not None else None, P(out), int(sor_mode),
                       P(d) if dump else None)
    if not dump:
        return out
    lv, off = {}, 0
    for l in range(params.sc_f, params.sc_l - 1, -1):
        n = nch * (P0.w0 >> l) * (P0.h0 >> l)
        shp = (P0.h0 >> l, P0.w0 >> l, nch)
        lv[l] = (d[off:off + n].reshape(shp).copy(), d[off + n:off + 2 * n].reshape(shp).copy())
        off += 2 * n
    return out, lv


def flow(I0p, I1p, params, sor_mode=0):
    """padded frames (hp, wp[, noc]) -> finest-scale flow (pyramid + OFClass)"""
    I0p, I1p = f32(I0p), f32(I1p)
    hp, wp = I0p.shape[:2]
    out = np.zeros((hp >> params.sc_l, wp >> params.sc_l, 1 if params.depth else 2), np.float32)
    lib().dis_flow(P(I0p), P(I1p), wp, hp, C.byref(params), P(out), int(sor_mode))
    return out


def flow_many(I0, I1, params, n_total, nthreads, with_pyramid=True, want_out=False):
    """bench.py's cpu_baseline, all-cores leg (dis_flow_many): I0, I1 = (nsrc, h, w[, noc]) unpadded float32 frames; n_total pairs
    (pair k = source pair k % nsrc) frame-parallel on nthreads pthreads.  Returns (seconds, flows of the first
    min(n_total, nsrc) pairs or None)."""
    I0, I1 = f32(I0), f32(I1)
    nsrc, h, w = I0.shape[:3]
    wp, hp, _, _ = padded_size(w, h, params.sc_f)
    out = None
    if want_out:
        out = np.zeros((min(n_total, nsrc), hp >> params.sc_l, wp >> params.sc_l, 1 if params.depth else 2), np.float32)
    L = lib()
    L.dis_flow_many.restype = C.c_double
    L.dis_flow_many.argtypes = [f32p, f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(DisParams), C.c_int, C.c_int, C.c_int, f32p]
    sec = L.dis_flow_many(P(I0), P(I1), int(I0[0].size), nsrc, w, h, C.byref(params), int(n_total), int(nthreads),
                          1 if with_pyramid else 0, P(out) if out is not None else None)
    return sec, out


def set_sum_order(order):
    """parity-sensitivity switch (tests): 0 = definition D1, 1 sequential, 2 Eigen-style 4-lane packets, 3 pairwise"""
    lib().dis_set_sum_order(int(order))


def set_mean_order(order):
    """parity-sensitivity switch (tests): order of the 2x2 mean's additions, 0 = definition D4"""
    lib().dis_set_mean_order(int(order))


def upsample_crop(fl, sc_l, padw, padh, w_org, h_org):
    fl = f32(fl)
    hl, wl, nch = fl.shape
    out = np.zeros((h_org, w_org, nch), np.float32)
    lib().dis_upsample_crop_n(P(fl), wl, hl, sc_l, padw, padh, w_org, h_org, nch, P(out))
    return out


def full_flow(img0, img1, op=2, sor_mode=0, params=None):
    """run_dense.cpp main(): unpadded frames -> full-resolution flow (h, w, 2)"""
    img0, img1 = f32(img0), f32(img1)
    h, w = img0.shape[:2]
    noc = 1 if img0.ndim == 2 else img0.shape[2]
    p = params or op_point(op, w, noc)
    wp, hp, padw, padh = padded_size(w, h, p.sc_f)
    fl = flow(pad_frame(img0, p.sc_f), pad_frame(img1, p.sc_f), p, sor_mode)
    return upsample_crop(fl, p.sc_l, padw, padh, w, h)


def read_flo(path):
    """Middlebury .flo (kroeger/run_dense.cpp:16-57, flow_code/C/flowIO.cpp:5-19)"""
    with open(path, "rb") as f:
        assert f.read(4) == b"PIEH"
        w, h = np.frombuffer(f.read(8), np.int32)
        return np.frombuffer(f.read(), np.float32).reshape(h, w, 2).copy()
