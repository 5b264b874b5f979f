"""ctypes driver of oracle/_ref/libfdf_ref_{gray,rgb}.so -- the reference's OWN FDF1.0.1 C sources
(kroeger/FDF1.0.1/{image,opticalflow_aux,solver}.c) compiled unmodified by `make -C oracle ref`.

TEST INFRASTRUCTURE ONLY.  Used to (a) pin oracle/dis_oracle.c's FDF restatement bit-for-bit and
(b) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).

ref_level_of() sequences the reference functions exactly as VarRefClass::RefLevelOF does
(kroeger/refine_variational.cpp:153-241); every arithmetic step except the trivial uu=wx+du
(:208-214, done with numpy float32 adds) executes reference code.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32p = C.POINTER(C.c_float)


class ImageT(C.Structure):          # FDF1.0.1/image.h:18-24
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("c1", f32p)]


class ColorImageT(C.Structure):     # FDF1.0.1/image.h:27-35
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int),
                ("c1", f32p), ("c2", f32p), ("c3", f32p)]


class ConvT(C.Structure):           # FDF1.0.1/image.h:47-52
    _fields_ = [("order", C.c_int), ("coeffs", f32p), ("coeffs_accu", f32p)]


def available(noc=1):
    return os.path.exists(os.path.join(_HERE, "_ref", "libfdf_ref_%s.so" % ("gray" if noc == 1 else "rgb")))


class FdfRef:
    def __init__(self, noc=1):
        self.noc = noc
        self.L = C.CDLL(os.path.join(_HERE, "_ref", "libfdf_ref_%s.so" % ("gray" if noc == 1 else "rgb")))
        L = self.L
        L.image_new.restype = C.POINTER(ImageT)
        L.color_image_new.restype = C.POINTER(ColorImageT)
        L.convolution_new.restype = C.POINTER(ConvT)
        L.compute_smoothness.argtypes = [C.c_void_p] * 5 + [C.c_float]
        L.compute_data.argtypes = [C.c_void_p] * 20 + [C.c_float] * 3
        L.sor_coupled.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_float]
        L.sor_coupled_slow_but_readable.argtypes = L.sor_coupled.argtypes
        # stereo depth variants (SELECTMODE 2 callers, refine_variational.cpp:243-330); the FDF sources do not depend on
        # SELECTMODE, so the same library carries them
        L.compute_data_DE.argtypes = [C.c_void_p] * 14 + [C.c_float] * 3
        L.sor_coupled_slow_but_readable_DE.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_float]
        d5 = (C.c_float * 3)(0.0, -8.0 / 12.0, 1.0 / 12.0)       # refine_variational.cpp:45-46
        d3 = (C.c_float * 2)(0.0, -0.5)                           # :47-48
        self.deriv = L.convolution_new(2, d5, 0)
        self.deriv_flow = L.convolution_new(1, d3, 0)

    # ---- image helpers ----
    def new(self, w, h, fill=None):
        im = self.L.image_new(w, h)
        a = self.view(im)
        a[...] = 0 if fill is None else fill
        return im

    def newc(self, w, h):
        if self.noc == 1:
            return self.new(w, h)
        im = self.L.color_image_new(w, h)
        self.viewc(im)[...] = 0
        return im

    def view(self, im):
        t = im.contents
        return np.ctypeslib.as_array(t.c1, shape=(t.height, t.stride))

    def viewc(self, im):
        t = im.contents
        if self.noc == 1:
            return np.ctypeslib.as_array(t.c1, shape=(1, t.height, t.stride))
        return np.ctypeslib.as_array(t.c1, shape=(3, t.height, t.stride))

    def from_planar(self, arr):
        """arr (noc, h, w) -> image_t / color_image_t"""
        noc, h, w = arr.shape
        im = self.newc(w, h)
        self.viewc(im)[:, :, :w] = arr
        return im

    def from_plane(self, arr):
        h, w = arr.shape
        im = self.new(w, h)
        self.view(im)[:, :w] = arr
        return im

    def free(self, *ims):
        for im in ims:
            if isinstance(im.contents, ColorImageT):
                self.L.color_image_delete(im)
            else:
                self.L.image_delete(im)

    # ---- kroeger/refine_variational.cpp:153-241 ----
    def ref_level_of(self, im1, im2, wx, wy, lvl, alpha=10.0, gamma=10.0, delta=5.0, innerit=1,
                     solverit=3, omega=1.6, dump=None, slow_solver=False):
        """im1, im2: (noc,h,w) float32 unpadded level images; wx, wy: (h,w).  Returns refined (wx, wy).
        dump: optional dict filled with every intermediate plane of the LAST inner iteration and
        the derivative planes (cropped to w)."""
        L, noc = self.L, self.noc
        _, h, w = im1.shape
        f = np.float32
        qa = f(0.25) * f(alpha)
        hg = f(gamma) * f(0.5) / f(3.0)
        hd = f(delta) * f(0.5) / f(3.0)
        I1, I2 = self.from_planar(im1), self.from_planar(im2)
        WX, WY = self.from_plane(wx), self.from_plane(wy)
        du, dv, mask, sh, sv, uu, vv, a11, a12, a22, b1, b2 = [self.new(w, h) for _ in range(12)]
        w2, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz = [self.newc(w, h) for _ in range(9)]
        L.image_warp(w2, mask, I2, WX, WY)
        L.get_derivatives(I1, w2, self.deriv, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz)
        self.view(uu)[...] = self.view(WX)
        self.view(vv)[...] = self.view(WY)
        inner = innerit * (lvl + 1)
        for it in range(inner):
            L.compute_smoothness(sh, sv, uu, vv, self.deriv_flow, float(qa))
            L.compute_data(a11, a12, a22, b1, b2, mask, WX, WY, du, dv, uu, vv, Ix, Iy, Iz, Ixx, Ixy, Iyy,
                           Ixz, Iyz, float(hd), 0.0, float(hg))
            L.sub_laplacian(b1, WX, sh, sv)
            L.sub_laplacian(b2, WY, sh, sv)
            if dump is not None and it == inner - 1:
                for k, im in (("sh", sh), ("sv", sv), ("a11", a11), ("a12", a12), ("a22", a22),
                              ("b1", b1), ("b2", b2), ("du_in", du), ("dv_in", dv)):
                    dump[k] = self.view(im)[:, :w].copy()
            if slow_solver:
                L.sor_coupled_slow_but_readable(du, dv, a11, a12, a22, b1, b2, sh, sv, solverit, float(omega))
            else:
                L.sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, solverit, float(omega))
            self.view(uu)[...] = self.view(WX) + self.view(du)
            self.view(vv)[...] = self.view(WY) + self.view(dv)
        if dump is not None:
            dump["mask"] = self.view(mask)[:, :w].copy()
            dump["du"] = self.view(du)[:, :w].copy()
            dump["dv"] = self.view(dv)[:, :w].copy()
            for k, im in (("w2", w2), ("Ix", Ix), ("Iy", Iy), ("Iz", Iz), ("Ixx", Ixx), ("Ixy", Ixy),
                          ("Iyy", Iyy), ("Ixz", Ixz), ("Iyz", Iyz)):
                dump[k] = self.viewc(im)[:, :, :w].copy()
        ox, oy = self.view(uu)[:, :w].copy(), self.view(vv)[:, :w].copy()
        self.free(I1, I2, WX, WY, du, dv, mask, sh, sv, uu, vv, a11, a12, a22, b1, b2,
                  w2, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz)
        return ox, oy

    # ---- kroeger/refine_variational.cpp:243-330 (SELECTMODE 2) ----
    def ref_level_de(self, im1, im2, wx, lvl, camlr=0, alpha=10.0, gamma=10.0, delta=5.0, innerit=1,
                     solverit=3, omega=1.6, dump=None):
        """stereo depth refinement of one level: im1, im2 (noc,h,w); wx (h,w) horizontal displacement.  Every arithmetic
        step except the clamped update uu = min/max(wx+du, 0) (:299-314, numpy float32) executes reference code."""
        L, noc = self.L, self.noc
        _, h, w = im1.shape
        f = np.float32
        qa = f(0.25) * f(alpha)
        hg = f(gamma) * f(0.5) / f(3.0)
        hd = f(delta) * f(0.5) / f(3.0)
        I1, I2 = self.from_planar(im1), self.from_planar(im2)
        WX = self.from_plane(wx)
        du, wy0, mask, sh, sv, uu, a11, b1 = [self.new(w, h) for _ in range(8)]
        w2, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz = [self.newc(w, h) for _ in range(9)]
        L.image_warp(w2, mask, I2, WX, wy0)
        L.get_derivatives(I1, w2, self.deriv, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz)
        self.view(uu)[...] = self.view(WX)
        inner = innerit * (lvl + 1)
        for it in range(inner):
            L.compute_smoothness(sh, sv, uu, wy0, self.deriv_flow, float(qa))
            L.compute_data_DE(a11, b1, mask, WX, du, uu, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz, float(hd), 0.0, float(hg))
            L.sub_laplacian(b1, WX, sh, sv)
            if dump is not None and it == inner - 1:
                for k, im in (("sh", sh), ("sv", sv), ("a11", a11), ("b1", b1), ("du_in", du)):
                    dump[k] = self.view(im)[:, :w].copy()
            L.sor_coupled_slow_but_readable_DE(du, a11, b1, sh, sv, solverit, float(omega))
            s = self.view(WX) + self.view(du)
            self.view(uu)[...] = np.where(s < 0, s, f(0)) if camlr == 0 else np.where(s > 0, s, f(0))
        if dump is not None:
            dump["du"] = self.view(du)[:, :w].copy()
        ox = self.view(uu)[:, :w].copy()
        self.free(I1, I2, WX, du, wy0, mask, sh, sv, uu, a11, b1, w2, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz)
        return ox
