"""Parameter structs of the reference's flow API (src/params.h:9-65), same field names and meaning."""
import math
from dataclasses import dataclass

from ._lib import FotgParams, check, lib


@dataclass
class img_params:
    """src/params.h:9-20.  Only width, height and padding are inputs (src/run_dense.cpp:258-263); the bounds and
    padded sizes are derived per scale exactly as src/oflow.cpp:84-95 does."""
    width: int = 0
    height: int = 0
    padding: int = 0
    l_bound: float = 0.0
    u_bound_width: float = 0.0
    u_bound_height: float = 0.0
    width_pad: int = 0
    height_pad: int = 0
    scale_fact: float = 1.0
    curr_lvl: int = 0


@dataclass
class opt_params:
    """src/params.h:23-65 (cublasHandle dropped: the reference never issues a BLAS call, src/oflow.cpp:58,149).
    `channels` is the one addition: 3 = the reference's interleaved RGB input (src/run_dense.cpp:147),
    1 = gray, matching kroeger's run_OF_INT (the parity oracle)."""
    coarsest_scale: int = 5
    finest_scale: int = 3
    patch_size: int = 8
    patch_stride: float = 0.4
    use_mean_normalization: bool = True
    grad_descent_iter: int = 12
    dp_thresh: float = 0.05          # src/oflow.cpp:53 hard-codes 0.05^2, kroeger/run_dense.cpp:227
    dr_thresh: float = 0.95
    res_thresh: float = 0.0
    verbosity: int = 0
    use_var_ref: bool = True
    var_ref_iter: int = 3
    var_ref_alpha: float = 10.0
    var_ref_gamma: float = 10.0
    var_ref_delta: float = 5.0
    var_ref_sor_weight: float = 1.6
    channels: int = 1
    sor_mode: int = 0                # 0 lexicographic (kroeger, parity), 1 red-black (src/ ordering)
    cost_func: int = 0               # kroeger/oflow.h:45: 0 L2, 1 L1, 2 pseudo-Huber (threshold norm_outlier)
    use_fbcon: bool = False          # kroeger/oflow.h:44 usefbcon: forward-backward merge in the densification
    depth_mode: bool = False         # kroeger SELECTMODE=2 (run_DE_*): stereo depth, one displacement channel
    u8_color: int = 0                # channels == 1 only: 8-bit frames arrive with 3 channels (1: B,G,R as cv::imread delivers, 2: R,G,B), gray on load (kroeger/run_dense.cpp:199-209)
    var_ref_inner_iter: int = 1      # kroeger tv_innerit (oflow.h:50, run_dense.cpp:288): inner iterations = var_ref_inner_iter * (level + 1); src/ hard-codes 1
    fast_math: bool = False          # tolerance mode of the patch loop and the refinement's arithmetic (fotg_params::fast_math); False = parity mode
    min_iter: int = -1               # kroeger optparam.min_iter (oflow.h:38); < 0: = grad_descent_iter, as src/ and the operating points have it
    # derived (src/oflow.cpp:45-48)
    outlier_thresh: float = 0.0
    steps: int = 0
    n_vals: int = 0
    n_scales: int = 0
    min_errval: float = 2.0
    norm_outlier: float = 5.0

    def derive(self):
        self.outlier_thresh = self.patch_size / 2.0
        self.steps = max(1, int(math.floor(self.patch_size * (1 - self.patch_stride))))
        self.n_vals = self.channels * self.patch_size ** 2
        self.n_scales = self.coarsest_scale - self.finest_scale + 1
        return self

    def to_c(self):
        p = FotgParams()
        p.sc_f, p.sc_l, p.ps = self.coarsest_scale, self.finest_scale, self.patch_size
        p.max_iter = p.min_iter = self.grad_descent_iter       # src/kernels/optimize.cu:225-229: min == max
        if 0 <= self.min_iter <= self.grad_descent_iter:
            p.min_iter = self.min_iter                          # kroeger: stop early on the dp / residual rate tests (patch.cpp:279-282)
        p.dp_thresh, p.dr_thresh, p.res_thresh = self.dp_thresh, self.dr_thresh, self.res_thresh
        p.patove, p.patnorm, p.noc = self.patch_stride, int(self.use_mean_normalization), self.channels
        p.usetvref = int(self.use_var_ref)
        p.tv_alpha, p.tv_gamma, p.tv_delta = self.var_ref_alpha, self.var_ref_gamma, self.var_ref_delta
        p.tv_innerit, p.tv_solverit, p.tv_sor = int(self.var_ref_inner_iter), self.var_ref_iter, self.var_ref_sor_weight
        p.sor_mode = self.sor_mode
        p.costfct, p.normoutlier, p.usefbcon = self.cost_func, self.norm_outlier, int(self.use_fbcon)
        p.depth = int(self.depth_mode)
        p.u8_color = int(self.u8_color)
        p.fast_math = int(self.fast_math)
        return p


def AutoFirstScaleSelect(imgwidth, fratio, patchsize):
    """src/run_dense.cpp:107-112"""
    scale = (2.0 * imgwidth) / (float(fratio) * float(patchsize))
    return max(0, int(math.floor(math.log2(scale))))


def operating_point(op_point, width_org, channels=1, sor_mode=0):
    """src/run_dense.cpp:168-209 / kroeger/run_dense.cpp:225-268 (evaluated by the library's fotg_op_point)"""
    c = FotgParams()
    check(lib().fotg_op_point(int(op_point), int(width_org), int(channels), c))
    return opt_params(coarsest_scale=c.sc_f, finest_scale=c.sc_l, patch_size=c.ps, patch_stride=round(c.patove, 6),
                      use_mean_normalization=bool(c.patnorm), grad_descent_iter=c.max_iter,
                      dp_thresh=round(c.dp_thresh, 6), dr_thresh=round(c.dr_thresh, 6), res_thresh=c.res_thresh,
                      use_var_ref=bool(c.usetvref), var_ref_iter=c.tv_solverit, var_ref_alpha=c.tv_alpha,
                      var_ref_gamma=c.tv_gamma, var_ref_delta=c.tv_delta, var_ref_sor_weight=round(c.tv_sor, 6),
                      channels=channels, sor_mode=sor_mode).derive()


def padded_size(w, h, coarsest_scale):
    """src/run_dense.cpp:231-237"""
    import ctypes as C
    wp, hp, pw, ph = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib().fotg_padded_size(w, h, coarsest_scale, wp, hp, pw, ph))
    return wp.value, hp.value, pw.value, ph.value
