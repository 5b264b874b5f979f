"""Host-side mirror of the reference's flow orchestrator API (src/oflow.h:22-46, src/patchgrid.h:13-86,
src/refine_variational.h:35-57) over the C-ABI of libfotg.so.  Same class and method names, argument meaning and
ownership as the reference; PyTorch is used only to hold device memory and streams.

    ofc = OFClass(op, iparams)                       # src/run_dense.cpp:277
    ofc.calc(I0, I1, iparams, None, outflow)         # src/run_dense.cpp:286

Numerics are those of the reference's kroeger/ CPU implementation, evaluated by hand-written HIP
kernels; nothing here computes on the CPU and there is no fallback path.
"""
import atexit
import ctypes as C
import sys
import weakref

import torch

from ._lib import FotgError, check, lib
from .params import img_params, opt_params, padded_size


_LIVE = weakref.WeakSet()
# id(opt_params handed out by an OFClass) -> that OFClass: how PatGridClass(_i_params, _op) and VarRefClass(.., _op, ..) -- the
# reference's constructor signatures, src/patchgrid.h:16 and src/refine_variational.h:38-39 -- find their engine context
# (the same registry the C++ shim keeps, include/fotg/patchgrid.h)
_REGISTRY = weakref.WeakValueDictionary()


def _context_of(_op):
    ofc = _REGISTRY.get(id(_op))
    if ofc is None or ofc._h is None:
        raise FotgError("these opt_params do not belong to a live OFClass (use the object's own `ofc.op`, as src/oflow.cpp:101,332 do)")
    return ofc


@atexit.register
def _close_all():
    # destroy contexts while the HIP runtime is still alive (not from __del__ during interpreter shutdown)
    for o in list(_LIVE):
        o.close()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device=None):
    """the caller's current stream ON THE CONTEXT'S DEVICE (not on whatever device happens to be current)"""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_f32(t, name, device=None, shape=None, dtype=torch.float32):
    """The C-ABI takes raw pointers and trusts the sizes: everything a kernel will read or write is checked here --
    dtype, contiguity, device and, where the caller knows it, the exact shape."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise FotgError("%s must be a contiguous %s CUDA(HIP) tensor" % (name, str(dtype).replace("torch.", "")))
    if device is not None and t.device != device:
        raise FotgError("%s lives on %s, the context on %s" % (name, t.device, device))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise FotgError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


class OFClass:
    """src/oflow.h:22-46.  One object = one fixed (size, parameters) configuration, reusable across calc() calls.

    iparams.width/height may be the ORIGINAL frame size: the replicate padding to multiples of 2^coarsest_scale that
    the reference's driver does first (src/run_dense.cpp:231-253) is folded into the pyramid kernel.  Passing already
    padded frames (the reference's calling convention) is the special case pad = 0.
    """

    def __init__(self, _op: opt_params, _i_params: img_params, max_batch: int = 1, device: int = 0):
        self.op = _op.derive()
        self.max_batch = int(max_batch)
        self.width_org, self.height_org = int(_i_params.width), int(_i_params.height)
        self.width, self.height, self.padw, self.padh = padded_size(self.width_org, self.height_org, self.op.coarsest_scale)
        if _i_params.padding not in (0, self.op.patch_size):
            raise FotgError("img_params.padding must equal patch_size (src/run_dense.cpp:263)")
        self.device = torch.device("cuda", device)
        self.nch = 1 if self.op.depth_mode else 2          # flow channels (stereo depth: one displacement, kroeger/oflow.cpp:76-80)
        h = C.c_void_p()
        cp = self.op.to_c()
        check(lib().fotg_create(cp, self.width_org, self.height_org, device, self.max_batch, h))
        self._h = h
        _LIVE.add(self)
        _REGISTRY[id(self.op)] = self
        check(lib().fotg_set_verbosity(self._h, int(self.op.verbosity)))
        # per-scale img_params exactly as src/oflow.cpp:84-95
        self.iparams = []
        ps = self.op.patch_size
        for i in range(self.op.n_scales):
            sl = self.op.finest_scale + i
            w, hh = self.width >> sl, self.height >> sl
            self.iparams.append(img_params(width=w, height=hh, padding=ps, l_bound=-ps / 2.0,
                                           u_bound_width=float(w + ps // 2 - 2), u_bound_height=float(hh + ps // 2 - 2),
                                           width_pad=w + 2 * ps, height_pad=hh + 2 * ps, scale_fact=2.0 ** -sl, curr_lvl=sl))
        self.grid = [PatGridClass(ip, self.op) for ip in self.iparams]        # src/oflow.cpp:101

    # -- geometry -------------------------------------------------------------------------------------------
    def out_size(self):
        w, h = C.c_int(), C.c_int()
        check(lib().fotg_out_size(self._h, w, h))
        return w.value, h.value

    def new_outflow(self, n=1):
        w, h = self.out_size()
        return torch.empty((n, h, w, self.nch), dtype=torch.float32, device=self.device)

    # -- the reference call ---------------------------------------------------------------------------------
    def calc(self, _I0, _I1, _iparams=None, initflow=None, outflow=None):
        """src/oflow.cpp:211-368.  _I0/_I1: device tensors (h, w, channels) or (h, w) float32; outflow: device tensor
        (h/2^finest, w/2^finest, 2), allocated if None.  Returns outflow."""
        single = _I0.dim() == (2 if self.op.channels == 1 else 3)
        I0 = _I0.unsqueeze(0) if single else _I0
        I1 = _I1.unsqueeze(0) if single else _I1
        out = self.calc_batch(I0, I1, initflow, None if outflow is None else (outflow.unsqueeze(0) if single else outflow))
        return out[0] if single else out

    def calc_batch(self, I0, I1, initflow=None, outflow=None):
        """n frame pairs at once: I0, I1 (n, h, w[, channels]); outflow (n, h_l, w_l, 2)"""
        I0, I1 = _dev_f32(I0, "I0", self.device), _dev_f32(I1, "I1", self.device)
        n = I0.shape[0]
        exp = (n, self.height_org, self.width_org) + ((self.op.channels,) if self.op.channels > 1 else ())
        if tuple(I0.shape) != exp and tuple(I0.shape) != exp + (1,):
            raise FotgError("frame shape %s does not match the configured %s" % (tuple(I0.shape), exp))
        if I1.shape != I0.shape:
            raise FotgError("I0 and I1 differ in shape")
        outflow, initflow = self._flow_args(n, outflow, initflow)
        check(lib().fotg_calc_batch(self._h, n, _ptr(I0), _ptr(I1), _ptr(initflow), _ptr(outflow), _stream(self.device)))
        return outflow

    def _flow_args(self, n, outflow, initflow):
        """outflow (n, Hp >> finest, Wp >> finest, nch), allocated if None; initflow None or (n, Hp >> (coarsest+1),
        Wp >> (coarsest+1), nch) -- the sizes fotg_calc_batch reads and writes (include/fotg.h)"""
        if n < 1 or n > self.max_batch:
            raise FotgError("batch of %d pairs, context created for max_batch = %d" % (n, self.max_batch))
        w, h = self.out_size()
        if outflow is None:
            outflow = torch.empty((n, h, w, self.nch), dtype=torch.float32, device=self.device)
        _dev_f32(outflow, "outflow", self.device, (n, h, w, self.nch))
        if initflow is not None:
            sc = self.op.coarsest_scale + 1
            _dev_f32(initflow, "initflow", self.device, (n, self.height >> sc, self.width >> sc, self.nch))
        return outflow, initflow

    def calc_batch_u8(self, I0, I1, initflow=None, outflow=None):
        """n pairs of 8-bit frames (n, h, w[, channels]) uint8 on the device; same result as calc_batch on float frames"""
        for t, nm in ((I0, "I0"), (I1, "I1")):
            _dev_f32(t, nm, self.device, dtype=torch.uint8)
        n = I0.shape[0]
        exp = (n, self.height_org, self.width_org) + self._u8_channels()
        if tuple(I0.shape) != exp or I1.shape != I0.shape:
            raise FotgError("frame shape %s does not match the configured %s" % (tuple(I0.shape), exp))
        outflow, initflow = self._flow_args(n, outflow, initflow)
        check(lib().fotg_calc_batch_u8(self._h, n, _ptr(I0), _ptr(I1), _ptr(initflow), _ptr(outflow), _stream(self.device)))
        return outflow

    def _u8_channels(self):
        """trailing shape of an 8-bit frame: (3,) for colour frames converted to gray on load (op.u8_color)"""
        return (3,) if self.op.u8_color else ((self.op.channels,) if self.op.channels > 1 else ())

    def calc_sequence(self, frames, initflow=None, outflow=None):
        """video mode: frames (n+1, h, w[, channels]) float32 or uint8 on the device -> the n flows frame k -> k+1; every
        frame's pyramid is built once.  Same bits as calc_batch(frames[:-1], frames[1:])"""
        if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.is_contiguous() and frames.dtype in (torch.float32, torch.uint8)):
            raise FotgError("frames must be a contiguous float32 or uint8 CUDA(HIP) tensor")
        if frames.device != self.device:
            raise FotgError("frames live on %s, the context on %s" % (frames.device, self.device))
        n = frames.shape[0] - 1
        exp = (n + 1, self.height_org, self.width_org) + (self._u8_channels() if frames.dtype == torch.uint8 else ((self.op.channels,) if self.op.channels > 1 else ()))
        if n < 1 or tuple(frames.shape) != exp:
            raise FotgError("frame shape %s does not match the configured %s" % (tuple(frames.shape), exp))
        outflow, initflow = self._flow_args(n, outflow, initflow)
        fn = lib().fotg_calc_sequence if frames.dtype == torch.float32 else lib().fotg_calc_sequence_u8
        check(fn(self._h, n + 1, _ptr(frames), _ptr(initflow), _ptr(outflow), _stream(self.device)))
        return outflow

    def upsample_crop(self, flow, out=None):
        """src/run_dense.cpp:293-303: x 2^finest, bilinear upsample, crop the padding -> (n, h_org, w_org, 2)"""
        n = flow.shape[0] if isinstance(flow, torch.Tensor) and flow.dim() == 4 else 0
        if n < 1 or n > self.max_batch:
            raise FotgError("flow must be (n, h_l, w_l, %d) with 1 <= n <= max_batch" % self.nch)
        w, h = self.out_size()
        flow = _dev_f32(flow, "flow", self.device, (n, h, w, self.nch))
        if out is None:
            out = torch.empty((n, self.height_org, self.width_org, self.nch), dtype=torch.float32, device=self.device)
        _dev_f32(out, "out", self.device, (n, self.height_org, self.width_org, self.nch))
        check(lib().fotg_upsample_crop(self._h, n, _ptr(flow), _ptr(out), _stream(self.device)))
        return out

    # -- pyramid (src/oflow.cpp:182-207 ConstructImgPyramids) -----------------------------------------------
    def ConstructImgPyramids(self, I0, I1):
        n = I0.shape[0]
        exp = (n, self.height_org, self.width_org) + ((self.op.channels,) if self.op.channels > 1 else ())
        for t, nm in ((I0, "I0"), (I1, "I1")):
            _dev_f32(t, nm, self.device)
            if tuple(t.shape) != exp and tuple(t.shape) != exp + (1,):
                raise FotgError("%s shape %s does not match the configured %s" % (nm, tuple(t.shape), exp))
        check(lib().fotg_pyramid(self._h, n, _ptr(I0), 0, _stream(self.device)))
        check(lib().fotg_pyramid(self._h, n, _ptr(I1), 1, _stream(self.device)))

    def level(self, which, sl, kind=0, n=1):
        """padded pyramid plane as a tensor VIEW-COPY (n, h+2ps, w+2ps, channels); kind 0 image, 1 dx, 2 dy"""
        p, stride = C.c_void_p(), C.c_long()
        check(lib().fotg_level_ptr(self._h, which, sl, kind, p, stride))
        ip = self.iparams[sl - self.op.finest_scale]
        out = torch.empty((n, ip.height_pad, ip.width_pad, self.op.channels), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize()
        for k in range(n):
            _hip_copy(out[k], p.value + 4 * stride.value * k)
        return out

    def level_ptr(self, which, sl, kind=0):
        p, stride = C.c_void_p(), C.c_long()
        check(lib().fotg_level_ptr(self._h, which, sl, kind, p, stride))
        return p.value, stride.value

    def take_stall(self):
        """for callers of the asynchronous entry points (calc_batch and friends return after enqueueing): AFTER their own
        synchronisation, True = a bounded inter-workgroup wait of this context timed out since the last query (FOTG_ERR_STALL:
        the flows computed since are not valid; re-submit).  Read-and-clear, does not synchronise."""
        return bool(lib().fotg_ctx_counter(self._h, b"take_stall"))

    def synchronize(self):
        """wait for the context's work on the current stream of its device and raise if a stall was flagged"""
        torch.cuda.current_stream(self.device).synchronize()
        if self.take_stall():
            check(5)                                        # FOTG_ERR_STALL

    def close(self):
        """OFClass::~OFClass (src/oflow.cpp:147-179)"""
        h = getattr(self, "_h", None)
        if h:
            self._h = None
            lib().fotg_destroy(h)

    def __del__(self):
        if not sys.is_finalizing():
            try:
                self.close()
            except Exception:
                pass


def gradient_magnitude(frames, coarsest_scale, out=None):
    """The reference's SELECTCHANNEL==2 input (kroeger/run_dense.cpp:138-147): frames (n, h, w[, channels]) float32 or uint8 on
    the device -> (n, Hp, Wp[, channels]) float32, the gradient magnitude of the replicate-padded frames (padding to multiples of
    2^coarsest_scale included).  Feed the result to an OFClass created for Wp x Hp."""
    if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.is_contiguous() and frames.dtype in (torch.float32, torch.uint8)):
        raise FotgError("frames must be a contiguous float32 or uint8 CUDA(HIP) tensor")
    if frames.ndim not in (3, 4) or (frames.ndim == 4 and frames.shape[3] not in (1, 3)):
        raise FotgError("frames must have shape (n, h, w) or (n, h, w, 1|3)")
    n, h, w = frames.shape[:3]
    noc = 1 if frames.ndim == 3 else frames.shape[3]
    from .params import padded_size
    wp, hp = padded_size(w, h, coarsest_scale)[:2]
    shape = (n, hp, wp) + (() if frames.ndim == 3 else (noc,))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=frames.device)
    _dev_f32(out, "out", frames.device, shape)
    fn = lib().fotg_gradient_magnitude if frames.dtype == torch.float32 else lib().fotg_gradient_magnitude_u8
    check(fn(frames.device.index or 0, n, _ptr(frames), w, h, noc, coarsest_scale, _ptr(out), _stream(frames.device)))
    return out


def _hip_copy(dst_tensor, src_ptr):
    """device->device copy of a raw library pointer into a tensor (test/inspection helper)"""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    r = hip.hipMemcpy(C.c_void_p(dst_tensor.data_ptr()), C.c_void_p(src_ptr), dst_tensor.numel() * 4, 3)
    if r != 0:
        raise FotgError("hipMemcpy failed: %d" % r)


class PatGridClass:
    """src/patchgrid.h:13-86: the grid of patches of one scale.  Methods take device tensors in the reference's padded
    level layout (h+2ps, w+2ps, channels) with a leading batch dimension."""

    def __init__(self, _i_params: img_params, _op: opt_params):
        """src/patchgrid.h:16; `_op` is the owning OFClass's `op` (src/oflow.cpp:101 passes &op)"""
        ofc = _context_of(_op)
        self._ofc = ofc
        self.i_params = _i_params
        self.lvl = _i_params.curr_lvl
        a, b = C.c_int(), C.c_int()
        check(lib().fotg_num_patches(ofc._h, self.lvl, a, b))
        self.n_patches_width, self.n_patches_height = a.value, b.value
        self.n_patches = a.value * b.value
        self._n = 1
        self._keep = []

    def GetNumPatches(self): return self.n_patches
    def GetNumPatchesW(self): return self.n_patches_width
    def GetNumPatchesH(self): return self.n_patches_height

    def GetRefPatchPos(self, i):
        """src/patchgrid.cpp:54-63: id = x*n_patches_height + y"""
        steps = self._ofc.op.steps
        offw = (self.i_params.width - (self.n_patches_width - 1) * steps) // 2
        offh = (self.i_params.height - (self.n_patches_height - 1) * steps) // 2
        x, y = divmod(i, self.n_patches_height)
        return (float(x * steps + offw), float(y * steps + offh))

    def InitializeGrid(self, _I0, _I0x, _I0y):
        n = _I0.shape[0] if isinstance(_I0, torch.Tensor) and _I0.dim() == 4 else 0
        if n < 1 or n > self._ofc.max_batch:
            raise FotgError("level images must be (n, h+2ps, w+2ps, channels) with 1 <= n <= max_batch")
        ts = [_dev_f32(t, nm, self._ofc.device, self._lvl_shape(n)) for t, nm in ((_I0, "I0"), (_I0x, "I0x"), (_I0y, "I0y"))]
        self._n = n
        self._keep = ts
        stride = ts[0][0].numel()
        check(lib().fotg_grid_init(self._ofc._h, self.lvl, self._n, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), stride, _stream(self._ofc.device)))

    def _lvl_shape(self, n):
        return (n, self.i_params.height_pad, self.i_params.width_pad, self._ofc.op.channels)

    def SetTargetImage(self, _I1):
        _dev_f32(_I1, "I1", self._ofc.device, self._lvl_shape(self._n))
        self._keep.append(_I1)
        check(lib().fotg_grid_set_target(self._ofc._h, self.lvl, _ptr(_I1), _I1[0].numel()))

    def InitializeFromCoarserOF(self, flow_prev):
        _dev_f32(flow_prev, "flow_prev", self._ofc.device, (self._n, self.i_params.height // 2, self.i_params.width // 2, self._ofc.nch))
        self._keep.append(flow_prev)
        check(lib().fotg_grid_init_from_coarser(self._ofc._h, self.lvl, self._n, _ptr(flow_prev), _stream(self._ofc.device)))

    def SetCamera(self, camlr):
        """depth mode: camparam::camlr of this grid (kroeger/oflow.h:28; 0 left: displacement <= 0, 1 right: >= 0)"""
        check(lib().fotg_grid_set_camera(self._ofc._h, self.lvl, int(camlr)))

    def Optimize(self):
        check(lib().fotg_grid_optimize(self._ofc._h, self.lvl, self._n, _stream(self._ofc.device)))

    def AggregateFlowDense(self, flowout=None):
        if flowout is None:
            flowout = torch.empty((self._n, self.i_params.height, self.i_params.width, self._ofc.nch), dtype=torch.float32, device=self._ofc.device)
        _dev_f32(flowout, "flowout", self._ofc.device, (self._n, self.i_params.height, self.i_params.width, self._ofc.nch))
        check(lib().fotg_grid_aggregate(self._ofc._h, self.lvl, self._n, _ptr(flowout), _stream(self._ofc.device)))
        return flowout

    def printTimings(self):
        """src/patchgrid.cpp:334-345, from the GPU times of the last flow call with op.verbosity > 0 (HIP events of the stages
        on the launch stream): patch extraction and the initialisation from the coarser flow are part of the LK launch"""
        t = (C.c_float * 5)()
        check(lib().fotg_level_timings(self._ofc._h, self.lvl, t))
        print("\n===============Timings (ms)===============")
        print("[extract]      %g\n[coarse]       %g\n[optiTime]      %g\n[aggregate]    %g\n[flow norm]    %g" % (t[0], t[1], t[2], t[3], 0.0))
        print("==========================================")
        return list(t)

    # test taps
    def read_state(self, pair=0, taps=False):
        import numpy as np
        nv = self._ofc.op.n_vals
        p = np.zeros((self.n_patches, 2), np.float32)
        w = np.zeros((self.n_patches, nv), np.float32)
        vp_ = lambda a: a.ctypes.data_as(C.c_void_p)
        if not taps:
            check(lib().fotg_grid_read(self._ofc._h, self.lvl, pair, vp_(p), vp_(w), None, None, None, None, None))
            return {"p_iter": p, "pweight": w}
        t, tx, ty = (np.zeros((self.n_patches, nv), np.float32) for _ in range(3))
        hes = np.zeros((self.n_patches, 3), np.float32)
        cnt = np.zeros((self.n_patches,), np.int32)
        check(lib().fotg_grid_read(self._ofc._h, self.lvl, pair, vp_(p), vp_(w), vp_(t), vp_(tx), vp_(ty), vp_(hes), vp_(cnt)))
        return {"p_iter": p, "pweight": w, "tmpl": t, "tdx": tx, "tdy": ty, "hes": hes, "cnt": cnt}


class VarRefClass:
    """src/refine_variational.h:35-57: like the reference, the constructor does all the work, in place on flowout.
    _I0/_I1: padded level images (n, h+2ps, w+2ps, channels) on the device; flowout (n, h, w, 2) on the device."""

    def __init__(self, _I0, _I1, _i_params: img_params, _op: opt_params, flowout):
        """src/refine_variational.h:38-39; `_op` is the owning OFClass's `op` (src/oflow.cpp:332 passes &op)"""
        ofc = _context_of(_op)
        n = flowout.shape[0] if isinstance(flowout, torch.Tensor) and flowout.dim() == 4 else 0
        if n < 1 or n > ofc.max_batch:
            raise FotgError("flowout must be (n, h, w, %d) with 1 <= n <= max_batch" % ofc.nch)
        lshape = (n, _i_params.height_pad, _i_params.width_pad, ofc.op.channels)
        _dev_f32(_I0, "I0", ofc.device, lshape); _dev_f32(_I1, "I1", ofc.device, lshape)
        _dev_f32(flowout, "flowout", ofc.device, (n, _i_params.height, _i_params.width, ofc.nch))
        check(lib().fotg_varref(ofc._h, _i_params.curr_lvl, n, _ptr(_I0), _ptr(_I1), _I0[0].numel(),
                                _ptr(flowout), _stream(ofc.device)))
        self.flowout = flowout
