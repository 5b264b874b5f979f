"""ctypes binding of libfotg.so (include/fotg.h).  The HIP library is the product: if it is missing this
module raises -- there is no CPU or PyTorch fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfotg.so")

f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class FotgParams(C.Structure):
    """struct fotg_params (include/fotg.h) == opt_params of src/params.h:23-65 / optparam of kroeger/oflow.h:33-76"""
    _fields_ = [("sc_f", C.c_int), ("sc_l", C.c_int), ("ps", C.c_int), ("max_iter", C.c_int),
                ("min_iter", C.c_int), ("dp_thresh", C.c_float), ("dr_thresh", C.c_float),
                ("res_thresh", C.c_float), ("patove", C.c_float), ("patnorm", C.c_int), ("noc", C.c_int),
                ("usetvref", C.c_int), ("tv_alpha", C.c_float), ("tv_gamma", C.c_float),
                ("tv_delta", C.c_float), ("tv_innerit", C.c_int), ("tv_solverit", C.c_int),
                ("tv_sor", C.c_float), ("sor_mode", C.c_int), ("costfct", C.c_int), ("normoutlier", C.c_float), ("usefbcon", C.c_int),
                ("depth", C.c_int), ("u8_color", C.c_int), ("fast_math", C.c_int)]


# every symbol include/fotg.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("fotg_op_point", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(FotgParams)]),
    ("fotg_padded_size", C.c_int, [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4),
    ("fotg_create", C.c_int, [C.POINTER(FotgParams), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fotg_destroy", None, [vp]),
    ("fotg_calc_batch", C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp]),
    ("fotg_calc_batch_u8", C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp]),
    ("fotg_debug_counter", C.c_long, [C.c_char_p]),
    ("fotg_ctx_counter", C.c_long, [vp, C.c_char_p]),
    ("fotg_set_verbosity", C.c_int, [vp, C.c_int]),
    ("fotg_level_timings", C.c_int, [vp, C.c_int, f32p]),
    ("fotg_calc_sequence", C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    ("fotg_calc_sequence_u8", C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    ("fotg_pipe_create", C.c_int, [C.POINTER(FotgParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fotg_pipe_destroy", None, [vp]),
    ("fotg_pipe_submit", C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_long)]),
    ("fotg_pipe_submit_u8", C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_long)]),
    ("fotg_pipe_submit_ex", C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_long)]),
    ("fotg_pipe_wait", C.c_int, [vp, C.c_long, vp, C.c_int]),
    ("fotg_pipe_sync", C.c_int, [vp]),
    ("fotg_pipe_ticket_event", C.c_int, [vp, C.c_long, C.POINTER(vp)]),
    ("fotg_pipe_context", C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    ("fotg_node_create", C.c_int, [C.POINTER(FotgParams), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fotg_node_destroy", None, [vp]),
    ("fotg_node_shard", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fotg_node_submit", C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_long)]),
    ("fotg_node_submit_u8", C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_long)]),
    ("fotg_node_submit_scatter", C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_long)]),
    ("fotg_node_submit_scatter_u8", C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_long)]),
    ("fotg_node_wait", C.c_int, [vp, C.c_long]),
    ("fotg_node_sync", C.c_int, [vp]),
    ("fotg_node_last_hip_error", C.c_int, [vp]),
    ("fotg_node_info", C.c_int, [vp] + [C.POINTER(C.c_int)] * 4),
    ("fotg_node_pipe", C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    ("fotg_calc", C.c_int, [vp, vp, vp, vp, vp]),
    ("fotg_upsample_crop", C.c_int, [vp, C.c_int, vp, vp, vp]),
    ("fotg_gradient_magnitude", C.c_int, [C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    ("fotg_gradient_magnitude_u8", C.c_int, [C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    ("fotg_level_size", C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fotg_out_size", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fotg_num_patches", C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fotg_pyramid", C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    ("fotg_pyramid_pair", C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp]),
    ("fotg_pyramid_pair_u8", C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp]),
    ("fotg_level_ptr", C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_long)]),
    ("fotg_grid_init", C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_long, vp]),
    ("fotg_grid_set_target", C.c_int, [vp, C.c_int, vp, C.c_long]),
    ("fotg_grid_init_from_coarser", C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    ("fotg_grid_set_camera", C.c_int, [vp, C.c_int, C.c_int]),
    ("fotg_grid_optimize", C.c_int, [vp, C.c_int, C.c_int, vp]),
    ("fotg_grid_aggregate", C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    ("fotg_grid_read", C.c_int, [vp, C.c_int, C.c_int] + [vp] * 7),
    ("fotg_enable_taps", C.c_int, [vp, C.c_int]),
    ("fotg_grid_set_trace", C.c_int, [vp, C.c_int, vp]),
    ("fotg_varref", C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_long, vp, vp]),
    ("fotg_varref_plane", C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, vp]),
    ("fotg_bench_sor_call", C.c_int, [vp, C.c_int, C.c_int, vp]),
    ("fotg_strerror", C.c_char_p, [C.c_int]),
    ("fotg_last_hip_error", C.c_int, []),
    ("fotg_version", C.c_char_p, []),
]

_LIB = None


class FotgError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = LIB_PATH
        exp = os.environ.get("FOTG_EXPERIMENTAL_LIB")
        if exp:
            # A/B timing of experimental builds (tools/exp_*.sh): the variant is NAMED, the product library is never overwritten.  Loud,
            # because such a build may be one that is documented as producing wrong results.
            import sys
            print("flowonthego_amd: FOTG_EXPERIMENTAL_LIB=%s replaces %s in this process" % (exp, LIB_PATH), file=sys.stderr)
            path = exp
        if not os.path.exists(path):
            raise FotgError("libfotg.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C flowonthego_amd/csrc`.  There is no CPU fallback." % path)
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(status):
    """status != 0 -> exception (the C++ shim include/fotg/oflow.h prints and exits here like the reference's
    checkCudaErrors, src/common/cuda_helper.h:286-299)"""
    if status != 0:
        L = lib()
        raise FotgError("fotg: %s (status %d, hip error %d)" % (L.fotg_strerror(status).decode(), status, L.fotg_last_hip_error()))
