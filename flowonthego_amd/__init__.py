"""flowonthego_amd -- MI355X-native Dense-Inverse-Search optical flow.

Host-side mirror of the reference's flow API (src/oflow.h, src/patchgrid.h, src/refine_variational.h, src/params.h)
over the C-ABI of libfotg.so (include/fotg.h), whose kernels are hand-written HIP for gfx950.
"""
from ._lib import FotgError, LIB_PATH, lib  # noqa: F401
from .params import AutoFirstScaleSelect, img_params, opt_params, operating_point, padded_size  # noqa: F401


def __getattr__(name):
    # torch-dependent classes are imported lazily so the CPU-only checks (symbol export, host logic) stay light
    if name in ("OFClass", "PatGridClass", "VarRefClass", "gradient_magnitude"):
        from . import oflow
        return getattr(oflow, name)
    if name == "FlowPipeline":
        from .pipeline import FlowPipeline
        return FlowPipeline
    if name == "FlowNode":
        from .node import FlowNode
        return FlowNode
    raise AttributeError(name)
