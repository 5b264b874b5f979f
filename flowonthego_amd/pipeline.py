"""FlowPipeline: several batches in flight (fotg_pipe_* of include/fotg.h).

The reference's OFClass::calc is synchronous, one pair at a time.  The engine's batches are latency-bound at batch 64 (the
refinement keeps a quarter of the CUs busy) and independent of each other, so a pipe runs `depth` engine contexts on
internal streams and hands consecutive batches to them in turn: submit() returns at once, up to `depth` batches overlap.
Results are bit-identical to OFClass.calc_batch."""
import ctypes as C

import torch

from ._lib import FotgError, check, lib
from .oflow import _dev_f32, _ptr, _stream
from .params import img_params, opt_params, padded_size


class FlowPipeline:
    def __init__(self, _op: opt_params, _i_params: img_params, max_batch: int = 1, depth: int = 3, device: int = 0):
        self.op = _op.derive()
        self.max_batch, self.depth = int(max_batch), int(depth)
        self.width_org, self.height_org = int(_i_params.width), int(_i_params.height)
        self.width, self.height, _, _ = padded_size(self.width_org, self.height_org, self.op.coarsest_scale)
        if _i_params.padding not in (0, self.op.patch_size):
            raise FotgError("img_params.padding must equal patch_size (src/run_dense.cpp:263)")
        self.device = torch.device("cuda", device)
        self.nch = 1 if self.op.depth_mode else 2
        h = C.c_void_p()
        check(lib().fotg_pipe_create(self.op.to_c(), self.width_org, self.height_org, device, self.max_batch, self.depth, h))
        self._h = h

    def out_size(self):
        return self.width >> self.op.finest_scale, self.height >> self.op.finest_scale

    def new_outflow(self, n=1):
        w, h = self.out_size()
        return torch.empty((n, h, w, self.nch), dtype=torch.float32, device=self.device)

    def submit(self, I0, I1, initflow=None, outflow=None, after_current_stream=True, no_recompute=False):
        """enqueue one batch (n, h, w[, channels]) float32 or uint8 behind the current torch stream's work; returns
        (ticket, outflow).  I0, I1 and outflow must stay alive and untouched until wait(ticket) / synchronize().
        Recompute contract (include/fotg.h): a host wait that finds a stall flag recomputes only batches that have NOT been handed
        to a stream through wait(ticket) -- after wait(ticket) the waiting stream owns the result and the tensors may be freed or
        reused; such a batch is reported (FotgError: FOTG_ERR_STALL) by wait(ticket, host=True) / synchronize(), or through
        take_stalls().  no_recompute=True says at submit that the buffers will not stay in place.
        after_current_stream=False starts at once: the caller then guarantees that nothing still enqueued on a torch stream
        writes the frames or touches `outflow` -- including earlier users of memory that torch's caching allocator has
        recycled into these tensors (the pipe's streams are not torch's; synchronize once after allocating the buffers).
        An outflow allocated here (outflow=None) always waits for the current stream for that reason."""
        u8 = I0.dtype == torch.uint8
        for t, nm in ((I0, "I0"), (I1, "I1")):
            _dev_f32(t, nm, self.device, dtype=torch.uint8 if u8 else torch.float32)
        n = I0.shape[0]
        exp = (n, self.height_org, self.width_org) + ((3,) if u8 and self.op.u8_color else (self.op.channels,) if self.op.channels > 1 else ())
        if (tuple(I0.shape) != exp and tuple(I0.shape) != exp + (1,)) or I1.shape != I0.shape:
            raise FotgError("frame shape %s does not match the configured %s" % (tuple(I0.shape), exp))
        if n < 1 or n > self.max_batch:
            raise FotgError("batch of %d pairs, pipe created for max_batch = %d" % (n, self.max_batch))
        w, h = self.out_size()
        if outflow is None:
            outflow = torch.empty((n, h, w, self.nch), dtype=torch.float32, device=self.device)
            after_current_stream = True
        _dev_f32(outflow, "outflow", self.device, (n, h, w, self.nch))
        if initflow is not None:
            sc = self.op.coarsest_scale + 1
            _dev_f32(initflow, "initflow", self.device, (n, self.height >> sc, self.width >> sc, self.nch))
        ticket = C.c_long()
        after = _stream(self.device) if after_current_stream else C.c_void_p(-1)      # FOTG_NO_STREAM
        if no_recompute:
            check(lib().fotg_pipe_submit_ex(self._h, n, _ptr(I0), _ptr(I1), int(u8), _ptr(initflow), _ptr(outflow), after, 1, ticket))     # FOTG_SUBMIT_NO_RECOMPUTE
        else:
            fn = lib().fotg_pipe_submit_u8 if u8 else lib().fotg_pipe_submit
            check(fn(self._h, n, _ptr(I0), _ptr(I1), _ptr(initflow), _ptr(outflow), after, ticket))
        return ticket.value, outflow

    def wait(self, ticket, host=False):
        """the current torch stream (host=True: the calling thread) waits for batch `ticket`"""
        check(lib().fotg_pipe_wait(self._h, int(ticket), _stream(self.device), 1 if host else 0))

    def synchronize(self):
        check(lib().fotg_pipe_sync(self._h))

    def take_stalls(self):
        """for callers of wait(ticket, host=False): AFTER their own synchronisation, how many slots report a timed-out
        inter-workgroup wait since the last query (FOTG_ERR_STALL: the batches computed since are not valid; re-submit).
        Read-and-clear, does not synchronise."""
        return sum(int(lib().fotg_ctx_counter(self.context(k), b"take_stall")) for k in range(self.depth))

    def context(self, slot):
        h = C.c_void_p()
        check(lib().fotg_pipe_context(self._h, int(slot), h))
        return h

    def close(self):
        if getattr(self, "_h", None):
            lib().fotg_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
