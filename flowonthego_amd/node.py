"""FlowNode: one process, several GPUs (fotg_node_* of include/fotg.h; SURVEY.md 8e "one process per node with one host thread
+ stream set per GPU").  A batch of n frame pairs is cut into contiguous shards -- slot d gets n // ndev pairs, the first n % ndev slots
one more: fotg_node_shard, the same ranges as flowonthego_amd.shard.shard_range --, every slot runs its shard through a pipe (fotg_pipe_*) on its device.  Results are
bit-identical to OFClass.calc_batch.  The reference drives one device (src/run_dense.cpp:277-289)."""
import ctypes as C

import torch

from ._lib import FotgError, check, lib
from .oflow import _dev_f32
from .params import img_params, opt_params, padded_size


def node_shard(n, ndev, d):
    """[begin, end) of slot d's pairs (fotg_node_shard)"""
    b, c = C.c_int(), C.c_int()
    check(lib().fotg_node_shard(int(n), int(ndev), int(d), b, c))
    return b.value, b.value + c.value


class FlowNode:
    def __init__(self, _op: opt_params, _i_params: img_params, devices, max_batch: int = 64, depth: int = 4):
        self.op = _op.derive()
        self.devices = [int(d) for d in devices]
        self.ndev, self.max_batch, self.depth = len(self.devices), int(max_batch), int(depth)
        self.width_org, self.height_org = int(_i_params.width), int(_i_params.height)
        self.width, self.height, _, _ = padded_size(self.width_org, self.height_org, self.op.coarsest_scale)
        if _i_params.padding not in (0, self.op.patch_size):
            raise FotgError("img_params.padding must equal patch_size (src/run_dense.cpp:263)")
        self.nch = 1 if self.op.depth_mode else 2
        h = C.c_void_p()
        devs = (C.c_int * self.ndev)(*self.devices)
        check(lib().fotg_node_create(self.op.to_c(), self.width_org, self.height_org, devs, self.ndev, self.max_batch, self.depth, h))
        self._h = h

    def out_size(self):
        return self.width >> self.op.finest_scale, self.height >> self.op.finest_scale

    def shard(self, n, d):
        return node_shard(n, self.ndev, d)

    def _frame_shape(self, n, u8=False):
        return (n, self.height_org, self.width_org) + ((3,) if u8 and self.op.u8_color else (self.op.channels,) if self.op.channels > 1 else ())

    def submit(self, n, I0, I1, outflow=None):
        """resident frames: I0[d], I1[d] = slot d's shard (its node_shard range of the n pairs) on device devices[d], float32 or
        uint8; outflow[d] (count_d, h_l, w_l, nch) on the same device, allocated if None.  The frames must be in place (the work
        starts at once: synchronise the producing streams first).  Returns (ticket, outflow list)."""
        u8 = any(t is not None and t.dtype == torch.uint8 for t in I0)
        w, h = self.out_size()
        outs = list(outflow) if outflow is not None else [None] * self.ndev
        for d in range(self.ndev):
            b, e = self.shard(n, d)
            if e == b:
                continue
            dev = torch.device("cuda", self.devices[d])
            for t, nm in ((I0[d], "I0[%d]" % d), (I1[d], "I1[%d]" % d)):
                _dev_f32(t, nm, dev, dtype=torch.uint8 if u8 else torch.float32)
                if tuple(t.shape) != self._frame_shape(e - b, u8) and tuple(t.shape) != self._frame_shape(e - b, u8) + (1,):
                    raise FotgError("%s has shape %s, slot %d's shard is %s" % (nm, tuple(t.shape), d, self._frame_shape(e - b, u8)))
            if outs[d] is None:
                outs[d] = torch.empty((e - b, h, w, self.nch), dtype=torch.float32, device=dev)
                torch.cuda.current_stream(dev).synchronize()        # (the allocator may hand out memory with work still enqueued on a torch stream)
            _dev_f32(outs[d], "outflow[%d]" % d, dev, (e - b, h, w, self.nch))
        arr = lambda ts: (C.c_void_p * self.ndev)(*[(t.data_ptr() if t is not None else None) for t in ts])
        ticket = C.c_long()
        fn = lib().fotg_node_submit_u8 if u8 else lib().fotg_node_submit
        check(fn(self._h, int(n), arr(I0), arr(I1), arr(outs), ticket))
        return ticket.value, outs

    def submit_scatter(self, I0, I1, outflow=None, chunk=None):
        """the whole batch (n, h, w[, channels]) float32 or uint8 on devices[0]; the other slots pull their shards in chunks of
        `chunk` pairs while computing (8-bit frames travel as bytes) and write their flows back into outflow (n, h_l, w_l, nch) on
        devices[0]"""
        dev = torch.device("cuda", self.devices[0])
        u8 = I0.dtype == torch.uint8
        I0, I1 = _dev_f32(I0, "I0", dev, dtype=torch.uint8 if u8 else torch.float32), _dev_f32(I1, "I1", dev, dtype=torch.uint8 if u8 else torch.float32)
        n = I0.shape[0]
        if (tuple(I0.shape) != self._frame_shape(n, u8) and tuple(I0.shape) != self._frame_shape(n, u8) + (1,)) or I1.shape != I0.shape:
            raise FotgError("frame shape %s does not match the configured %s" % (tuple(I0.shape), self._frame_shape(n, u8)))
        w, h = self.out_size()
        if outflow is None:
            outflow = torch.empty((n, h, w, self.nch), dtype=torch.float32, device=dev)
            torch.cuda.current_stream(dev).synchronize()
        _dev_f32(outflow, "outflow", dev, (n, h, w, self.nch))
        ticket = C.c_long()
        check((lib().fotg_node_submit_scatter_u8 if u8 else lib().fotg_node_submit_scatter)(self._h, n, C.c_void_p(I0.data_ptr()), C.c_void_p(I1.data_ptr()), C.c_void_p(outflow.data_ptr()),
                                             int(chunk or self.max_batch), ticket))
        return ticket.value, outflow

    def wait(self, ticket):
        check(lib().fotg_node_wait(self._h, int(ticket)))

    def synchronize(self):
        check(lib().fotg_node_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().fotg_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
