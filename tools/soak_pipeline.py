#!/usr/bin/env python3
"""Determinism soak of the pipelined path (not part of the test suite): D slots with different frames, many submits; every
result of every slot must equal the single-context result of that slot's frames (a race between overlapping batches -- shared
state between contexts, an event recorded on the wrong stream -- would show up as an occasional mismatch).
usage: python tools/soak_pipeline.py [submits] [batch] [in flight] [op-point]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass
from flowonthego_amd.pipeline import FlowPipeline
it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
op = F.operating_point(int(sys.argv[4]) if len(sys.argv) > 4 else 2, bench.W, 1)
ip = F.img_params(width=bench.W, height=bench.H, padding=op.patch_size)
ofc = OFClass(op, ip, max_batch=batch)
pipe = FlowPipeline(op, ip, max_batch=batch, depth=D)
slots = [bench.synth_batch(batch, 300 + k, dev) for k in range(D)]
refs = [ofc.calc_batch(a, b).clone() for a, b in slots]
outs = [[pipe.new_outflow(batch) for _ in range(2)] for _ in range(D)]       # two output buffers per slot, used in turn
# the pipe's streams are not torch's: everything enqueued on the torch stream so far (the references, and whatever used the
# memory the allocator has just recycled into `outs`) has to be finished before submits that do not wait for that stream
torch.cuda.synchronize()
bad = 0
for k in range(it):
    s = k % D
    pipe.submit(slots[s][0], slots[s][1], None, outs[s][(k // D) % 2], after_current_stream=False)
    if k % (8 * D) == 8 * D - 1:                                             # check everything that has been written so far
        pipe.synchronize()
        for s2 in range(D):
            for o in outs[s2]:
                if not torch.equal(o, refs[s2]):
                    bad += 1
                    print("submit", k, "slot", s2, "differs: max abs", float((o - refs[s2]).abs().max()), flush=True)
pipe.synchronize()
print("submits", it, "batch", batch, "in flight", D, "mismatches", bad)
sys.exit(1 if bad else 0)
