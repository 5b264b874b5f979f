#!/usr/bin/env python3
"""What runs beside what, from a rocprofv3 --kernel-trace of the four-batches-in-flight bench run (tools/profile_round.sh keeps it
under gpurun_out/prof_<tag>/fl4): for every kernel class its total running time, the share of that time during which at least
one pyramid launch of ANOTHER queue was running, the mean number of kernels running beside it, and per queue the share of the
steady-state window with a kernel running.   usage: tools/overlap.py <kernel_trace.csv> [out.md]"""
import csv, collections, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fotg" not in n:
        continue
    cls = n.split("fotg::")[1].split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls, r["Queue_Id"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])))
# the pipelined part of the run: the queues that carry the sub-divided pyramid launches (only pipes cut them); their dispatches
# come in bursts (bench.py's timed windows, each closed by a host wait): keep the bursts, trim a sixth of each at both ends
qcount = collections.Counter(q for _, _, c, q, g in rows if c == "pyr_base_kernel")
small = collections.Counter(q for _, _, c, q, g in rows if c == "pyr_base_kernel" and g < 1000000)
pipeq = {q for q in qcount if small[q] > 0.5 * qcount[q]}
rows = sorted(r for r in rows if r[3] in pipeq)
bursts, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - end > 300000:
        bursts.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
bursts.append(cur)
bursts = [b for b in bursts if len(b) > 200]
keep, window = [], 0
for b in bursts:
    b0, b1 = b[0][0], max(x[1] for x in b)
    lo_, hi_ = b0 + (b1 - b0) // 6, b1 - (b1 - b0) // 6
    keep += [x for x in b if x[0] >= lo_ and x[1] <= hi_]
    window += hi_ - lo_
rows = keep
lo, hi = 0, window
ev = []
for i, (s, e, c, q, g) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set()
tot = collections.Counter(); with_pyr = collections.Counter(); conc = collections.Counter()
busy_q = collections.Counter(); nk_time = collections.Counter()
prev = None
for t, d, i in ev:
    if prev is not None and t > prev and active:
        dt = t - prev
        nk_time[len(active)] += dt
        for q in {rows[j][3] for j in active}:
            busy_q[q] += dt
        for j in active:
            c, q = rows[j][2], rows[j][3]
            tot[c] += dt
            conc[c] += dt * (len(active) - 1)
            if any(rows[k][2] == "pyr_base_kernel" and rows[k][3] != q for k in active):
                with_pyr[c] += dt
    if d == 1: active.add(i)
    else: active.discard(i)
    prev = t
out = []
out.append("# kernels of the four-batches-in-flight run (batch 64 x 1080p op-pt 2 + refinement): what runs beside what")
out.append("# source: rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 3 --windows 3` (tools/profile_round.sh), the middle two thirds of every pipelined window: %.1f ms in all, %d dispatches on %d queues" % ((hi - lo) / 1e6, len(rows), len(pipeq)))
nsteps = sum(1 for r in rows if r[2] == "pyr_finish_kernel")
out.append("# CAVEAT: under the tracer a launch costs the host ~15 us, so the traced run is launch-bound: %.3f ms per step here against" % ((hi - lo) / 1e6 / max(nsteps, 1)))
out.append("# 0.354 ms untraced (profiles/r03_bench_line.json).  The queues below are idle most of the time waiting for the host; the untraced")
out.append("# run keeps them fed (host issue 0.106 ms per step).  Read this table for WHICH kernels share the chip, not for how long.")
out.append("")
out.append("| kernel class | running time (ms) | share with a pyramid launch of another slot running | mean number of other kernels running |")
out.append("|---|---|---|---|")
for c, v in tot.most_common():
    out.append("| %s | %.2f | %.0f %% | %.2f |" % (c, v / 1e6, 100.0 * with_pyr[c] / v, conc[c] / v))
out.append("")
out.append("| kernels running at once | share of the window |")
out.append("|---|---|")
nk_time[0] = (hi - lo) - sum(nk_time.values())      # the rest of the kept windows: nothing running
for k in sorted(nk_time):
    out.append("| %d | %.1f %% |" % (k, 100.0 * nk_time[k] / (hi - lo)))
out.append("")
out.append("per queue (slot), share of the window with one of its kernels running: " + ", ".join("%.0f %%" % (100.0 * busy_q[q] / (hi - lo)) for q in sorted(busy_q)))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
