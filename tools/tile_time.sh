#!/bin/bash
# 4K op-pt 4 pair time + tile solver parity tests + per-grid durations of the tile kernel (run on the GPU box)
timeout 100 python tools/time_4k_op4.py
timeout 300 python -m pytest tests -m gpu -x -q -k "tile_solver or taller" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k4 -o k -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/k4/k_kernel_trace.csv")):
    name = r["Kernel_Name"].split("(")[0][:60]
    key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
tot = collections.defaultdict(float)
for k, v in d.items(): tot[k] = sum(v) / 7
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]: print("%-62s grid %8d n/pair %5.1f avg us %8.1f per-pair ms %.3f" % (k[0], k[1], len(d[k]) / 7, sum(d[k]) / len(d[k]), v / 1000))
PY
