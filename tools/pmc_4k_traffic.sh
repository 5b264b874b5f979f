#!/bin/bash
# HBM traffic counters of the 4K op-pt 4 kernels, separate passes (run on the GPU box): tools/pmc_4k_traffic.sh tag
OUT=$PWD/gpurun_out/pmc4k_traffic_${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections, json
res = {}
for tag, f, cn in (("f", "$OUT/f/f_counter_collection.csv", "FETCH_SIZE"), ("w", "$OUT/w/w_counter_collection.csv", "WRITE_SIZE")):
    d = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cn: continue
        key = "%s [grid %s]" % (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])
        d[key] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[key] += 1
    for k in d: res.setdefault(k, {})[cn + "_KB_avg_per_launch"] = d[k] / n[k]; res[k]["launches_" + cn] = n[k]
for k, v in res.items():
    v["hbm_bytes_per_launch_corrected"] = (2 * v.get("FETCH_SIZE_KB_avg_per_launch", 0) + v.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/time_4k_op4.py",
       "workload": "BASELINE configs[3]: one 3840x2160 gray f32 pair, operating point 4; 7 calls",
       "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters x 1024); the x 2 is the guide's gfx950 correction for 16-byte-per-lane streaming reads",
       "kernels": {k: res[k] for k in sorted(res, key=lambda k: -res[k]["hbm_bytes_per_launch_corrected"] * res[k].get("launches_FETCH_SIZE", 1))[:14]}}
json.dump(out, open("$OUT/${1:-a}_4k_pmc_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items(): print(k, round(v["hbm_bytes_per_launch_corrected"] / 1e6, 2), "MB x", v.get("launches_FETCH_SIZE"))
PY
