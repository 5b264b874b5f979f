#!/bin/bash
# instruction counters of the headline kernels (run on the GPU box): tools/pmc_bench.sh "SQ_WAVES SQ_INSTS_VALU ..." tag
OUT=$PWD/gpurun_out/pmcb_${2:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "fotg" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("(")[0][:44], int(r["Grid_Size"]))
    d[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[key] += 1
for k in sorted(d, key=lambda k: -max(d[k].values()))[:12]:
    print("%-46s grid %9d n %3d " % (k[0], k[1], n[k]) + " ".join("%s=%.4g" % (c, v / n[k]) for c, v in sorted(d[k].items())))
PY
