#!/bin/bash
# instruction counters of the 4K op-pt 4 kernels (run on the GPU box): tools/pmc_4k.sh "SQ_WAVES SQ_INSTS_VALU ..." tag [--fast]
OUT=$PWD/gpurun_out/pmc4k_${2:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py $3 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    key = (r["Kernel_Name"].split("(")[0][:50], int(r["Grid_Size"]))
    d[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[key] += 1
for k in sorted(d, key=lambda k: -d[k].get("SQ_INSTS_VALU", 0))[:10]:
    print("%-52s grid %8d n %3d " % (k[0], k[1], n[k]) + " ".join("%s=%.4g" % (c, v / n[k]) for c, v in sorted(d[k].items())))
PY
