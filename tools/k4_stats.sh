OUT=$PWD/gpurun_out/k4fast; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py --fast > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/k_kernel_stats.csv")))
for r in rows[:25]: print(r["Name"][:70].ljust(70), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
