#!/bin/bash
# profiles of one round, run ON the GPU box from the repo root: tools/profile_round.sh r02
# (1) kernel stats of the batch-64 headline launches only (one batch at a time: clean per-kernel durations; a second pass with the
#     default four batches in flight), (2)+(3) HBM traffic counters in their own passes (no other tracing)
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 20 --warmup 3 --windows 3 --no-cpu-baseline --no-breakdown > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fl4 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --windows 3 --no-cpu-baseline --no-breakdown > $OUT/fl4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k4 -o k -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/k4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/valu -o p -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown > $OUT/valu.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/valu4k -o p -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/valu4k.log 2>&1
# the tolerance mode (fotg_params::fast_math): kernel stats and instruction counters of the same two workloads
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k4f -o k -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py --fast > $OUT/k4f.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/valuf -o p -- python3 $GRAFT_REPO_ROOT/bench.py --fast-math --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown > $OUT/valuf.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/valu4kf -o p -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py --fast > $OUT/valu4kf.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f4k -o f -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/f4k.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w4k -o w -- python3 $GRAFT_REPO_ROOT/tools/time_4k_op4.py > $OUT/w4k.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/make_valu_profile.py $OUT/valu $OUT/valu4k $TAG $OUT/valuf $OUT/valu4kf >> $OUT/summary_valu.txt 2>&1
python3 tools/make_4k_profiles.py $OUT/f4k $OUT/w4k $OUT/k4f $OUT/k4f.log $TAG >> $OUT/summary_valu.txt 2>&1
python3 tools/make_profiles.py $OUT/stats $OUT/fetch $OUT/write $TAG $OUT/k4 $OUT/k4.log $OUT/fl4 > $OUT/summary.txt 2>&1
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
cp $OUT/bench_line.json profiles/${TAG}_bench_line.json              # the ONE short stdout line the driver parses
cp gpurun_out/bench_detail.json profiles/${TAG}_bench_detail.json     # everything else (stage tables, rooflines list, notes)
cp profiles/${TAG}_bench_detail.json $OUT/
cp profiles/${TAG}_* $OUT/ 2>/dev/null
tail -20 $OUT/summary.txt; tail -2 $OUT/k4.log
