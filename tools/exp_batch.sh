#!/bin/bash
# pairs/s for different (batch, batches in flight) at the same or similar number of resident pairs
run() { echo -n "batch $1 x in-flight $2 (queues $3): "; GPU_MAX_HW_QUEUES=$3 python bench.py --no-cpu-baseline --no-breakdown --batch $1 --in-flight $2 --steps $4 --windows 9 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['one_batch_at_a_time']['value']), d['ms_per_step'])"; }
run 256 2 4 25
run 256 3 6 25
run 256 4 6 25
run 512 1 4 12
run 512 2 4 12
