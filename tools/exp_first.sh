#!/bin/bash
export GPU_MAX_HW_QUEUES=6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "varref or end_to_end or batch64 or batch_1080p or natural or random_sizes or streaming or tile or taller or 4k" 2>&1 | tail -2
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[4]'], round(d['u8_frames']['in_flight']['value']))"; }
run "FOTG_VR_FIRST_DATA=0"
run "FOTG_VR_FIRST_DATA=1"
run "FOTG_VR_FIRST_DATA=0"
run "FOTG_VR_FIRST_DATA=1"
python tools/time_4k_op4.py 2>&1 | tail -1
FOTG_VR_FIRST_DATA=0 python tools/time_4k_op4.py 2>&1 | tail -1
