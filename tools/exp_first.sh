#!/bin/bash
# A/B of a creation-time switch: parity subset, then bench with the switch off / on.   usage: tools/exp_first.sh FOTG_VR_DENSIFY
SW=${1:-FOTG_VR_FIRST_DATA}
export GPU_MAX_HW_QUEUES=6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "varref or end_to_end or batch64 or batch_1080p or natural or random_sizes or uint8 or sequence or golden or custom_patch or cost or initflow or degenerate or early" 2>&1 | tail -2
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s, round(d['u8_frames']['in_flight']['value']))"; }
run "$SW=0"
run "$SW=1"
run "$SW=0"
run "$SW=1"
