#!/bin/bash
# the fixed-point loop of level 4 in one launch per level (FOTG_VR_LEVEL / FOTG_PIPE_VR_LEVEL)
export GPU_MAX_HW_QUEUES=6
FOTG_VR_LEVEL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "end_to_end or batch64 or batch_1080p or natural_images_1080p or random_sizes or uint8 or sequence or golden" 2>&1 | tail -3
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[4]'], round(d['u8_frames']['in_flight']['value']))"; }
run "X=0"
run "FOTG_PIPE_VR_LEVEL=1"
run "FOTG_VR_LEVEL=1"
run "X=0"
run "FOTG_PIPE_VR_LEVEL=1"
