#!/bin/bash
# loaders' lead of the streaming solver (FOTG_VR_LEAD / FOTG_PIPE_VR_LEAD): co-run with the pyramid, parity, in-flight rate
export GPU_MAX_HW_QUEUES=6
for l in 3 4 5 6 8; do echo "== lead $l"; FOTG_VR_LEAD=$l python tools/corun.py 2>&1 | grep "five sor"; done
FOTG_VR_LEAD=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming or end_to_end or batch64 or batch_1080p or natural_images_1080p" 2>&1 | tail -2
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[4]'], round(d['u8_frames']['in_flight']['value']))"; }
run "FOTG_PIPE_VR_LEAD=3"
run "FOTG_PIPE_VR_LEAD=4"
run "FOTG_PIPE_VR_LEAD=5"
run "FOTG_PIPE_VR_LEAD=6"
run "FOTG_PIPE_VR_LEAD=8"
run "FOTG_PIPE_VR_LEAD=8 FOTG_VR_LEAD=8"
run "FOTG_PIPE_VR_LEAD=3"
