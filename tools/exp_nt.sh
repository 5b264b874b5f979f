#!/bin/bash
# workgroup sizes of the chain kernels against the in-flight rate (a 16-wave workgroup starves under another batch's pyramid)
export GPU_MAX_HW_QUEUES=6
for nt in 1024 512; do echo "== stream NT $nt"; FOTG_VR_STREAM_NT=$nt python tools/corun.py 2>&1 | grep "five sor"; done
FOTG_VR_STREAM_NT=512 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming or end_to_end or batch64 or batch_1080p or natural_images_1080p" 2>&1 | tail -2
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[5]'], s['varref[4]'], round(d['u8_frames']['in_flight']['value']))"; }
run "X=1"
run "FOTG_PIPE_VR_STREAM_NT=512"
run "FOTG_PIPE_VR_STREAM_NT=512 FOTG_PIPE_VR_FUSED_NT=512"
run "FOTG_PIPE_VR_FUSED_NT=512"
run "FOTG_VR_STREAM_NT=512 FOTG_PIPE_VR_STREAM_NT=512"
run "X=1"
