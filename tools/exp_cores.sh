#!/bin/bash
# in-flight rate against the register / LDS footprint of the fused-level kernels
export GPU_MAX_HW_QUEUES=6
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[6]'], s['varref[5]'], s['varref[4]'])"; }
run "X=0"
run "FOTG_VR_FUSED_NT=512"
run "FOTG_VR_FUSED_RES=0"
run "FOTG_VR_FUSED_RES=0 FOTG_VR_CLDS=0"
run "FOTG_VR_CLDS=0"
run "X=0"
