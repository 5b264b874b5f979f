#!/bin/bash
# experiment: all sweeps of the fused levels in one wave (FOTG_VR_ONEWAVE)
export GPU_MAX_HW_QUEUES=6
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "varref or end_to_end or random_sizes or batch64" 2>&1 | tail -5
FOTG_VR_FUSED_NT=512 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "varref or end_to_end or batch64" 2>&1 | tail -5
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(round(d['value']), round(d['one_batch_at_a_time']['value']), s['varref[6]'], s['varref[5]'], s['varref[4]'])"; }
run "FOTG_VR_ONEWAVE=0"
run "FOTG_VR_ONEWAVE=1"
run "FOTG_VR_ONEWAVE=0 FOTG_VR_FUSED_NT=512"
run "FOTG_VR_ONEWAVE=1 FOTG_VR_FUSED_NT=512"
run "FOTG_VR_ONEWAVE=3 FOTG_VR_FUSED_NT=512"
