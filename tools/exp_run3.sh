#!/bin/bash
# like exp_run2.sh, all stage times; optional parity subset per variant
export GPU_MAX_HW_QUEUES=6
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo "== $(basename $lib)"
  if [ "$lib" != /tmp/libfotg_base.so ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyramid or patchgrid or varref or end_to_end or batch64 or uint8 or random_sizes" 2>&1 | tail -2; fi
  for i in 1 2; do
  python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['one_batch_at_a_time']['value']), d['stage_ms'], round(d['u8_frames']['in_flight']['value']))"
  done
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
