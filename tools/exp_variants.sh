#!/bin/bash
# headline figures with each tools/exp/libfotg_*.so variant swapped in (scratch copy on the GPU box only): tools/exp_variants.sh [test]
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so /tmp/libfotg_base.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo -n "$(basename $lib): "
  timeout 600 python bench.py --no-cpu-baseline --windows 9 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
d=r.get('roofline_dominant',{})
print('in flight', round(r['value']), 'one at a time', round(r['one_batch_at_a_time']['value']), 'sor call us', round(d.get('ms_per_launch',0)*1e3,2), 'varref4', r['stage_ms'].get('varref[4]'))"
  if [ "$1" = test ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_solver or varref_golden or batch64" 2>&1 | tail -1; fi
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
