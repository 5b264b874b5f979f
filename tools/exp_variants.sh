#!/bin/bash
# headline figures for the product library and each tools/exp/libfotg_*.so variant, selected through FOTG_EXPERIMENTAL_LIB
# (flowonthego_amd/_lib.py; the product library is never overwritten): tools/exp_variants.sh [test]
for lib in "" tools/exp/libfotg_*.so ""; do
  [ -n "$lib" ] && [ ! -e "$lib" ] && continue
  export FOTG_EXPERIMENTAL_LIB=${lib:+$PWD/$lib}; [ -z "$lib" ] && unset FOTG_EXPERIMENTAL_LIB
  echo -n "$(basename ${lib:-libfotg.so}): "
  timeout 600 python bench.py --no-cpu-baseline --windows 9 >/dev/null 2>&1; python -c "
import json
r=json.load(open('gpurun_out/bench_detail.json'))
d=r.get('roofline_dominant',{})
print('in flight', round(r['value']), 'one at a time', round(r['one_batch_at_a_time']['value']), 'sor call us', round(d.get('ms_per_launch',0)*1e3,2), 'varref4', r['stage_ms'].get('varref[4]'))"
  if [ "$1" = test ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_solver or varref_golden or batch64" 2>&1 | tail -1; fi
done
unset FOTG_EXPERIMENTAL_LIB
