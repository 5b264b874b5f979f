#!/bin/bash
# like exp_run.sh, printing value / one-batch / pyramid stage time
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo -n "$(basename $lib): "
  python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['one_batch_at_a_time']['value']), d['stage_ms']['pyramid(I0,I1)'])"
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
