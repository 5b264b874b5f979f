#!/bin/bash
# stage times of the headline workload with each tools/exp/libfotg_*.so variant swapped in (scratch copy on the GPU box only)
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so /tmp/libfotg_base.so tools/exp/libfotg_*.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo -n "$(basename $lib): "; timeout 300 python tools/stage_times.py 1920 1080 2 64 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sum_ms'], d['stage_ms']['varref[4]'], d['stage_ms']['varref[5]'], d['stage_ms']['varref[6]'])"
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
