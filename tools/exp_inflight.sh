#!/bin/bash
# in-flight rate of the headline workload with each tools/exp/libfotg_*.so variant swapped in, alternating (scratch copy on the GPU box only)
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for rep in 1 2 3; do
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo -n "$(basename $lib): "; timeout 300 python tools/inflight_rate.py 4 64 100 9 1 2>&1 | tail -1
done
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
