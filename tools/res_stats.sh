#!/bin/bash
# run tools/res_stats.py with every tools/exp/libfotg_stats*.so swapped in (scratch copy on the GPU box only)
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in tools/exp/libfotg_stats*.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo "== $(basename $lib)"
  python tools/res_stats.py "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
