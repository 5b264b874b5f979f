#!/bin/bash
# tile-solver call times (and, with "test", the tile parity test) with each tools/exp/libfotg_*.so variant swapped in (scratch copy on the GPU box only)
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so /tmp/libfotg_base.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo -n "$(basename $lib): "; timeout 300 python tools/tile_call_time.py $TILE_ARGS 2>&1 | tail -1
  if [ "$1" = test ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_solver_pipeline" 2>&1 | tail -2; fi
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
