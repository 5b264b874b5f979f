#!/bin/bash
# 4K operating-point-4 pair time (and, with "test", the tile / level-pipeline parity tests) with each tools/exp/libfotg_*.so variant swapped in
# (scratch copy on the GPU box only): tools/exp_4k.sh [test]
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so /tmp/libfotg_base.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo "$(basename $lib): $(timeout 300 python tools/time_4k_op4.py $TIME_ARGS 2>&1 | grep 'ms per pair' | tr '\n' ';')  fast: $(timeout 300 python tools/time_4k_op4.py --fast 2>&1 | grep 'ms per pair' | head -1)"
  if [ "$1" = stamps ]; then FOTG_STAMPS_BRIEF=1 timeout 300 python tools/levelpipe_stamps.py 2>&1 | tail -10; fi
  if [ "$1" = test ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_solver or level_pipeline" 2>&1 | tail -1; fi
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
