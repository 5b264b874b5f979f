#!/bin/bash
# 4K operating-point-4 pair time (and, with "test", the tile / level-pipeline parity tests) for the product library and each
# tools/exp/libfotg_*.so variant.  A variant is selected through FOTG_EXPERIMENTAL_LIB (flowonthego_amd/_lib.py): the product library is
# never overwritten, so an interrupted run cannot leave an experimental build installed.  tools/exp_4k.sh [test|stamps]
for lib in "" tools/exp/libfotg_*.so ""; do
  [ -n "$lib" ] && [ ! -e "$lib" ] && continue
  export FOTG_EXPERIMENTAL_LIB=${lib:+$PWD/$lib}; [ -z "$lib" ] && unset FOTG_EXPERIMENTAL_LIB
  echo "$(basename ${lib:-libfotg.so}): $(timeout 300 python tools/time_4k_op4.py $TIME_ARGS 2>&1 | grep 'ms per pair' | tr '\n' ';')  fast: $(timeout 300 python tools/time_4k_op4.py --fast 2>&1 | grep 'ms per pair' | head -1)"
  if [ "$1" = stamps ]; then FOTG_STAMPS_BRIEF=1 timeout 300 python tools/levelpipe_stamps.py 2>&1 | tail -10; fi
  if [ "$1" = test ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_solver or level_pipeline" 2>&1 | tail -1; fi
done
unset FOTG_EXPERIMENTAL_LIB
