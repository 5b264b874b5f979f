#!/bin/bash
# tile-solver call times (and, with "test", the tile parity test) for the product library and each tools/exp/libfotg_*.so variant, selected
# through FOTG_EXPERIMENTAL_LIB (flowonthego_amd/_lib.py; the product library is never overwritten)
for lib in "" tools/exp/libfotg_*.so ""; do
  [ -n "$lib" ] && [ ! -e "$lib" ] && continue
  export FOTG_EXPERIMENTAL_LIB=${lib:+$PWD/$lib}; [ -z "$lib" ] && unset FOTG_EXPERIMENTAL_LIB
  echo -n "$(basename ${lib:-libfotg.so}): "; timeout 300 python tools/tile_call_time.py $TILE_ARGS 2>&1 | tail -1
  if [ "$1" = test ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_solver_pipeline" 2>&1 | tail -2; fi
done
unset FOTG_EXPERIMENTAL_LIB
