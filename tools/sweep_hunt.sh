#!/bin/bash
# bug hunt: the random-parameter parity sweeps (tests/test_gpu_parity.py, tests/test_gpu_depth.py) with other seeds and more cases
# tools/sweep_hunt.sh [first seed] [seeds] [cases per seed]     (FOTG_TEST_SWEEP_BIG=1: the entry-point sweep on HD .. 4K frames;
# FOTG_TEST_SWEEP_TALL=1: the parameter sweeps on narrow frames of 1 100 .. 2 600 rows refined at full resolution)
S0=${1:-1000}; NS=${2:-6}; NC=${3:-150}
for ((s = S0; s < S0 + NS; ++s)); do
  FOTG_TEST_SWEEP_SEED=$s FOTG_TEST_SWEEP_CASES=$NC timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_depth.py -q -m gpu -x -k "${SWEEP_K:-sweep}" 2>&1 | tail -3 | tr '\n' ' '; echo " [seed $s]"
done
