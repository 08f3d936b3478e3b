#!/bin/bash
# longer randomized differential soak: seeds 7..30, 1500 geometries each, GPU bytes and pixels against the oracle
set -e
for seed in $(seq 7 30); do
  CIMG_TEST_SEED=$seed CIMG_TEST_ROUNDS=1500 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "randomized_geometries_against" 2>&1 | tail -1
done
