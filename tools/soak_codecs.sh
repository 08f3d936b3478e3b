#!/bin/bash
# randomized differential soak of the two format-valid codecs (zstd / lz4hc write side): seeds 1..12, 3000 rounds each
set -e
for seed in $(seq 1 12); do
  CIMG_TEST_SEED=$seed CIMG_TEST_ROUNDS=3000 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k randomized_geometries_zstd 2>&1 | tail -1
done
