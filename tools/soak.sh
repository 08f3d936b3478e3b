#!/bin/bash
set -e
for seed in 1 2 3 4 5 6; do
  CIMG_TEST_SEED=$seed CIMG_TEST_ROUNDS=1500 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "randomized_geometries_against" 2>&1 | tail -1
done
timeout -k 10 300 python bench.py --steps 3000 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --family natural 2>/dev/null | cut -c1-160
