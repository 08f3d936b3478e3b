#!/bin/bash
for pad in 0 3600 6400 10000 16000; do
  CIMG_LEAN_LDS_PAD=$pad CIMG_LIB=$PWD/gpurun_in/lib_BOTHOLD.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('pad $pad dec', k['cimg_decode_blocks']['avg_us'])"
done
