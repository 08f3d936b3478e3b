#!/bin/bash
for n in 1 2 3 4 5; do
  export CIMG_ENC_WGS_PER_CU=$n
  timeout -k 10 100 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('encode workgroups per CU $n:', 'encode %.1f us' % k['cimg_encode_streams']['avg_us'])"
done
