#!/bin/bash
for mode in events noevents; do
  if [ $mode = noevents ]; then export CIMG_BENCH_NO_EVENTS=1; else unset CIMG_BENCH_NO_EVENTS; fi
  timeout -k 10 100 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('$mode value', d['value'], 'ms/step', d['ms_per_step'], 'kernel sum us', round(sum(v['avg_us'] for v in k.values()),1))"
done
