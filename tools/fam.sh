#!/bin/bash
for f in zero random natural tiled; do
  timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --family $f 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('$f', 'value', d['value'], 'ratio', d['config']['compression_ratio'], ' '.join('%s=%sus' % (n.replace('cimg_',''), v['avg_us']) for n,v in k.items()))"
done
