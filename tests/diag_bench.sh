#!/bin/bash
# quick A/B: kernel times for the headline family and a many-short-sequences family
for fam in tiled natural; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --family $fam | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('$fam', 'value', d['value'], 'ratio', d['config']['compression_ratio'], ' '.join('%s=%sus' % (n.replace('cimg_',''), v['avg_us']) for n,v in k.items()))"
done
