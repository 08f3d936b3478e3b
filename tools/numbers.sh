#!/bin/bash
# the table of DESIGN.md section 9: kernel microseconds per 128 MiB and combined GB/s for every data family / codec / filter
run() {
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('$*', '| value', d['value'], 'ratio', d['config']['compression_ratio'], ' '.join('%s=%sus' % (n.replace('cimg_',''), v['avg_us']) for n,v in k.items()))"
}
for fam in tiled natural; do for codec in lz4 blosclz; do run --family $fam --codec $codec; done; done
run --family tiled --codec lz4 --filter bitshuffle
run --family tiled --codec blosclz --filter bitshuffle
run --family zero
run --family random
run --config 4
