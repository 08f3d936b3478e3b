#!/bin/bash
# A/B of differently built libraries: gpurun_in/libvar_*.so (correctness repro, then kernel times)
for lib in gpurun_in/libvar_*.so; do
  echo "== $lib"
  CIMG_LIB=$PWD/$lib timeout -k 3 30 python tools/diag_variant.py 2>&1 | grep -vE "^  File|^$|Thread|amdgpu.ids" | tail -1 || exit 1
  CIMG_LIB=$PWD/$lib timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print('value', d['value'], ' '.join('%s=%sus' % (n.replace('cimg_',''), v['avg_us']) for n,v in k.items()))" || exit 1
done
