#!/bin/bash
# same-box A/B of differently built libraries (gpurun_in/<lib>, CIMG_LIB): usage variants.sh <family> lib...
fam=$1; shift
for lib in "$@"; do
  CIMG_LIB=$PWD/gpurun_in/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --family $fam 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib', '$fam', 'enc', [v for n,v in k.items() if n.startswith('cimg_encode')][0]['avg_us'], 'layout+emit', k['cimg_layout_chunks']['avg_us'] + k['cimg_emit_blocks']['avg_us'], 'dec', [v for n,v in k.items() if n.startswith('cimg_decode')][0]['avg_us'], 'ms/step', d['ms_per_step'], 'value', d['value'])
except Exception as e:
    print('$lib', '$fam', 'FAILED', e)"
done
