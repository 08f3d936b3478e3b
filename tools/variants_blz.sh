#!/bin/bash
fam=$1; shift
for lib in "$@"; do
  CIMG_LIB=$PWD/gpurun_in/$lib python bench.py --no-cpu-baseline --steps 30 --family $fam --codec blosclz 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib', '$fam', 'blosclz enc', [v for n,v in k.items() if n.startswith('cimg_encode')][0]['avg_us'], 'dec', [v for n,v in k.items() if n.startswith('cimg_decode')][0]['avg_us'], 'value', d['value'])"
done
