#!/bin/bash
fam=$1; shift
for lib in "$@"; do
  CIMG_LIB=$PWD/gpurun_in/$lib python bench.py --no-cpu-baseline --steps 30 --family $fam --codec blosclz 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib', '$fam', 'blosclz enc', k['cimg_encode_streams']['avg_us'], 'dec', k['cimg_decode_blocks']['avg_us'], 'value', d['value'])"
done
