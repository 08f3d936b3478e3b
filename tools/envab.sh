#!/bin/bash
# usage: envab.sh "<ENV=VAL ...>" [bench args]   -- prints encode/decode kernel times
envs=$1; shift
env $envs python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$envs', 'enc', k['cimg_encode_streams']['avg_us'], 'emit', k['cimg_emit_blocks']['avg_us'], 'dec', k['cimg_decode_blocks']['avg_us'], 'value', d['value'])"
