#!/bin/bash
# usage: envab2.sh "<ENV=VAL ...>" [bench args]  -- prints value and ms_per_step
envs=$1; shift
env $envs python bench.py --no-cpu-baseline --steps 50 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$envs', 'value', d['value'], 'ms/step', d['ms_per_step'], 'sum kernels us', round(sum(v['avg_us'] for v in k.values()),1))"
