#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/prof/icache
mkdir -p $out
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/$name.err
done
python3 profiles/tools/summarize_pmc.py $out
