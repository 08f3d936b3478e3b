#!/bin/bash
# profiles/tools/collect.sh <tag> -- run ON THE GPU BOX (via gpurun) from the repo root.
# Produces gpurun_out/prof/<tag>/{kernel_stats.csv, pmc_per_launch.json, bench.json, bench_under_rocprof.json}.
# Counter passes are separate runs (--pmc only with --kernel-trace, never with other trace domains).
set -e
tag=${1:-run}
out=gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
echo "[collect] kernel trace"; 
# (CIMG_BENCH_NO_NATURAL: the default run also times a few steps of the natural family for its headline line; under the profiler
# they would land in the same kernels' averages)
CIMG_BENCH_NO_NATURAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B > $out/bench_under_rocprof.json 2> $out/kt.err
cp $(ls $out/kt/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_IFETCH SQC_ICACHE_MISSES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  echo "[collect] pmc $set"
  CIMG_BENCH_NO_NATURAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_$name.err
done
python3 profiles/tools/summarize_pmc.py $out > $out/pmc_per_launch.json
# the plain bench below takes roofline.traffic from profiles/<tag>/pmc_per_launch.json when its source hash matches: put the fresh one there first
mkdir -p profiles/$tag && cp $out/pmc_per_launch.json profiles/$tag/pmc_per_launch.json
echo "[collect] plain bench"
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
