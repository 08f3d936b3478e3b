# profiles/tools/extra.sh <tag> -- run ON THE GPU BOX (via gpurun) from the repo root: everything in profiles/<tag>/ besides the four files of collect.sh
set -x
tag=${1:-r05}
out=gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config 1 > $out/bench_config1.json 2> $out/bench_config1.err
timeout -k 10 300 python bench.py --config 3 > $out/bench_config3.json 2> $out/bench_config3.err
timeout -k 10 500 python bench.py --config 5 > $out/bench_config5.json 2> $out/bench_config5.err
echo "[extra] config benches done"
timeout -k 10 200 python tools/diag_stamps_hw.py > $out/decode_slots.txt 2>&1
echo "[extra] slots done"
{ echo "## tools/diag_zstd_dev.py (default: walk + lit + seq + replay)"; timeout -k 10 200 python tools/diag_zstd_dev.py 2>&1 | tail -4;
  echo "## --clevel 5 (split planes)"; timeout -k 10 200 python tools/diag_zstd_dev.py --clevel 5 2>&1 | tail -4;
  echo "## --dtype uint8 (byte-wide pixels: one stream per block at every level)"; timeout -k 10 200 python tools/diag_zstd_dev.py --dtype uint8 2>&1 | tail -4;
  echo "## --dtype uint16"; timeout -k 10 200 python tools/diag_zstd_dev.py --dtype uint16 2>&1 | tail -4;
  echo "## CIMG_ZSTD_LANES=0 (the walkers decode sequences and literals themselves)"; CIMG_ZSTD_LANES=0 timeout -k 10 200 python tools/diag_zstd_dev.py 2>&1 | tail -4;
  echo "## CIMG_ZSTD_FUSED=1 (cimg_decode_zstd alone)"; CIMG_ZSTD_FUSED=1 timeout -k 10 200 python tools/diag_zstd_dev.py 2>&1 | tail -4; } > $out/zstd_read_path.txt
echo "[extra] zstd read path done"
rm -rf $out/zkt; CIMG_DIAG_EXIT=clean timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/zkt -- python3 tools/diag_zstd_dev.py > $out/zkt.log 2>&1
f=$(ls $out/zkt/*/*_kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $out/zstd_kernel_stats.csv; fi
echo "[extra] zstd rocprof done"
timeout -k 10 900 bash tools/numbers.sh > $out/numbers.txt 2>&1
echo "[extra] numbers done"
{ echo "## CIMG_HOST_REGISTER_MIB=0 tools/diag_hostpin.py"; CIMG_HOST_REGISTER_MIB=0 timeout -k 10 200 python tools/diag_hostpin.py 2>&1 | tail -6;
  echo "## CIMG_HOST_REGISTER_MIB=32 tools/diag_hostpin.py"; CIMG_HOST_REGISTER_MIB=32 timeout -k 10 200 python tools/diag_hostpin.py 2>&1 | tail -6;
  echo "## tools/diag_real_sizes.py"; timeout -k 10 300 python tools/diag_real_sizes.py 2>&1 | tail -12;
  echo "## tools/diag_pymodule.py"; timeout -k 10 300 python tools/diag_pymodule.py 2>&1 | tail -12; } > $out/host_path.txt
echo "[extra] host path done"
{ echo "## tools/diag_dtypes.py lz4"; timeout -k 10 400 python tools/diag_dtypes.py lz4 2>&1 | grep -v amdgpu;
  echo "## tools/diag_dtypes.py blosclz"; timeout -k 10 400 python tools/diag_dtypes.py blosclz 2>&1 | grep -v amdgpu;
  echo "## CIMG_ENC_RT=1 tools/diag_dtypes.py lz4   (the register-table form of the encoder, experimental: encode_rt_kernel.h)"; CIMG_ENC_RT=1 timeout -k 10 400 python tools/diag_dtypes.py lz4 2>&1 | grep -v amdgpu; } > $out/dtypes.txt
echo "[extra] dtypes done"
timeout -k 5 60 tests/ubench/regtab > $out/ubench_regtab.txt 2>&1
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rehearse_nccl.py 2>&1 | grep "nccl rehearsal" > $out/nccl_rehearsal.txt
echo "[extra] ubench + rccl rehearsal done"
