#!/bin/bash
# the GPU parity tests under every documented engine switch: alternate launch shapes must give the same bytes
for e in "CIMG_NO_LEAN=1" "CIMG_NO_ASSEMBLE_IN_LAUNCH=1" "CIMG_ENC_BLOCK_ITEMS=0" "CIMG_ENC_BLOCK_ITEMS=1" "CIMG_ENC_GANG=1" "CIMG_ENC_GANG=3" "CIMG_ENC_WGS_PER_CU=1" "CIMG_LEAN_WGS_PER_CU=3" "CIMG_SYNC_SPIN=1" "CIMG_ENC_HYBRID=0" "CIMG_NO_SIDE_STREAM=1" "CIMG_ENC_WHOLE_ROUNDS=0" "CIMG_ENC_WHOLE_ROUNDS=3" "CIMG_ZSTD_FUSED=1" "CIMG_ZSTD_LANES=0" "CIMG_ZSTD_LANES=5" "CIMG_ZSTD_PLAN_CAP=1024" "CIMG_ZSTD_WALK_STAGE=8192" "CIMG_ZSTD_PLAN_MIB=4" "CIMG_ZSTD_PLAN_FAIL=1" "CIMG_ENC_RT=1"; do
  echo "== $e: $(env $e timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
done
