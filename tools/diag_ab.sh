#!/bin/bash
# one GPU visit per experiment: small correctness repro, encoder cycle profile, then kernel times
set -e
timeout -k 3 30 python tools/diag_variant.py 2>&1 | grep -vE "^  File|^$|Thread|amdgpu.ids" | tail -2
timeout -k 5 100 python tools/diag_encprof.py tiled 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 bash tools/diag_bench.sh 2>&1 | grep value
