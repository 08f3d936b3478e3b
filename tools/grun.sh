#!/bin/bash
# build everything that travels (the .so files are not rebuilt on the GPU box), then run a command there:  tools/grun.sh [timeout] '<command>'
set -e
cd "$(dirname "$0")/.."
t=900
if [[ "$1" =~ ^[0-9]+$ ]]; then t=$1; shift; fi
python -c "import __graft_entry__ as g; g.build()" > /tmp/cimg_build.log 2>&1 || { tail -30 /tmp/cimg_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
