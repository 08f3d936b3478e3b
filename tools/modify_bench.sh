#!/bin/bash
# builds tools/modify_bench.cpp against libcimg_hip.so and runs it with double-buffered and with serial iterator windows
set -e
cd "$(dirname "$0")/.."
g++ -std=c++20 -O2 -I include -I compressed-image_amd/include tools/modify_bench.cpp -o tools/modify_bench -L compressed-image_amd -lcimg_hip -Wl,-rpath,$PWD/compressed-image_amd -pthread
echo "double-buffered:"; tools/modify_bench
echo "serial (CIMG_ITERATOR_SERIAL=1):"; CIMG_ITERATOR_SERIAL=1 tools/modify_bench
