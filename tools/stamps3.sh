set -e
for t in 64 128; do echo "== lean threads $t"; CIMG_LEAN_THREADS=$t CIMG_VERBOSE=1 timeout -k 10 120 python tools/diag_stamps.py tiled 2>&1 | grep -v "^\[cimg\] \(enc\|comp\)"; done
echo "== pair"; CIMG_LEAN_PAIR=1 CIMG_VERBOSE=1 timeout -k 10 120 python tools/diag_stamps.py tiled 2>&1 | grep -v "^\[cimg\] \(enc\|comp\)"
