# Device deinterleave kernel: microseconds and GB/s (read N + write N) for the shapes an image reader produces.  Diagnostics only.
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip
eng = hip.Engine(0)
for nch, ts, w, h in ((4, 2, 4096, 4096), (3, 2, 4096, 4096), (4, 4, 4096, 4096), (4, 1, 4096, 4096), (3, 4, 8192, 8192)):
    npix = w * h
    raw = np.random.default_rng(1).integers(0, 256, npix * nch * ts, dtype=np.uint8)
    stride = (npix * ts + 15) & ~15
    d_in, d_out = eng.alloc(raw.size), eng.alloc(stride * nch)
    d_in.upload(raw)
    for _ in range(3): eng.deinterleave_device(d_in.ptr, nch, ts, npix, d_out.ptr, stride)
    eng.synchronize()
    eng.enable_timing(True); eng.reset_timing()
    for _ in range(20): eng.deinterleave_device(d_in.ptr, nch, ts, npix, d_out.ptr, stride)
    eng.synchronize()
    ms, n = eng.kernel_time(4)
    eng.enable_timing(False)
    us = ms / max(n, 1) * 1e3
    print("%d channels x %d-byte elements, %dx%d: %.1f us per launch (%d timed), %.0f GB/s (N read + N written)" % (nch, ts, w, h, us, n, 2 * raw.size / us / 1e3))
    d_in.free(); d_out.free()
os._exit(0)
