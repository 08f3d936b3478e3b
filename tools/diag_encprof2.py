"""-DCIMG_PROFILE build (gpurun_in/libcimg_hip_prof.so, or CIMG_PROF_LIB): raw cycle laps and counts of the LZ4 encoder per byte plane,
any dtype / family.  usage: python tools/diag_encprof2.py [family] [dtype]
bins: 0 chain-miss exit (+clear) | 1 head entry | 2 run path | 3 window | 4 extend | 5 park+emit | 6 last literals | 7 chain hit / narrow
counts: 0 run path | 1 windows | 2 sequences | 3 extends run | 4 chain misses (+long extends) | 5 run-path zero-lit | 6 plane load cycles | 7 chain hits + narrow hits"""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(120, exit=True)
from cimg import hip, synth
hip.LIB_PATH = os.environ.get("CIMG_PROF_LIB", os.path.join(os.getcwd(), "gpurun_in", "libcimg_hip_prof.so"))
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
dt = np.dtype(sys.argv[2] if len(sys.argv) > 2 else "float16")
eng = hip.Engine(0)
host = np.ascontiguousarray(getattr(synth, fam + "_channel")(dt.type, 4096, (32 << 20) // (4096 * dt.itemsize))).view(np.uint8).ravel()
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_comp = eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(dt.itemsize)
for _ in range(2):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
eng.debug_stamps(True)
eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
st = eng.read_stamps(0).astype(np.float64)
ts = dt.itemsize
per = len(st) // ts
print(fam, dt.name, "items", len(st))
for k in range(ts):
    s = st[k * per:(k + 1) * per]
    cyc = s[:, :8].mean(axis=0); cnt = s[:, 8:16].mean(axis=0)
    print("  plane %d: total %.0f cycles | laps %s | counts %s" % (ts - 1 - k, cyc.sum(), " ".join("%.0f" % c for c in cyc), " ".join("%.1f" % c for c in cnt)))
    if cnt[7] > 0: print("     bin7 per count7: %.0f cycles; bin0 per count4: %.0f; (bin1+bin4+bin5) per sequence: %.0f" % (cyc[7] / cnt[7], cyc[0] / max(cnt[4], 1), (cyc[1] + cyc[4] + cyc[5]) / max(cnt[2], 1)))
os._exit(0)
