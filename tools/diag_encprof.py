import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(60, exit=True)
from cimg import hip, synth
hip.LIB_PATH = os.path.join(os.getcwd(), "gpurun_in", "libcimg_hip_prof.so")
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
eng = hip.Engine(0)
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_comp = eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(2):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
eng.debug_stamps(True)
eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
st = eng.read_stamps(0).astype(np.float64)
print(fam, "items", len(st))
names = ["clear+first", "pos+vread", "runpath", "window", "extend", "emit", "lastlit", "narrow"]
total_blocks = len(st) // 2
for plane, sl in (("plane1 (high byte)", slice(0, total_blocks)), ("plane0 (low byte)", slice(total_blocks, None))):
    s = st[sl]
    cyc = s[:, :8].mean(axis=0); cnt = s[:, 8:16].mean(axis=0)
    print(" ", plane, "total cycles %.0f" % cyc.sum(), " counts: runpath %.1f windows %.1f extends %.1f" % (cnt[0], cnt[1], cnt[2]))
    print("    " + "  ".join("%s=%.0f" % (nm, c) for nm, c in zip(names, cyc)))
    print("    extend blocks run %.1f  long-extend iterations %.1f  run-path zero-literal hits %.1f  narrow-path hits %.1f" % (cnt[3], cnt[4], cnt[5], cnt[7]))
    print("    plane load + run check %.0f cyc" % cnt[6])
    if cnt[0] > 0: print("    per runpath %.0f cyc" % (cyc[2] / cnt[0]))
    if cnt[1] > 0: print("    per window  %.0f cyc (pos+vread per iteration %.0f)" % (cyc[3] / cnt[1], cyc[1] / (cnt[0] + cnt[1])))
    if cnt[2] > 0: print("    per extend  %.0f cyc, per emit %.0f cyc" % (cyc[4] / cnt[2], cyc[5] / max(cnt[0] + cnt[2] - cnt[0], 1)))
os._exit(0)
