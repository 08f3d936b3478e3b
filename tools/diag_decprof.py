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
d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(2):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
st = eng.read_stamps(1).astype(np.float64)
names = ["scalar rest", "batch parse", "chain walk", "scan+literals", "batch matches", "scalar header", "scalar copy", "parallel matches"]
cyc = st[:, :8].mean(axis=0); cnt = st[:, 8:16].mean(axis=0)
print(fam, "blocks", len(st), "(last LZ4 stream of each block) total cycles %.0f" % cyc.sum())
print("   " + "  ".join("%s=%.0f" % (nm, c) for nm, c in zip(names[:8], cyc[:8])))
print("   batch matches %.1f  batches %.1f  scalar-path sequences %.1f (of them longer than 64: %.1f)" % tuple(cnt[:4]))
if cnt[1] > 0:
    print("   per batch: parse %.0f  walk %.0f  scan+literals %.0f  matches %.0f (%.0f per match)" % (cyc[1] / cnt[1], cyc[2] / cnt[1], cyc[3] / cnt[1], cyc[4] / cnt[1], cyc[4] / max(cnt[0], 1)))
if cnt[2] > 0:
    print("   per scalar-path sequence %.0f" % (cyc[0] / cnt[2]))
os._exit(0)
