"""Where the time of the two-wave lean decode (CIMG_LEAN_PAIR=1) goes: per block, cycles of wave 0 in compute / post, of
wave 1 in consume, and of both in the two barriers of a step."""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(60, exit=True)
from cimg import hip, synth
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
for _ in range(3):
    cb = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
st = eng.read_stamps(1).astype(np.float64)
a, b = st[:, 4:8], st[:, 8:12]
print(fam, "blocks", len(st), "steps per block %.1f" % a[:, 3].mean())
print("   wave 0 (producer): compute+post %.0f  (-) %.0f  barrier %.0f cycles per block; per step %.0f / %.0f / %.0f" % (
    a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), (a[:, 0] / a[:, 3]).mean(), (a[:, 1] / a[:, 3]).mean(), (a[:, 2] / a[:, 3]).mean()))
print("   wave 1 (consumer): consume %.0f  barriers %.0f cycles per block; per step %.0f / %.0f" % (
    b[:, 0].mean(), b[:, 2].mean(), (b[:, 0] / b[:, 3]).mean(), (b[:, 2] / b[:, 3]).mean()))
print("   producer detail (CIMG_PAIR_PROF builds): parse %.0f  post %.0f  scalar steps %.0f cycles per block" % (st[:, 5].mean(), st[:, 2].mean(), st[:, 3].mean()))
os._exit(0)
