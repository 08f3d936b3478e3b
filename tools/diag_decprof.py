"""-DCIMG_PROFILE build (make -C compressed-image_amd prof; CIMG_LIB=gpurun_in/libcimg_hip_prof.so): cycle laps of the LZ4 decoder
inside the lean decode launch, per block.  Laps (decode_kernel.h: CIMG_PROF_LAP): 0 scalar-path work, 1 per-lane header parse,
2 token chain walk, 3 positions + literals, 7 lane-parallel matches, 4 in-order matches of a batch, 5 scalar-path header, 6 scalar-path copy.
Counts: 0 in-order matches inside batches, 1 batches, 2 scalar-path sequences, 3 of those longer than 64 bytes."""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(90, exit=True)
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
for _ in range(2):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
st = eng.read_stamps(1).astype(np.float64)
st = st[st[:, :8].sum(axis=1) > 0]
names = ["scalar-path work", "lane header parse", "token chain walk", "positions + literals", "in-order matches", "scalar-path header", "scalar-path copy", "lane-parallel matches"]
tot = st[:, :8].sum(axis=1)
print(fam, "blocks with laps:", len(st), " cycles per block: mean %.0f p50 %.0f  (= %.1f us at 2.3 GHz)" % (tot.mean(), np.median(tot), tot.mean() / 2300))
for k in (1, 2, 3, 7, 4, 0, 5, 6):
    print("   lap %d %-24s mean %8.0f cycles  %5.1f %%" % (k, names[k], st[:, k].mean(), 100 * st[:, k].mean() / tot.mean()))
cn = ["in-order matches in batches", "batches", "scalar-path sequences", "scalar-path matches > 64 B"]
for k in range(4):
    print("   count %d %-28s mean %.1f" % (k, cn[k], st[:, 8 + k].mean()))
