"""Chunks that are no multiple of the block size (every image whose row size does not divide 4 MiB: a 1920- or 3840-pixel float16
row gives chunks of 127 full blocks + one of 31744 bytes): device-resident compress / decompress batches of configs[1]'s pixels cut
that way, wall time per call and kernel times (LABNOTES.md, round 4: assembling such chunks inside the two launches was tried)."""
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
eng = hip.Engine(0)
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
chunk = 127 * 32768 + 31744
nch = host.size // chunk
host = host[:nch * chunk]
stride = (chunk + 32 + 63) // 64 * 64
d_raw, d_comp, d_out = eng.alloc(host.size), eng.alloc(nch * stride), eng.alloc(host.size)
d_raw.upload(host)
roff = np.arange(nch, dtype=np.int64) * chunk; coff = np.arange(nch, dtype=np.int64) * stride
p = hip.cparams(2)
for _ in range(3):
    cb = eng.compress_device(p, d_raw.ptr, roff, [chunk] * nch, d_comp.ptr, coff, [chunk + 32] * nch)
    eng.decompress_device(d_comp.ptr, coff, [chunk] * nch, [32768] * nch, d_out.ptr, roff)
assert d_out.download(host.size).tobytes() == host.tobytes()
reps = 30
t0 = time.perf_counter()
for _ in range(reps): eng.compress_device(p, d_raw.ptr, roff, [chunk] * nch, d_comp.ptr, coff, [chunk + 32] * nch)
t1 = time.perf_counter()
for _ in range(reps): eng.decompress_device(d_comp.ptr, coff, [chunk] * nch, [32768] * nch, d_out.ptr, roff)
t2 = time.perf_counter()
eng.enable_timing(True); eng.reset_timing()
for _ in range(5):
    eng.compress_device(p, d_raw.ptr, roff, [chunk] * nch, d_comp.ptr, coff, [chunk + 32] * nch)
    eng.decompress_device(d_comp.ptr, coff, [chunk] * nch, [32768] * nch, d_out.ptr, roff)
k = {n: eng.kernel_time(i) for i, n in enumerate(("encode", "layout", "emit", "decode"))}
print("%s, %d chunks of %d bytes (127 blocks + 31744), %s: compress %.1f us per batch, decompress %.1f us; kernels (timed one by one, serialising the side launch): %s" % (
    fam, nch, chunk, "assembly behind the launches",
    (t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6, {n: "%d x %.1f us" % (c, ms / max(c, 1) * 1e3) for n, (ms, c) in k.items()}))
os._exit(0)
