"""PCIe-inclusive rate of the host-buffer C ABI (cimg_*_batch_host) and of the single-chunk blosc2 shim on config 2."""
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
eng = hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
sizes = [chunk] * (n // chunk)
p = hip.cparams(2)
for _ in range(2):
    chunks = eng.compress_host(p, host, sizes, [chunk + 32] * len(sizes))
    outs, st = eng.decompress_host(chunks)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    chunks = eng.compress_host(p, host, sizes, [chunk + 32] * len(sizes))
t1 = time.perf_counter()
for _ in range(reps):
    outs, st = eng.decompress_host(chunks)
t2 = time.perf_counter()
assert b"".join(o.tobytes() for o in outs) == host.tobytes()
enc, dec = n * reps / (t1 - t0) / 1e9, n * reps / (t2 - t1) / 1e9
print("host-buffer batch API (pageable numpy buffers, python wrapper included): compress %.2f GB/s  decompress %.2f GB/s  combined %.2f GB/s" % (enc, dec, 2 * n * reps / (t2 - t0) / 1e9))
# single-chunk shim: one blosc2_compress_ctx per 4 MiB chunk, as the unmodified reference would call it
L = hip.load()
cp = hip.Blosc2CParams(); cp.compcode, cp.clevel, cp.typesize, cp.nthreads, cp.blocksize, cp.splitmode = 1, 9, 2, 4, 32768, 3; cp.filters[5] = 1
cctx = L.blosc2_create_cctx(cp)
dp = hip.Blosc2DParams(); dp.nthreads = 1
dctx = L.blosc2_create_dctx(dp)
dst = np.zeros(chunk + 32, np.uint8); out = np.zeros(chunk, np.uint8)
t0 = time.perf_counter()
for i in range(len(sizes)):
    r = L.blosc2_compress_ctx(cctx, host[i * chunk:].ctypes.data, chunk, dst.ctypes.data, dst.size)
    L.blosc2_decompress_ctx(dctx, dst.ctypes.data, 2**31 - 1, out.ctypes.data, out.size)
t1 = time.perf_counter()
print("blosc2 shim, one chunk per call: compress+decompress %.2f GB/s" % (2 * n / (t1 - t0) / 1e9))
eng.close()
# raw C ABI timing (no python allocations inside the timed region)
import ctypes as C
eng = hip.Engine(0)
nch = len(sizes)
raw_off = np.arange(nch, dtype=np.int64) * chunk
stride = chunk + 64
comp_off = np.arange(nch, dtype=np.int64) * stride
nb = np.full(nch, chunk, np.int32); ds = np.full(nch, chunk + 32, np.int32); cb = np.zeros(nch, np.int32)
comp = np.zeros(nch * stride, np.uint8); back = np.zeros(n, np.uint8); st = np.zeros(nch, np.int32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(2):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(host), P(raw_off), P(nb), P(comp), P(comp_off), P(ds), P(cb))
    L.cimg_decompress_batch_host(eng.handle, nch, P(comp), P(comp_off), P(back), P(raw_off), P(nb), P(st))
t0 = time.perf_counter()
for _ in range(reps):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(host), P(raw_off), P(nb), P(comp), P(comp_off), P(ds), P(cb))
t1 = time.perf_counter()
for _ in range(reps):
    L.cimg_decompress_batch_host(eng.handle, nch, P(comp), P(comp_off), P(back), P(raw_off), P(nb), P(st))
t2 = time.perf_counter()
assert back.tobytes() == host.tobytes()
print("C ABI host-buffer batch calls (pageable): compress %.2f GB/s (%.1f ms)  decompress %.2f GB/s (%.1f ms)  combined %.2f GB/s" % (
    n * reps / (t1 - t0) / 1e9, (t1 - t0) / reps * 1e3, n * reps / (t2 - t1) / 1e9, (t2 - t1) / reps * 1e3, 2 * n * reps / (t2 - t0) / 1e9))
eng.close()

# the same from / to page-locked memory (cimg_host_malloc): what PCIe itself allows
eng = hip.Engine(0)
L.cimg_host_malloc.restype = C.c_void_p
L.cimg_host_malloc.argtypes = [C.c_size_t]
def pinned(nbytes):
    ptr = L.cimg_host_malloc(nbytes)
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))
ph, pc, pb = pinned(n), pinned(nch * stride), pinned(n)
ph[:] = host
for _ in range(2):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(ph), P(raw_off), P(nb), P(pc), P(comp_off), P(ds), P(cb))
    L.cimg_decompress_batch_host(eng.handle, nch, P(pc), P(comp_off), P(pb), P(raw_off), P(nb), P(st))
t0 = time.perf_counter()
for _ in range(reps):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(ph), P(raw_off), P(nb), P(pc), P(comp_off), P(ds), P(cb))
t1 = time.perf_counter()
for _ in range(reps):
    L.cimg_decompress_batch_host(eng.handle, nch, P(pc), P(comp_off), P(pb), P(raw_off), P(nb), P(st))
t2 = time.perf_counter()
assert pb.tobytes() == host.tobytes()
print("C ABI host-buffer batch calls (page-locked): compress %.2f GB/s (%.1f ms)  decompress %.2f GB/s (%.1f ms)  combined %.2f GB/s" % (
    n * reps / (t1 - t0) / 1e9, (t1 - t0) / reps * 1e3, n * reps / (t2 - t1) / 1e9, (t2 - t1) / reps * 1e3, 2 * n * reps / (t2 - t0) / 1e9))
eng.close()
