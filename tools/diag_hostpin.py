"""Host-buffer C ABI from PAGEABLE memory with and without page-locking the caller's span for the call (engine.hip: HostPin,
CIMG_HOST_REGISTER_MIB): configs[1], raw C calls, no Python allocation inside the timed region.  Run once per setting:
    CIMG_HOST_REGISTER_MIB=0 python tools/diag_hostpin.py ; CIMG_HOST_REGISTER_MIB=32 python tools/diag_hostpin.py"""
import sys, os, time, ctypes as C
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
L = hip.load()
eng = hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nch = n // chunk
p = hip.cparams(2)
raw_off = np.arange(nch, dtype=np.int64) * chunk
stride = chunk + 64
comp_off = np.arange(nch, dtype=np.int64) * stride
nb = np.full(nch, chunk, np.int32); ds = np.full(nch, chunk + 32, np.int32); cb = np.zeros(nch, np.int32)
comp = np.zeros(nch * stride, np.uint8); back = np.zeros(n, np.uint8); st = np.zeros(nch, np.int32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(2):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(host), P(raw_off), P(nb), P(comp), P(comp_off), P(ds), P(cb))
    L.cimg_decompress_batch_host(eng.handle, nch, P(comp), P(comp_off), P(back), P(raw_off), P(nb), P(st))
reps = 8
t0 = time.perf_counter()
for _ in range(reps):
    L.cimg_compress_batch_host(eng.handle, C.byref(p), nch, P(host), P(raw_off), P(nb), P(comp), P(comp_off), P(ds), P(cb))
t1 = time.perf_counter()
for _ in range(reps):
    L.cimg_decompress_batch_host(eng.handle, nch, P(comp), P(comp_off), P(back), P(raw_off), P(nb), P(st))
t2 = time.perf_counter()
assert back.tobytes() == host.tobytes()
print("CIMG_HOST_REGISTER_MIB=%s pageable host buffers: compress %.2f GB/s (%.2f ms)  decompress %.2f GB/s (%.2f ms)  combined %.2f GB/s" % (
    os.environ.get("CIMG_HOST_REGISTER_MIB", "default"), n * reps / (t1 - t0) / 1e9, (t1 - t0) / reps * 1e3, n * reps / (t2 - t1) / 1e9, (t2 - t1) / reps * 1e3, 2 * n * reps / (t2 - t0) / 1e9))
eng.close()
