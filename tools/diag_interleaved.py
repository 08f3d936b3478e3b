# The producer path: interleaved RGBA float16 4096x4096 in host memory -> compressed chunks in host memory.
# (a) cimg_compress_batch_host_interleaved_begin + _fetch (upload once, split on the device, compress from there)
# (b) numpy deinterleave on the host, then the pipelined cimg_compress_batch_host.   Diagnostics only.
import sys, os, time, ctypes as C
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
eng = hip.Engine(0)
nch, w, h, ts = 4, 4096, 4096, 2
planes = [synth.tiled_channel(np.float16, w, h, c=c).view(np.uint16).ravel() for c in range(nch)]
inter = np.ascontiguousarray(np.stack(planes, axis=1)).view(np.uint8).ravel()        # pixel-major
npix, chunk = w * h, 4 * 1024 * 1024
stride = (npix * ts + 15) & ~15
per = npix * ts // chunk
raw_off = np.array([c * stride + k * chunk for c in range(nch) for k in range(per)], np.int64)
nb = np.full(raw_off.size, chunk, np.int32); dest = np.full(raw_off.size, chunk + 32, np.int32); cb = np.zeros(raw_off.size, np.int32)
p = hip.cparams(ts)
L = hip.load()
def ptr(a): return a.ctypes.data_as(C.c_void_p)
out = np.zeros(raw_off.size * (chunk + 64), np.uint8)
def fused():
    rc = L.cimg_compress_batch_host_interleaved_begin(eng.handle, C.byref(p), nch, npix, ptr(inter), raw_off.size, ptr(raw_off), ptr(nb), ptr(dest), ptr(cb))
    assert rc == 0, rc
    comp_off = np.concatenate([[0], np.cumsum((cb[:-1].astype(np.int64) + 63) & ~63)]).astype(np.int64)
    rc = L.cimg_compress_batch_host_fetch(eng.handle, raw_off.size, ptr(out), ptr(comp_off))
    assert rc == 0, rc
    return comp_off
def host_split():
    planar = np.ascontiguousarray(inter.view(np.uint16).reshape(npix, nch).T).view(np.uint8).ravel()
    return eng.compress_host(p, planar, [chunk] * raw_off.size, [chunk + 32] * raw_off.size)
comp_off = fused(); chunks_b = host_split()
for i in range(raw_off.size):
    assert out[comp_off[i]:comp_off[i] + cb[i]].tobytes() == chunks_b[i], i
for name, fn in (("fused (device deinterleave)", fused), ("numpy deinterleave + host batch", host_split)):
    fn(); t = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t) / 5
    print("%-34s %.1f ms per image = %.1f GB/s of pixels (pageable host memory)" % (name, dt * 1e3, inter.size / dt / 1e9))
os._exit(0)
