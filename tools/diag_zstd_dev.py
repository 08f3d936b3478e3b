"""Development benchmark of the zstd READ path (cimg_decode_zstd) on chunks the box's libzstd writes the way the reference would
(clevel 9 = zstd level 22, one frame per 32 KiB block; made through the checker's chunk layer, outside anything timed):
4096 x 8192 float32 (128 MiB, 4096 blocks) of the tiled and the natural family.  Prints kernel ms, GB/s and the check.
usage: python tools/diag_zstd_dev.py [tiled|natural ...] [--clevel N] [--mib M] [--dtype float32|uint16|uint8|float16]"""
import sys, os, time, ctypes as C
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
import _oracle as O
from cimg import hip, synth
args = [a for a in sys.argv[1:] if a in ("tiled", "natural", "zero", "random")]
clevel = int(sys.argv[sys.argv.index("--clevel") + 1]) if "--clevel" in sys.argv else 9
mib = int(sys.argv[sys.argv.index("--mib") + 1]) if "--mib" in sys.argv else 128
dtype = np.dtype(sys.argv[sys.argv.index("--dtype") + 1]) if "--dtype" in sys.argv else np.dtype(np.float32)
fams = [a for a in args] or ["tiled", "natural"]
L = O.lib()
L.orc_bench_compress.argtypes = [C.POINTER(O.CParams), C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_int]
L.orc_bench_compress.restype = C.c_int64
eng = hip.Engine(0)
CHUNK = 4 << 20
for fam in fams:
    W, H = 4096, mib * (1 << 20) // (4096 * dtype.itemsize)
    host = np.ascontiguousarray(getattr(synth, fam + "_channel")(dtype.type, W, H)).view(np.uint8).ravel()
    n = host.size; nch = n // CHUNK; stride = CHUNK + 64
    comp = np.zeros(nch * stride, np.uint8); cb = np.zeros(nch, np.int32)
    p = O.cparams(dtype.itemsize, clevel=clevel, blocksize=32768, compcode=O.ZSTD)
    cores = min(len(os.sched_getaffinity(0)), 64)
    t0 = time.perf_counter()
    r = L.orc_bench_compress(C.byref(p), host.ctypes.data, nch, CHUNK, comp.ctypes.data, stride, CHUNK + 32, cb.ctypes.data, min(cores, nch), max(1, cores // min(cores, nch)))
    assert r > 0
    t_make = time.perf_counter() - t0
    d_comp, d_out = eng.alloc(comp.size), eng.alloc(n)
    d_comp.upload(comp)
    coff = np.arange(nch, dtype=np.int64) * stride; roff = np.arange(nch, dtype=np.int64) * CHUNK
    eng.decompress_device(d_comp.ptr, coff, [CHUNK] * nch, [32768] * nch, d_out.ptr, roff, comp_size=cb)
    ok = d_out.download(n).tobytes() == host.tobytes()
    eng.enable_timing(True); eng.reset_timing()
    for _ in range(3): eng.decompress_device(d_comp.ptr, coff, [CHUNK] * nch, [32768] * nch, d_out.ptr, roff, comp_size=cb)
    ms, k = eng.kernel_time(hip.K_DECODE_ZSTD)
    eng.enable_timing(False)
    if not k:
        print("%s %s clevel %d: no chunk went through the zstd read path (stored chunks: ratio %.2f)" % (fam, dtype.name, clevel, n / float(cb.sum())))
        d_comp.free(); d_out.free()
        continue
    parts = []
    for kid in (hip.K_ZSTD_WALK, hip.K_ZSTD_LIT, hip.K_ZSTD_SEQ, hip.K_ZSTD_REPLAY, hip.K_ZSTD_FUSED):
        pm, pk = eng.kernel_time(kid)
        if pk: parts.append("%s %.2f ms" % (hip.KERNELS[kid], pm / pk))
    print("   " + ", ".join(parts))
    print(("%s " + dtype.name + " %d MiB, libzstd clevel %d (ratio %.2f, made in %.1f s): cimg_decode_zstd %.2f ms = %.1f GB/s, pixels %s") % (
        fam, mib, clevel, n / float(cb.sum()), t_make, ms / k, n / (ms / k * 1e-3) / 1e9, "bit-exact" if ok else "DIFFER"))
    d_comp.free(); d_out.free()
eng.close()
if os.environ.get("CIMG_DIAG_EXIT") == "clean":      # (under rocprofv3: an abrupt exit loses the profiler's output)
    sys.exit(0)
os._exit(0)
