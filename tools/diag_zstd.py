# Throughput of the zstd read path (csrc/zstd_kernel.h, the slow path): 4096 x 4096 float16 tiled channel(s) coded chunk by
# chunk with the box's libzstd the way c-blosc2 frames them (tests/golden/make_zstd_golden.py), decoded by one batch call.
import sys, os, time, ctypes as C
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "tests", "golden")]
import numpy as np
from cimg import hip, synth
import make_zstd_golden as G
z = C.CDLL("libzstd.so.1")
z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
z.ZSTD_compress.restype = C.c_size_t; z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
z.ZSTD_isError.argtypes = [C.c_size_t]
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = hip.Engine(0)
for fam, clevel in (("tiled", 3), ("tiled", 9), ("natural", 3)):
    chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(nch)]
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    chunk = 4 * 1024 * 1024
    t0 = time.perf_counter()
    chunks = [G.frame(z, host[i:i + chunk], 2, 32768, clevel) for i in range(0, host.size, chunk)]
    t_enc = time.perf_counter() - t0
    csize = sum(len(c) for c in chunks)
    outs, st = eng.decompress_host(chunks)                      # warm (and check)
    assert not st.any() and b"".join(o.tobytes() for o in outs) == host.tobytes()
    t0 = time.perf_counter()
    for _ in range(3): eng.decompress_host(chunks)
    dt = (time.perf_counter() - t0) / 3
    # device-resident: the kernels alone (HIP events of the engine; the zstd launch is a decode launch of its own)
    sizes = [len(c) for c in chunks]
    coff = np.concatenate([[0], np.cumsum([(n + 63) & ~63 for n in sizes[:-1]])]).astype(np.int64)
    blob = np.zeros(int(coff[-1]) + sizes[-1] + 64, np.uint8)
    for o, c in zip(coff, chunks): blob[o:o + len(c)] = np.frombuffer(c, np.uint8)
    d_comp, d_out = eng.alloc(blob.size), eng.alloc(host.size)
    d_comp.upload(blob)
    nb = [chunk] * len(chunks); roff = np.arange(len(chunks), dtype=np.int64) * chunk
    eng.decompress_device(d_comp.ptr, coff, nb, [32768] * len(chunks), d_out.ptr, roff)
    eng.enable_timing(True); eng.reset_timing()
    t0 = time.perf_counter()
    for _ in range(3): eng.decompress_device(d_comp.ptr, coff, nb, [32768] * len(chunks), d_out.ptr, roff)
    wall = (time.perf_counter() - t0) / 3
    ms, k = eng.kernel_time(hip.K_DECODE_ZSTD)
    eng.enable_timing(False)
    print("   device-resident: wall %.1f ms per batch; decode kernels %.1f ms per batch over %d timed launches" % (wall * 1e3, ms / 3, k))
    lz = eng.compress_host(hip.cparams(2), host, [chunk] * len(chunks), [chunk + 32] * len(chunks))
    eng.decompress_host(lz)
    t0 = time.perf_counter()
    for _ in range(3): eng.decompress_host(lz)
    print("   the same pixels as lz4 chunks through the same host call: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
    print("%s clevel %d (%s): %d chunks, ratio %.2f; libzstd compress (python loop, 1 thread) %.2f s; python helper decompress_host (joins and allocates on the host, pageable PCIe) %.1f ms = %.2f GB/s"
          % (fam, clevel, "split planes" if clevel <= 5 else "one stream per block", len(chunks), host.size / csize, t_enc, dt * 1e3, host.size / dt / 1e9))
os._exit(0)
