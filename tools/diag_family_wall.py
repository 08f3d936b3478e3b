"""Wall time of the two device-resident batch calls, per family (zero / random / tiled / natural): what a step costs beyond its
kernels.   python3 tools/diag_family_wall.py [family ...]      CIMG_DIAG_DTYPE=uint16 CIMG_DIAG_CODEC=0 CIMG_DIAG_FILTER=2 CIMG_DIAG_CHUNK=<bytes> vary the rest."""
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
eng = hip.Engine(0)
dt = np.dtype(os.environ.get('CIMG_DIAG_DTYPE', 'float16'))
codec = int(os.environ.get('CIMG_DIAG_CODEC', '1'))
filt = int(os.environ.get('CIMG_DIAG_FILTER', '1'))
for fam in sys.argv[1:] or ["zero", "random", "tiled"]:
    chans = [synth.zero_channel(dt, 4096, 4096) if fam == "zero" else getattr(synth, fam + "_channel")(dt, 4096, 4096, c=c) for c in range(4)]
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    chunk = int(os.environ.get('CIMG_DIAG_CHUNK', str(4 * 1024 * 1024)))
    n = host.size // chunk * chunk
    nchunks, stride = n // chunk, (chunk + 32 + 63) // 64 * 64
    d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride + 64)
    d_raw.upload(host[:n] if n == host.size else np.ascontiguousarray(host[:n]))
    raw_off = np.arange(nchunks, dtype=np.int64) * chunk; comp_off = np.arange(nchunks, dtype=np.int64) * stride
    p = hip.cparams(dt.itemsize, compcode=codec, filters=(0, 0, 0, 0, 0, filt))
    for _ in range(5):
        cb = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
        eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    t1 = time.perf_counter()
    for _ in range(K): eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    t2 = time.perf_counter()
    eng.enable_timing(True)
    out = {}
    for name, call in (("compress", lambda: eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)),
                       ("decompress", lambda: eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off))):
        eng.reset_timing()
        for _ in range(10): call()
        out[name] = {k: (eng.kernel_time(k)[1] // 10, round(eng.kernel_time(k)[0] / 10 * 1000, 1)) for k in range(7) if eng.kernel_time(k)[1]}
    eng.enable_timing(False)
    print(fam, dt.name, "codec", codec, "filter", filt, "compressed bytes", int(np.sum(cb)), "chunks", nchunks, "wall per call us: compress %.1f decompress %.1f; kernels by id (launches, us per call):" % ((t1 - t0) / K * 1e6, (t2 - t1) / K * 1e6), out)
os._exit(0)
