"""Device-resident compress / decompress calls over batch sizes (chunks of 4 MiB, tiled float16): wall time per call and the
kernels' share -- cliffs in either say where a launch shape or a host step does not scale."""
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
eng = hip.Engine(0)
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
chunk = 4 * 1024 * 1024
maxc = 256
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
base = np.concatenate([c.view(np.uint8).ravel() for c in chans])            # 128 MiB = 32 chunks
host = np.tile(base, maxc * chunk // base.size)
stride = chunk + 64
d_raw, d_out, d_comp = eng.alloc(host.size), eng.alloc(host.size), eng.alloc(maxc * stride)
d_raw.upload(host)
p = hip.cparams(2)
for nchunks in [int(x) for x in os.environ.get("CIMG_DIAG_SIZES", "1,2,4,8,16,32,64,128,256").split(",")]:
    raw_off = np.arange(nchunks, dtype=np.int64) * chunk; comp_off = np.arange(nchunks, dtype=np.int64) * stride
    for _ in range(3):
        eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
        eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    t1 = time.perf_counter()
    for _ in range(K): eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    t2 = time.perf_counter()
    eng.enable_timing(True); eng.reset_timing()
    for _ in range(5):
        eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
        eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    ke, kd = eng.kernel_time(hip.K_ENCODE)[0] / 5 * 1000, eng.kernel_time(hip.K_DECODE)[0] / 5 * 1000
    kl, km = eng.kernel_time(hip.K_LAYOUT)[0] / 5 * 1000, eng.kernel_time(hip.K_EMIT)[0] / 5 * 1000
    eng.enable_timing(False)
    n = nchunks * chunk
    cw, dw = (t1 - t0) / K * 1e6, (t2 - t1) / K * 1e6
    print("%4d chunks (%5d MiB): compress wall %8.1f us (kernels: encode %8.1f layout %5.1f emit %5.1f) %6.1f GB/s | decompress wall %7.1f us (kernel %7.1f) %7.1f GB/s"
          % (nchunks, n >> 20, cw, ke, kl, km, n / cw / 1e3, dw, kd, n / dw / 1e3))
os._exit(0)
