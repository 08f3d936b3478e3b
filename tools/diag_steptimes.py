"""Per-step wall times of the bench's step (compress _begin, decompress _begin, both _fetch): where does a long run lose time?"""
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
eng = hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks, dtype=np.int64) * chunk; comp_off = np.arange(nchunks, dtype=np.int64) * stride
nb = np.full(nchunks, chunk, np.int32); ds = np.full(nchunks, chunk + 32, np.int32); bs = np.full(nchunks, 32768, np.int32)
p = hip.cparams(2)
timing = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng.enable_timing(timing)
def step():
    k = eng.compress_device_begin(p, d_raw.ptr, raw_off, nb, d_comp.ptr, comp_off, ds)
    eng.decompress_device_begin(d_comp.ptr, comp_off, nb, bs, d_out.ptr, raw_off)
    eng.compress_device_fetch(k); eng.decompress_device_fetch(k)
for _ in range(10): step()
ts = []
for i in range(600):
    t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print("timing period", timing, "median %.0f us  p90 %.0f  max %.0f at step %d; steps above 2 ms: %s" % (np.median(ts), np.percentile(ts, 90), ts.max(), ts.argmax(), [(int(i), int(ts[i])) for i in np.nonzero(ts > 2000)[0]]))
print("  sum %.1f ms; sum without outliers %.1f ms" % (ts.sum() / 1e3, ts[ts < 2000].sum() / 1e3))
