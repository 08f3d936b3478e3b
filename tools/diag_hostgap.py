# Where does the host's share of a round trip go?  Wall time of each of the four calls of bench.py's step.  Diagnostics only.
import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
E_ = hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
raw_off = np.arange(nchunks, dtype=np.int64) * chunk; comp_off = np.arange(nchunks, dtype=np.int64) * stride
sizes, dest, bs = np.full(nchunks, chunk, np.int32), np.full(nchunks, chunk + 32, np.int32), np.full(nchunks, 32768, np.int32)
p = hip.cparams(2)
d_raw, d_comp, d_out = E_.alloc(n), E_.alloc(nchunks * stride), E_.alloc(n)
d_raw.upload(host)
T = np.zeros(5)
K = 300
for it in range(K + 20):
    t0 = time.perf_counter()
    k = E_.compress_device_begin(p, d_raw.ptr, raw_off, sizes, d_comp.ptr, comp_off, dest); t1 = time.perf_counter()
    E_.decompress_device_begin(d_comp.ptr, comp_off, sizes, bs, d_out.ptr, raw_off); t2 = time.perf_counter()
    E_.compress_device_fetch(k); t3 = time.perf_counter()
    E_.decompress_device_fetch(k); t4 = time.perf_counter()
    if it >= 20: T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
print("per step, us: compress begin %.1f, decompress begin %.1f, compress fetch (wait) %.1f, decompress fetch %.1f, total %.1f" % tuple(T / K * 1e6))
os._exit(0)
