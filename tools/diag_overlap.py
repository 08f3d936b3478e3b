# Do compress batches and decompress batches of DIFFERENT data overlap on the device when they come from two engines
# (two streams)?  Thread 1 compresses K times, thread 2 decompresses K times; compare with each alone.  Diagnostics only.
import sys, os, time, threading
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
K = 200
A, B = hip.Engine(0), hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
sizes, dest, bs = [chunk] * nchunks, [chunk + 32] * nchunks, [32768] * nchunks
p = hip.cparams(2)
d_raw, d_compA = A.alloc(n), A.alloc(nchunks * stride)
d_compB, d_out = B.alloc(nchunks * stride), B.alloc(n)
d_raw.upload(host)
A.compress_device(p, d_raw.ptr, raw_off, sizes, d_compA.ptr, comp_off, dest)
d_compB.upload(d_compA.download())
def comp(k):
    for _ in range(k): A.compress_device(p, d_raw.ptr, raw_off, sizes, d_compA.ptr, comp_off, dest)
def dec(k):
    for _ in range(k): B.decompress_device(d_compB.ptr, comp_off, sizes, bs, d_out.ptr, raw_off)
comp(5); dec(5)
t = time.perf_counter(); comp(K); tc = (time.perf_counter() - t) / K * 1e6
t = time.perf_counter(); dec(K); td = (time.perf_counter() - t) / K * 1e6
t = time.perf_counter()
th = [threading.Thread(target=comp, args=(K,)), threading.Thread(target=dec, args=(K,))]
[x.start() for x in th]; [x.join() for x in th]
tb = (time.perf_counter() - t) / K * 1e6
print("compress alone %.1f us/batch, decompress alone %.1f us/batch, sum %.1f;  both at once (two engines, two threads): %.1f us per pair" % (tc, td, tc + td, tb))
# decode-heavy: how many decompress batches fit while K compress batches run?
done = [0]
def dec_until(ev):
    while not ev.is_set():
        B.decompress_device(d_compB.ptr, comp_off, sizes, bs, d_out.ptr, raw_off); done[0] += 1
ev = threading.Event()
t2 = threading.Thread(target=dec_until, args=(ev,)); t2.start()
t = time.perf_counter(); comp(K); tc2 = (time.perf_counter() - t) / K * 1e6
ev.set(); t2.join()
print("compress with decompress running flat out beside it: %.1f us/batch, %d decompress batches done meanwhile (%.2f per compress batch)" % (tc2, done[0], done[0] / K))
assert np.array_equal(d_out.download(), host)
os._exit(0)
