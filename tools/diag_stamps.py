import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(60, exit=True)
from cimg import hip, synth
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
eng = hip.Engine(0)
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(3):
    cb = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
st = eng.read_stamps(1)
st = st[st[:, 1] > 0]
t0 = st[:, 1].min()
T = lambda k: (st[:, 4 * k + 1] - t0) / 100.0
start, s0, s1, end = T(0), T(1), T(2), T(3)
print(fam, "decode wgs", len(st), "span us %.1f" % end.max(), " shader clock MHz %.0f" % np.median((st[:, 12] - st[:, 0]) / np.maximum(st[:, 13] - st[:, 1], 1) * 100))
print("   start-time percentiles us:", np.percentile(start, [0, 25, 50, 75, 100]).round(1), " end:", np.percentile(end, [0, 25, 50, 75, 100]).round(1))
lz_end = None
for name, d in (("total", end - start), ("header walk", s0 - start), ("stage coded bytes", s1 - s0), ("LZ4 + unshuffle", end - s1)):
    print("   %-34s mean %.1f p50 %.1f p90 %.1f max %.1f us" % (name, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
print("   avg concurrent WGs per CU %.2f" % ((end - start).sum() / end.max() / 256))
# where did the workgroups run?  HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
hw = st[:, 2].astype(np.int64); xcc = st[:, 3].astype(np.int64) & 15
cu_key = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
simd = (hw >> 4) & 3
early = start < 2.0
keys, cnt = np.unique(cu_key[early], return_counts=True)
print("   CUs seen %d (early %d); early workgroups per CU: min %d median %d max %d; histogram %s" % (len(np.unique(cu_key)), len(keys), cnt.min(), np.median(cnt), cnt.max(), np.bincount(cnt).tolist()))
print("   early by xcc:", np.bincount(xcc[early], minlength=8).tolist(), " all by xcc:", np.bincount(xcc, minlength=8).tolist(), " early wave-0 by simd:", np.bincount(simd[early], minlength=4).tolist())
hist, edges = np.histogram(start, bins=[0, 1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128])
print("   start histogram (us bins", edges.tolist(), "):", hist.tolist())
