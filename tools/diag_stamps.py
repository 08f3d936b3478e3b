"""Lean decode launch (persistent, one wave per LDS plane): in-kernel stamps per BLOCK -- 0: block begins, 1: walk finished,
2: coded bytes in LDS + stored plane requested (+ next blocks' loads issued: stamp 2 sits in front of them), 3: un-shuffle done.
Prints the phases of the FIRST block of a wave (everything exposed) and of the later ones (walk and staging pipelined)."""
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
idx = np.nonzero(st[:, 1] > 0)[0]
st = st[idx]
t0 = st[:, 1].min()
T = lambda k: (st[:, 4 * k + 1] - t0) / 100.0
start, s0, s1, end = T(0), T(1), T(2), T(3)
print(fam, "blocks stamped", len(st), "span us %.1f" % end.max(), " shader clock MHz %.0f" % np.median((st[:, 12] - st[:, 0]) / np.maximum(st[:, 13] - st[:, 1], 1) * 100))
G = int((start < np.percentile(start, 40)).sum()) if len(st) else 0
first = start < 8.0
print("   blocks that begin in the first 8 us (= waves of the launch):", int(first.sum()))
for label, sel in (("first block of a wave", first), ("later blocks", ~first)):
    if not sel.any():
        continue
    print("  ", label, "(%d)" % sel.sum())
    print("      begin percentiles us:", np.percentile(start[sel], [0, 25, 50, 75, 100]).round(1), " end:", np.percentile(end[sel], [0, 25, 50, 75, 100]).round(1))
    for name, d in (("total", end - start), ("walk (begin -> walk done)", s0 - start), ("staging (-> stored plane requested)", s1 - s0), ("next loads + chain + un-shuffle", end - s1)):
        d = d[sel]
        print("      %-38s mean %.1f p50 %.1f p90 %.1f max %.1f us" % (name, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
# gap between the end of a wave's block and the begin of its next one (block b + G)
hw = st[:, 2].astype(np.int64); xcc = st[:, 3].astype(np.int64) & 15
print("   by xcc:", np.bincount(xcc, minlength=8).tolist())
