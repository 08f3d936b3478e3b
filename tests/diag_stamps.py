import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(40, exit=True)
from cimg import hip, synth
eng = hip.Engine(0)
chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(5):
    cb = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
cb = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
enc = np.zeros((0, 8), np.uint64)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
dec = eng.read_stamps(1)
np.savez_compressed("gpurun_out/stamps.npz", enc=enc, dec=dec)
for name, st in (("decode", dec),):
    st = st[st[:, 1] > 0]
    t0 = st[:, 1].min()
    start = (st[:, 1] - t0) / 100.0      # us
    end = (st[:, 5] - t0) / 100.0
    dur = end - start
    cyc = (st[:, 4] - st[:, 0]).astype(np.float64)
    clk = cyc / np.maximum(dur, 1e-3) / 1e3  # GHz
    print(name, "wgs", len(st), "span us", end.max(), "dur us: mean %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()),
          "clock GHz %.2f" % np.median(clk))
    # concurrency: average number of WGs alive
    print("   avg concurrent WGs %.1f (per CU %.2f)" % (dur.sum() / end.max(), dur.sum() / end.max() / 256))
    if name == "encode":
        w = np.arange(len(enc))[enc[:, 1] > 0]
        plane = (w % 16) // 8
        for pl in (0, 1):
            d = dur[plane == pl]
            print("   plane", pl, "dur us mean %.1f p50 %.1f max %.1f" % (d.mean(), np.median(d), d.max()))
    hw = st[:, 2].astype(np.uint32)
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    xcc = st[:, 3].astype(np.uint32) & 0xF
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    print("   distinct CUs", len(np.unique(key)), "wgs/CU min/max", np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max())
