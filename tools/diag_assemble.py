"""Encode launch with in-launch assembly: per work item, when it was taken (stamp 0), when its codec work ended (3), when its wave
finished laying a chunk out -- only the 32 items whose wave closed a chunk -- (1), and when its streams were copied into place (2)."""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, faulthandler; faulthandler.dump_traceback_later(90, exit=True)
from cimg import hip, synth
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
eng = hip.Engine(0)
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_comp = eng.alloc(n), eng.alloc(nchunks * stride)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(3):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
eng.debug_stamps(True)
eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
st = eng.read_stamps(0)
t0 = st[:, 1].min()
T = lambda k: np.where(st[:, 4 * k + 1] > 0, (st[:, 4 * k + 1].astype(np.float64) - t0) / 100.0, np.nan)
start, closed, placed, end = T(0), T(1), T(2), T(3)
wave = st[:, 4].astype(np.int64)
print(fam, "items", len(st), " codec work ends: p50 %.1f p90 %.1f max %.1f us" % tuple(np.nanpercentile(end, [50, 90, 100])))
c = closed[~np.isnan(closed)]
print("  chunks closed inside the launch: %d; layout finished at us: %s" % (c.size, np.sort(c).round(1).tolist()))
cl = np.nonzero(~np.isnan(closed))[0]
print("  layout duration (codec end of the closing item -> chunk published) us: mean %.1f max %.1f" % ((closed[cl] - end[cl]).mean(), (closed[cl] - end[cl]).max()))
print("  streams in place: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us (%d items never stamped)" % (*np.nanpercentile(placed, [10, 50, 90, 99, 100]), int(np.isnan(placed).sum())))
lag = placed - end
print("  codec end -> in place, per item: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.nanpercentile(lag, [10, 50, 90, 100])))
# per wave: when did it run out of codec work, when was its last item in place, how many items
wave = np.unique(wave, return_inverse=True)[1]; nw = wave.max() + 1
last_end = np.full(nw, 0.0); last_placed = np.full(nw, 0.0); cnt = np.zeros(nw, int)
np.maximum.at(last_end, wave, np.nan_to_num(end)); np.maximum.at(last_placed, wave, np.nan_to_num(placed)); np.add.at(cnt, wave, 1)
own = last_placed - last_end
print("  per wave: items mean %.2f; out of codec work at p50 %.1f p90 %.1f max %.1f; last item in place at p50 %.1f p90 %.1f max %.1f" % (cnt.mean(), *np.percentile(last_end, [50, 90, 100]), *np.percentile(last_placed, [50, 90, 100])))
late = last_end > np.percentile(last_end, 85)
print("  the waves that run out of work last (%d): time from their last codec end to their last item in place: mean %.1f max %.1f us; the others: mean %.1f" % (late.sum(), own[late].mean(), own[late].max(), own[~late].mean()))
order = np.argsort(np.nan_to_num(placed))[::-1][:8]
print("  the 8 items in place last: " + ", ".join("item %d (wave %d): codec end %.0f, in place %.0f" % (i, wave[i], end[i], placed[i]) for i in order))
