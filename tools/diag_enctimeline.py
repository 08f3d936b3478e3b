# When does every work item of the encode launch start and end, and on which persistent workgroup?  (production build,
# in-kernel stamps; diagnostics only)
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
p = hip.cparams(2, compcode=int(os.environ.get("CIMG_DIAG_CODEC", "1")))
for _ in range(3):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
eng.debug_stamps(True)
eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
st = eng.read_stamps(0)
items = len(st)
t0 = st[:, 1].min()
start, end = (st[:, 1] - t0) / 100.0, (st[:, 13] - t0) / 100.0
wg = st[:, 4].astype(np.int64)
dur = end - start
half = items // 2
print(fam, "items", items, "span us %.1f" % end.max(), "workgroups", len(np.unique(wg)))
for name, sl in (("first half of the queue (high-byte planes)", slice(0, half)), ("second half (low-byte planes)", slice(half, None))):
    d = dur[sl]
    print("  %-44s duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f;  starts %.1f..%.1f  ends ..%.1f" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max(), start[sl].min(), start[sl].max(), end[sl].max()))
# per workgroup: items taken, busy time, last end
wg = np.unique(wg, return_inverse=True)[1]
last = np.zeros(wg.max() + 1); busy = np.zeros(wg.max() + 1); cnt = np.zeros(wg.max() + 1, dtype=int); hi = np.zeros(wg.max() + 1, dtype=int)
np.maximum.at(last, wg, end); np.add.at(busy, wg, dur); np.add.at(cnt, wg, 1); np.add.at(hi, wg[:half], 1)
print("  per workgroup: items mean %.2f; high-byte planes taken histogram %s" % (cnt.mean(), np.bincount(hi).tolist()))
print("  last end per workgroup us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f;  busy fraction of span mean %.3f" % (*np.percentile(last, [0, 10, 50, 90, 100]), (busy / end.max()).mean()))
print("  sum of item time / (workgroups x span) = %.3f   ideal span (perfect balance) %.1f us" % (dur.sum() / (len(last) * end.max()), dur.sum() / len(last)))
# does sharing a SIMD matter?  HW_ID simd [5:4], cu [11:8], sh [12], se [15:13]
hw = st[:, 2].astype(np.int64); xcc = st[:, 3].astype(np.int64) & 15
simd_key = (xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3)
wg_simd = np.zeros(wg.max() + 1, dtype=np.int64); wg_simd[wg] = simd_key
_, inv, per_simd = np.unique(wg_simd, return_inverse=True, return_counts=True)
share = per_simd[inv]                       # waves on the SIMD of each workgroup
for k in np.unique(share):
    m = share[wg[:half]] == k
    print("  high-byte plane duration on a SIMD holding %d encode wave(s): mean %.1f us (%d items)" % (k, dur[:half][m].mean(), m.sum()))
# the tail: which items end last, and how long did they take?
order = np.argsort(end)[::-1][:12]
print("  the 12 items that end last: " + ", ".join("%s %d: %.0f us long, ends %.0f" % ("hi" if i < half else "lo", i, dur[i], end[i]) for i in order))
lo = dur[half:]
print("  low-byte plane items longer than 20 us: %d of %d (mean of those %.1f us)" % ((lo > 20).sum(), lo.size, lo[lo > 20].mean() if (lo > 20).any() else 0))
# the long low-byte items: when do they start, on what kind of workgroup, and which stamps are far apart?
ls, ld = start[half:], dur[half:]
for a, b in ((0, 250), (250, 270), (270, 280), (280, 290), (290, 400)):
    m = (ls >= a) & (ls < b)
    if m.any(): print("  low-byte items starting in [%d, %d) us: %d, duration mean %.1f p50 %.1f p90 %.1f max %.1f" % (a, b, m.sum(), ld[m].mean(), *np.percentile(ld[m], [50, 90]), ld[m].max()))
long_i = half + np.nonzero(ld > 25)[0]
if long_i.size:
    staged = (st[long_i, 5] - st[long_i, 1]) / 100.0 if st.shape[1] > 5 else None
    print("  long low-byte items: on workgroups that took %s high-byte planes; blocks %s ..." % (np.bincount(hi[wg[long_i]]).tolist(), (long_i[:16] - half).tolist()))
    for i in long_i[:6]:
        print("    item %d stamps (us from item start): %s" % (i, [round((int(x) - int(st[i, 1])) / 100.0, 1) if x else None for x in st[i, :16]]))
os._exit(0)
