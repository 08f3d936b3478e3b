import sys, os
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
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
for _ in range(3):
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
eng.debug_stamps(True)
eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
st = eng.read_stamps(1)
t0 = st[:, 1].min()
T = lambda k: (st[:, 4 * k + 1].astype(np.float64) - t0) / 100.0
start, s1, end = T(0), T(2), T(3)
hw = st[:, 2].astype(np.int64); xcc = st[:, 3].astype(np.int64) & 15
# HW_ID gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wid = hw & 15
first = start < 8.0
dur = end - s1
print("first blocks: chain+unshuffle by simd:", [round(float(dur[first & (simd == k)].mean()), 1) for k in range(4)], "counts", [int((first & (simd == k)).sum()) for k in range(4)])
print("by wave slot:", {int(k): (round(float(dur[first & (wid == k)].mean()), 1), int((first & (wid == k)).sum())) for k in np.unique(wid[first])})
print("by xcc:", [round(float(dur[first & (xcc == k)].mean()), 1) for k in range(8)])
print("by se:", [round(float(dur[first & (se == k)].mean()), 1) for k in np.unique(se)])
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
import collections
cnt = collections.Counter(key[first].tolist())
per_simd = np.array([cnt[k] for k in key[first]])
for c in sorted(set(per_simd.tolist())):
    print("waves sharing a SIMD =", c, ": n", int((per_simd == c).sum()), "mean dur %.1f" % dur[first][per_simd == c].mean())
cukey = key // 10
cntcu = collections.Counter(cukey[first].tolist())
print("waves per CU histogram:", collections.Counter(cntcu.values()))
# slowest 10%: what do they share
slow = first & (dur > np.percentile(dur[first], 90))
print("slow waves: by simd", [int((slow & (simd == k)).sum()) for k in range(4)], "by xcc", [int((slow & (xcc == k)).sum()) for k in range(8)])
later = ~first
print("later blocks by wave slot:", {int(k): (round(float(dur[later & (wid == k)].mean()), 1), int((later & (wid == k)).sum())) for k in np.unique(wid[later])})
print("end of the wave's last block by slot: ", {int(k): [round(float(x), 1) for x in np.percentile(end[later & (wid == k)], [0, 50, 90, 100])] for k in np.unique(wid[later])}, " span %.1f" % end.max())
print("later blocks dur by simd:", [round(float(dur[later & (simd == k)].mean()), 1) for k in range(4)])
