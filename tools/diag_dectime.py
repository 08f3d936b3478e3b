# decode kernel time of whatever library CIMG_LIB names (ablation builds: timing only).  Diagnostics only.
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
fam = sys.argv[1] if len(sys.argv) > 1 else "tiled"
eng = hip.Engine(0)
chans = [getattr(synth, fam + "_channel")(np.float16, 4096, 4096, c=c) for c in range(4)]
host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
n, chunk = host.size, 4 * 1024 * 1024
nchunks, stride = n // chunk, chunk + 64
d_raw, d_comp, d_out = eng.alloc(n), eng.alloc(nchunks * stride), eng.alloc(n)
d_raw.upload(host)
raw_off = np.arange(nchunks) * chunk; comp_off = np.arange(nchunks) * stride
p = hip.cparams(2)
eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
for _ in range(3): eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off, check=False)
eng.enable_timing(True); eng.reset_timing()
for _ in range(10): eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off, check=False)
ms, k = eng.kernel_time(3)
print(os.path.basename(os.environ.get("CIMG_LIB", "default")), fam, "decode us %.1f" % (ms / k * 1e3))
os._exit(0)
