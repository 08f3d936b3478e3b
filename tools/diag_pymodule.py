"""Throughput of the pybind11 module (the reference's Python surface) on the GPU box: Image from numpy, back to numpy."""
import importlib.util, os, sys, sysconfig, time
import numpy as np
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, "compressed-image_amd")]
from cimg import synth
path = os.path.join(ROOT, "compressed-image_amd", "compressed_image" + sysconfig.get_config_var("EXT_SUFFIX"))
spec = importlib.util.spec_from_file_location("compressed_image", path)
ci = importlib.util.module_from_spec(spec); spec.loader.exec_module(ci)
arr = np.stack([synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)])
n = arr.nbytes
for _ in range(2):
    img = ci.Image(np.float16, [arr[c] for c in range(4)], 4096, 4096, ["R", "G", "B", "A"], ci.Codec.lz4, 9)
    back = img.get_decompressed()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    img = ci.Image(np.float16, [arr[c] for c in range(4)], 4096, 4096, ["R", "G", "B", "A"], ci.Codec.lz4, 9)
t1 = time.perf_counter()
for _ in range(reps):
    back = img.get_decompressed()
t2 = time.perf_counter()
assert np.array_equal(np.asarray(back).reshape(arr.shape).view(np.uint16), arr.view(np.uint16))
print("compressed_image.Image (4 x 4096^2 f16): construct %.2f GB/s (%.1f ms)  get_decompressed %.2f GB/s (%.1f ms)  ratio %.3f" % (
    n * reps / (t1 - t0) / 1e9, (t1 - t0) / reps * 1e3, n * reps / (t2 - t1) / 1e9, (t2 - t1) / reps * 1e3, img.compression_ratio() if hasattr(img, "compression_ratio") else 0))
ch = ci.Channel(arr[0], 4096, 4096, ci.Codec.lz4, 9)
for _ in range(3):                       # (the first results of a new size allocate their page-locked buffers: 5 ms each)
    b = ch.get_decompressed()
t0 = time.perf_counter()
for _ in range(reps):
    ch = ci.Channel(arr[0], 4096, 4096, ci.Codec.lz4, 9)
t1 = time.perf_counter()
for _ in range(reps):
    b = ch.get_decompressed()
t2 = time.perf_counter()
print("compressed_image.Channel (4096^2 f16): construct %.2f GB/s  get_decompressed %.2f GB/s" % (arr[0].nbytes * reps / (t1 - t0) / 1e9, arr[0].nbytes * reps / (t2 - t1) / 1e9))
