"""The pybind11 module on image sizes people have: 1920x1080 and 3840x2160, four float16 channels -- every chunk ends with a
leftover block.  Construct + get_decompressed, GB/s of pixels; power-of-two sizes beside them."""
import importlib.util, os, sys, sysconfig, time
import numpy as np
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, "compressed-image_amd")]
from cimg import synth
path = os.path.join(ROOT, "compressed-image_amd", "compressed_image" + sysconfig.get_config_var("EXT_SUFFIX"))
spec = importlib.util.spec_from_file_location("compressed_image", path)
ci = importlib.util.module_from_spec(spec); spec.loader.exec_module(ci)
for w, h in ((1920, 1080), (2048, 1024), (3840, 2160), (4096, 2048), (4096, 4096), (5000, 3000)):
    arr = np.stack([synth.tiled_channel(np.float16, w, h, c=c) for c in range(4)])
    names = ["R", "G", "B", "A"]
    for _ in range(3):
        img = ci.Image(np.float16, [arr[c] for c in range(4)], w, h, names, ci.Codec.lz4, 9)
        back = img.get_decompressed()
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps): img = ci.Image(np.float16, [arr[c] for c in range(4)], w, h, names, ci.Codec.lz4, 9)
    t1 = time.perf_counter()
    for _ in range(reps): back = img.get_decompressed()
    t2 = time.perf_counter()
    assert np.array_equal(np.asarray(back).reshape(arr.shape).view(np.uint16), arr.view(np.uint16))
    n = arr.nbytes
    print("%4d x %4d x 4 float16 (%6.1f MiB): construct %6.2f ms = %5.1f GB/s   get_decompressed %6.2f ms = %5.1f GB/s" % (
        w, h, n / 2**20, (t1 - t0) / reps * 1e3, n * reps / (t1 - t0) / 1e9, (t2 - t1) / reps * 1e3, n * reps / (t2 - t1) / 1e9))
os._exit(0)
