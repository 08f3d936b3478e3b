import importlib.util, os, sys, sysconfig, time
import numpy as np
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, "compressed-image_amd")]
from cimg import synth
path = os.path.join(ROOT, "compressed-image_amd", "compressed_image" + sysconfig.get_config_var("EXT_SUFFIX"))
spec = importlib.util.spec_from_file_location("compressed_image", path)
ci = importlib.util.module_from_spec(spec); spec.loader.exec_module(ci)
def bench(f, reps=8):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps * 1e3
for h in (512, 1024, 2048, 4096, 8192):
    a = synth.tiled_channel(np.float16, 4096, h)
    ch = ci.Channel(a, 4096, h, ci.Codec.lz4, 9)
    t_c = bench(lambda: ci.Channel(a, 4096, h, ci.Codec.lz4, 9))
    t_d = bench(lambda: ch.get_decompressed())
    t_e = bench(lambda: np.empty((h, 4096), np.float16).fill(1))
    print("4096 x %5d f16 (%4d MiB, %2d chunks): construct %6.2f ms  get_decompressed %6.2f ms   (np.empty+fill %5.2f ms)" % (h, a.nbytes >> 20, ch.num_chunks() if hasattr(ch, "num_chunks") else -1, t_c, t_d, t_e))
