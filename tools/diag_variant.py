import sys, os, faulthandler
faulthandler.dump_traceback_later(12, exit=True)
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np
from cimg import hip, synth
import _oracle as O
eng = hip.Engine(0)
a = synth.tiled_channel(np.float16, 1024, 64)
raw = a.view(np.uint8).ravel()
chunks = eng.compress_host(hip.cparams(2), raw, [raw.size], [raw.size + 32])
r, want = O.compress(O.cparams(2), raw, destsize=raw.size + 32)
print(os.environ.get("CIMG_LIB", "product").split("/")[-1], "ok" if chunks[0] == want else "WRONG BYTES", len(chunks[0]), r, flush=True)
os._exit(0)
