import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd")]
import numpy as np, torch
from cimg import hip, synth
eng = hip.Engine(0)
CHUNK, BLOCK = 4 << 20, 32768
for fam in ("tiled", "natural"):
    host = np.ascontiguousarray(getattr(synth, fam + "_channel")(np.float32, 4096, 8192)).view(np.uint8).ravel()
    N = host.size; nch = N // CHUNK
    for name, p, kid in (("lz4 split", hip.cparams(4, clevel=9, blocksize=BLOCK), hip.K_ENCODE), ("lz4 never-split", hip.cparams(4, clevel=9, blocksize=BLOCK, splitmode=2), hip.K_ENCODE),
                         ("zstd clevel 9", hip.cparams(4, clevel=9, blocksize=BLOCK, compcode=hip.ZSTD), hip.K_ENCODE_ZSTD), ("zstd clevel 5 (split)", hip.cparams(4, clevel=5, blocksize=BLOCK, compcode=hip.ZSTD), hip.K_ENCODE_ZSTD)):
        raw_off = np.arange(nch, dtype=np.int64) * CHUNK; comp_off = np.arange(nch, dtype=np.int64) * (CHUNK + 32)
        d_raw = torch.from_numpy(host).cuda(); d_comp = torch.empty(nch * (CHUNK + 32), dtype=torch.uint8, device="cuda")
        cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, [CHUNK] * nch, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nch)
        eng.enable_timing(1); eng.reset_timing()
        for _ in range(3): cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, [CHUNK] * nch, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nch)
        ms, k = eng.kernel_time(kid); eng.enable_timing(False)
        print("%s float32 %s: encode %.2f ms per 128 MiB, ratio %.3f" % (fam, name, ms / k, N / float(np.asarray(cb).sum())))
eng.close(); os._exit(0)
