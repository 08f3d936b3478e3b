"""Development benchmark of the zstd WRITE path (cimg_encode_streams_zstd): 4096 x 8192 float32 of the tiled / natural family
(128 MiB, 4 MiB chunks, 32 KiB blocks, clevel 9 = one frame per block), device-resident; kernel time, ratio, and -- unless
CIMG_DIAG_NO_CHECK is set (ablation builds write frames nobody can read) -- the decode check.
usage: python tools/diag_zstd_enc.py [tiled|natural ...] [--mib M]"""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd")]
import numpy as np
import torch
from cimg import hip, synth
fams = [a for a in sys.argv[1:] if a in ("tiled", "natural")] or ["tiled", "natural"]
mib = int(sys.argv[sys.argv.index("--mib") + 1]) if "--mib" in sys.argv else 128
eng = hip.Engine(0)
CHUNK, BLOCK = 4 << 20, 32768
for fam in fams:
    W, H = 4096, mib * (1 << 20) // (4096 * 4)
    host = np.ascontiguousarray(getattr(synth, fam + "_channel")(np.float32, W, H)).view(np.uint8).ravel()
    N = host.size; nch = N // CHUNK
    p = hip.cparams(4, clevel=9, blocksize=BLOCK, compcode=hip.ZSTD)
    nbytes = [CHUNK] * nch
    raw_off = np.arange(nch, dtype=np.int64) * CHUNK
    comp_off = np.arange(nch, dtype=np.int64) * (CHUNK + 32)
    d_raw = torch.from_numpy(host).cuda()
    d_comp = torch.empty(nch * (CHUNK + 32), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
    cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nch)
    eng.enable_timing(1); eng.reset_timing()
    for _ in range(3):
        cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nch)
    ms, k = eng.kernel_time(hip.K_ENCODE_ZSTD)
    parts = []
    for kid in range(len(hip.KERNELS)):
        if kid in (hip.K_ENCODE_ZSTD,): continue
        pm, pk = eng.kernel_time(kid)
        if pk: parts.append("%s %.2f ms" % (hip.KERNELS[kid], pm / pk))
    eng.enable_timing(False)
    ok = "not checked"
    if not os.environ.get("CIMG_DIAG_NO_CHECK"):
        eng.decompress_device(d_comp.data_ptr(), comp_off, nbytes, [BLOCK] * nch, d_out.data_ptr(), raw_off)
        ok = "bit-exact" if torch.equal(d_out, d_raw) else "DIFFER"
    print("%s float32 %d MiB, zstd clevel 9: cimg_encode_streams_zstd %.2f ms = %.1f GB/s, ratio %.3f, round trip %s  [%s]" % (
        fam, mib, ms / k, N / (ms / k * 1e-3) / 1e9, N / float(np.asarray(cb).sum()), ok, ", ".join(parts)))
eng.close()
os._exit(0)
