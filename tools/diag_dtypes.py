"""configs[1]'s geometry (128 MiB, 4 MiB chunks, 32 KiB blocks, byte shuffle, clevel 9) on other pixel types: kernel time of encode and
decode per 128 MiB, device-resident, round trip checked.  usage: python tools/diag_dtypes.py [lz4|blosclz] [dtype ...]"""
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd")]
import numpy as np, torch
from cimg import hip, synth
codec = "blosclz" if "blosclz" in sys.argv[1:] else "lz4"
dts = [a for a in sys.argv[1:] if a not in ("lz4", "blosclz")] or ["uint8", "uint16", "float16", "float32"]
eng = hip.Engine(0)
CHUNK, BLOCK = 4 << 20, 32768
for dt in dts:
    dtype = np.dtype(dt)
    for fam in ("tiled", "natural"):
        host = np.ascontiguousarray(getattr(synth, fam + "_channel")(dtype.type, 4096, (128 << 20) // (4096 * dtype.itemsize))).view(np.uint8).ravel()
        N = host.size; nch = N // CHUNK
        p = hip.cparams(dtype.itemsize, clevel=9, blocksize=BLOCK, compcode=hip.BLOSCLZ if codec == "blosclz" else hip.LZ4)
        raw_off = np.arange(nch, dtype=np.int64) * CHUNK; comp_off = np.arange(nch, dtype=np.int64) * (CHUNK + 32)
        d_raw = torch.from_numpy(host).cuda(); d_comp = torch.empty(nch * (CHUNK + 32), dtype=torch.uint8, device="cuda"); d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
        def step():
            cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, [CHUNK] * nch, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nch)
            eng.decompress_device(d_comp.data_ptr(), comp_off, [CHUNK] * nch, [BLOCK] * nch, d_out.data_ptr(), raw_off, comp_size=cb)
            return cb
        cb = step()
        ok = torch.equal(d_out, d_raw)
        eng.enable_timing(1); eng.reset_timing()
        for _ in range(4): step()
        ems, ek = eng.kernel_time(hip.K_ENCODE); dms, dk = eng.kernel_time(hip.K_DECODE); eng.enable_timing(False)
        st = eng.decode_stats()
        print("%-8s %-8s %s: encode %8.1f us, decode %7.1f us per 128 MiB = %6.1f GB/s round trip, ratio %.3f, %s" % (
            dt, fam, codec, ems / max(ek, 1) * 1e3, dms / max(dk, 1) * 1e3, 2 * N / ((ems / max(ek, 1) + dms / max(dk, 1)) * 1e-3) / 1e9, N / float(np.asarray(cb).sum()), "bit-exact" if ok else "DIFFER"), flush=True)
eng.close(); os._exit(0)
