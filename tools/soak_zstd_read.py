"""Soak of the zstd read path on the GPU: chunks of random geometry (element size 1 .. 16, block size 1 KiB .. 144 KiB, a ragged
last block, clevel 1 .. 9 = split and unsplit, every filter, four data families) made with the box's libzstd the way c-blosc2
frames them (tests/golden/make_zstd_golden.py), decoded in batches of mixed geometry, compared with their pixels.
usage: python tools/soak_zstd_read.py [cases] [seed]"""
import sys, os, ctypes as C, ctypes.util
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "tests", "golden")]
import numpy as np
from cimg import hip, synth
import make_zstd_golden as G
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
name = ctypes.util.find_library("zstd")
if not name:
    print("no libzstd on this box"); sys.exit(0)
z = C.CDLL(name)
z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
z.ZSTD_compress.restype = C.c_size_t; z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
z.ZSTD_isError.argtypes = [C.c_size_t]
rng = np.random.default_rng(seed)
eng = hip.Engine(0)
batch, want, desc, done, bad = [], [], [], 0, 0
refused = []
def flush():
    global batch, want, desc, done, bad
    if not batch: return
    try:
        outs, status = eng.decompress_host(batch)
    except hip.CodecError as ex:
        # (the call as a whole was refused -- a geometry the planner does not take -- or a chunk failed: one chunk per call then says which)
        if len(batch) > 1:
            b2, w2, d2 = batch, want, desc
            for c, w, d in zip(b2, w2, d2):
                batch, want, desc = [c], [w], [d]
                flush()
            return
        refused.append((desc[0], str(ex)))
        done += 1
        batch, want, desc = [], [], []
        return
    for o, w, d, st in zip(outs, want, desc, status):
        ok = st == 0 and o.tobytes() == w.tobytes()
        if not ok:
            bad += 1
            print("MISMATCH", d, "status", st, flush=True)
    done += len(batch)
    batch, want, desc = [], [], []
for k in range(cases):
    ts = int(rng.choice([1, 2, 2, 3, 4, 4, 4, 8, 12, 16]))
    bs = int(rng.choice([1024, 4096, 16384, 32768, 32768, 32768, 65536, 131072, 147456]))
    bs -= bs % ts
    nblocks = int(rng.integers(1, 9))
    nbytes = (nblocks - 1) * bs + int(rng.integers(1, bs // ts + 1)) * ts
    if nbytes < bs: bs = nbytes                                   # (c-blosc2 never writes a block size above the chunk's bytes)
    fam = int(rng.integers(0, 4))
    if fam == 0:   raw = np.ascontiguousarray(synth.tiled_channel(np.float32, 1024, -(-nbytes // 4096) + 1)).view(np.uint8).ravel()[:nbytes]
    elif fam == 1: raw = np.ascontiguousarray(synth.natural_channel(np.uint16, 1024, -(-nbytes // 2048) + 1)).view(np.uint8).ravel()[:nbytes]
    elif fam == 2: raw = (np.arange(nbytes, dtype=np.int64) // max(1, int(rng.integers(1, 400))) % 251).astype(np.uint8)
    else:          raw = np.where(rng.random(nbytes) < 0.9, 7, rng.integers(0, 256, nbytes)).astype(np.uint8)
    raw = np.ascontiguousarray(raw)
    clevel = int(rng.choice([1, 3, 5, 6, 8, 9]))
    filt = str(rng.choice(["shuffle", "shuffle", "bitshuffle", "none"]))
    batch.append(G.frame(z, raw, ts, bs, clevel, filt)); want.append(raw); desc.append((ts, bs, nbytes, fam, clevel, filt))
    if len(batch) == int(os.environ.get("SOAK_BATCH", "12")): flush()
flush()
st = eng.zstd_stats()
for d, msg in refused: print("REFUSED", d, msg)
print("zstd read path soak: %d chunks of random geometry, %d mismatches, %d calls refused; zstd batches %d, plans refused %d" % (done, bad, len(refused), st["zstd_batches"], st["blocks_refused"]))
eng.close()
os._exit(1 if bad else 0)
