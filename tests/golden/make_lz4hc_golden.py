#!/usr/bin/env python3
"""Generate tests/golden/lz4hc_kat.npz: blosc2 chunks whose streams were coded by LZ4_compress_HC of the system
liblz4 (1.9.3), framed the way c-blosc2 frames an lz4hc chunk (codec format 1, compcode 2, DONT_SPLIT: lz4hc never
splits; level = 2 * clevel - 1).

The path only has to DECODE such chunks (enums::codec::lz4hc, compressed_image/include/compressed/enums.h:18-24):
the payloads are ordinary LZ4 blocks.  No LZ4HC encoder is restated anywhere in this repository.

Run:  python tests/golden/make_lz4hc_golden.py      (needs liblz4.so.1; output committed)
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "compressed-image_amd"))
from cimg import synth  # noqa: E402


def shuffle(ts, blk):
    ne = blk.size // ts
    out = blk.copy()
    out[:ne * ts] = blk[:ne * ts].reshape(ne, ts).T.ravel()
    return out


def frame(lz4, src, ts, blocksize, clevel):
    nbytes = src.size
    nblocks = -(-nbytes // blocksize)
    body = b""
    bstarts = []
    base = 32 + 4 * nblocks
    for j in range(nblocks):
        blk = src[j * blocksize:(j + 1) * blocksize]
        f = shuffle(ts, blk)
        bstarts.append(base + len(body))
        if (f == f[0]).all():
            v = int(f[0])
            body += struct.pack("<i", -v) + (b"\x01" if v else b"")
            continue
        out = np.zeros(blk.size + 64, np.uint8)
        r = lz4.LZ4_compress_HC(f.ctypes.data, out.ctypes.data, blk.size, blk.size, 2 * clevel - 1)
        if r <= 0 or r == blk.size:
            body += struct.pack("<i", blk.size) + f.tobytes()
        else:
            body += struct.pack("<i", r) + out[:r].tobytes()
    cbytes = base + len(body)
    flags = 0x01 | 0x04 | 0x10 | (1 << 5)
    hdr = struct.pack("<BBBBiii", 5, 1, flags, ts, nbytes, blocksize, cbytes) + bytes([0, 0, 0, 0, 0, 1]) + bytes([2, 0]) + bytes(8)
    assert len(hdr) == 32
    return hdr + struct.pack(f"<{nblocks}i", *bstarts) + body


def main():
    lz4 = C.CDLL("liblz4.so.1")
    lz4.LZ4_versionString.restype = C.c_char_p
    lz4.LZ4_compress_HC.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rng = np.random.Generator(np.random.PCG64(5))
    cases = {
        "tiled_u16": (synth.tiled_channel(np.uint16, 1024, 36), 2, 32768, 9),
        "natural_f32": (synth.natural_channel(np.float32, 512, 20), 4, 32768, 5),
        "u8_small_blocks": (synth.tiled_channel(np.uint8, 256, 33), 1, 1024, 9),
        "mixed_u16": (np.concatenate([np.zeros(16384, np.uint16), rng.integers(0, 65536, 4096, dtype=np.uint16),
                                      np.full(16384, 0x0707, np.uint16), synth.natural_channel(np.uint16, 128, 101).ravel()]), 2, 32768, 9),
    }
    store = {"lz4_version": np.array(lz4.LZ4_versionString().decode()), "cases": np.array(list(cases))}
    for name, (arr, ts, bs, clevel) in cases.items():
        src = np.ascontiguousarray(arr).view(np.uint8).ravel()
        chunk = frame(lz4, src, ts, bs, clevel)
        store["in|" + name] = src
        store["chunk|" + name] = np.frombuffer(chunk, np.uint8)
        print(name, src.size, "->", len(chunk))
    path = os.path.join(HERE, "lz4hc_kat.npz")
    np.savez_compressed(path, **store)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
