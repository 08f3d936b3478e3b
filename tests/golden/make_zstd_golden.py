#!/usr/bin/env python3
"""Generate tests/golden/zstd_kat.npz: (a) single zstd frames made by the system libzstd (ZSTD_compress) from a spread of
inputs and levels, (b) blosc2 chunks whose streams are such frames, framed the way c-blosc2 frames a zstd chunk (codec
format 4, compcode 5; streams split per byte plane for clevel <= 5, one stream per block above; level = 2 * clevel - 1,
clevel 9 -> ZSTD_maxCLevel()).

The path only has to DECODE such chunks (enums::codec::zstd, compressed_image/include/compressed/enums.h:18-24;
csrc/zstd_decode.h): the expected answer of every vector is the input itself.  No zstd encoder exists in this repository.

Run:  python tests/golden/make_zstd_golden.py      (needs libzstd.so.1; output committed)
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "compressed-image_amd"))
from cimg import synth  # noqa: E402


def zcompress(z, data, level):
    src = np.ascontiguousarray(data).view(np.uint8).ravel()
    cap = z.ZSTD_compressBound(src.size)
    out = np.zeros(cap, np.uint8)
    r = z.ZSTD_compress(out.ctypes.data, cap, src.ctypes.data, src.size, level)
    assert not z.ZSTD_isError(r), (src.size, level)
    return out[:r].copy()


def shuffle(ts, blk):
    ne = blk.size // ts
    out = blk.copy()
    out[:ne * ts] = blk[:ne * ts].reshape(ne, ts).T.ravel()
    return out


def bitshuffle(ts, blk):
    """blosc2's bit shuffle of one block, through the oracle's restatement (tests/_oracle.py; only the on-the-fly tests use it)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import _oracle as O
    return np.asarray(O.bitshuffle(ts, blk), np.uint8).copy()


def frame(z, src, ts, blocksize, clevel, filt="shuffle"):
    """filt: "shuffle" (what the reference sets), "bitshuffle" (one stream per block, as c-blosc2 does for bit rows) or "none"."""
    nbytes = src.size
    nblocks = -(-nbytes // blocksize)
    level = 2 * clevel - 1 if clevel < 9 else z.ZSTD_maxCLevel()
    if clevel == 8:
        level = z.ZSTD_maxCLevel() - 2
    body = b""
    bstarts = []
    base = 32 + 4 * nblocks
    split_chunk = clevel <= 5 and ts > 1 and filt == "shuffle"
    for j in range(nblocks):
        blk = src[j * blocksize:(j + 1) * blocksize]
        f = shuffle(ts, blk) if filt == "shuffle" else (bitshuffle(ts, blk) if filt == "bitshuffle" else blk)
        bstarts.append(base + len(body))
        leftover = blk.size != blocksize
        ns = ts if (split_chunk and not leftover and blk.size % ts == 0) else 1
        ne = blk.size // ns
        for s in range(ns):
            st = f[s * ne:(s + 1) * ne]
            if (st == st[0]).all():
                v = int(st[0])
                body += struct.pack("<i", -v) + (b"\x01" if v else b"")
                continue
            out = zcompress(z, st, level)
            if out.size >= st.size:
                body += struct.pack("<i", st.size) + st.tobytes()
            else:
                body += struct.pack("<i", out.size) + out.tobytes()
    cbytes = base + len(body)
    flags = 0x01 | 0x04 | (0 if split_chunk else 0x10) | (4 << 5)
    hdr = struct.pack("<BBBBiii", 5, 1, flags, ts, nbytes, blocksize, cbytes) + bytes([0, 0, 0, 0, 0, {"shuffle": 1, "bitshuffle": 2, "none": 0}[filt]]) + bytes([5, 0]) + bytes(8)
    assert len(hdr) == 32
    return hdr + struct.pack(f"<{nblocks}i", *bstarts) + body


def main():
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_versionString.restype = C.c_char_p
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    rng = np.random.Generator(np.random.PCG64(11))
    text = np.frombuffer((b"the quick brown fox jumps over the lazy dog, " * 40 + b"pack my box with five dozen liquor jugs; " * 30) * 3, np.uint8)
    inputs = {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([7], np.uint8),
        "zeros_4k": np.zeros(4096, np.uint8),
        "rle_run": np.full(20000, 0x5A, np.uint8),
        "random_2k": rng.integers(0, 256, 2048, dtype=np.uint8),
        "random_9k": rng.integers(0, 256, 9000, dtype=np.uint8),
        "nibbles_16k": rng.integers(0, 16, 16384, dtype=np.uint8),
        "skewed_16k": np.minimum(rng.geometric(0.3, 16384), 255).astype(np.uint8),
        "text": text,
        "ramp": (np.arange(30000) // 7 % 251).astype(np.uint8),
        "tiled_hi_plane": np.ascontiguousarray(synth.tiled_channel(np.float16, 1024, 16).view(np.uint8).reshape(-1, 2)[:, 1]),
        "tiled_lo_plane": np.ascontiguousarray(synth.tiled_channel(np.float16, 1024, 16).view(np.uint8).reshape(-1, 2)[:, 0]),
        "natural_u16_block": np.ascontiguousarray(synth.natural_channel(np.uint16, 512, 32)).view(np.uint8).ravel(),
        "natural_plane": np.ascontiguousarray(synth.natural_channel(np.uint16, 512, 32).view(np.uint8).reshape(-1, 2)[:, 1]),
        "repeats_far": np.concatenate([rng.integers(0, 256, 3000, dtype=np.uint8)] * 6),
        "sparse": (rng.random(32768) < 0.02).astype(np.uint8) * rng.integers(1, 255, 32768, dtype=np.uint8),
        "two_blocks_150k": np.concatenate([np.tile(rng.integers(0, 64, 997, dtype=np.uint8), 142), rng.integers(0, 256, 9000, dtype=np.uint8)]),
    }
    levels = [-5, 1, 3, 5, 9, 15, 19, 22]
    store = {"zstd_version": np.array(z.ZSTD_versionString().decode()), "levels": np.array(levels)}
    names = []
    total = 0
    for name, arr in inputs.items():
        store["in|" + name] = arr
        for lv in levels:
            if arr.size > 100000 and lv not in (1, 19):
                continue
            fr = zcompress(z, arr, lv)
            key = f"{name}|L{lv}"
            store["frame|" + key] = fr
            names.append(key)
            total += fr.size
    store["frames"] = np.array(names)
    chunks = {
        "tiled_u16_split": (synth.tiled_channel(np.uint16, 1024, 36), 2, 32768, 3),
        "tiled_f16_unsplit": (synth.tiled_channel(np.float16, 1024, 40), 2, 32768, 9),
        "natural_f32_split": (synth.natural_channel(np.float32, 512, 20), 4, 32768, 5),
        "u8_small_blocks": (synth.tiled_channel(np.uint8, 256, 33), 1, 1024, 7),
        "mixed_u16": (np.concatenate([np.zeros(16384, np.uint16), rng.integers(0, 65536, 4096, dtype=np.uint16),
                                      np.full(16384, 0x0707, np.uint16), synth.natural_channel(np.uint16, 128, 101).ravel()]), 2, 32768, 1),
    }
    store["chunks"] = np.array(list(chunks))
    for name, (arr, ts, bs, clevel) in chunks.items():
        src = np.ascontiguousarray(arr).view(np.uint8).ravel()
        chunk = frame(z, src, ts, bs, clevel)
        store["cin|" + name] = src
        store["chunk|" + name] = np.frombuffer(chunk, np.uint8)
        print(name, src.size, "->", len(chunk))
    path = os.path.join(HERE, "zstd_kat.npz")
    np.savez_compressed(path, **store)
    print(len(names), "frames,", total, "bytes;", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
