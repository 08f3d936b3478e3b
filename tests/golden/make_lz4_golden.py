#!/usr/bin/env python3
"""Generate tests/golden/lz4_kat.npz from the SYSTEM liblz4 (1.9.3, /usr/lib/x86_64-linux-gnu).

liblz4 is a system library, not reference source: c-blosc2 (the reference's codec, absent from
/root/reference) calls LZ4_compress_fast(stream, dst, n, maxout, 10 - clevel) per split stream
(SURVEY.md section 8a N4).  These known-answer vectors pin the oracle's LZ4 block layer (and through
it the HIP encoder) to a real LZ4 implementation.  Caveat recorded in DESIGN.md: c-blosc2 2.17
vendors lz4 1.10.0, not 1.9.3.

Run:  python tests/golden/make_lz4_golden.py      (needs liblz4.so.1; output is committed)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "compressed-image_amd"))
from cimg import synth  # noqa: E402


def streams_of(arr, ts, block=32768):
    """byte-shuffled split streams of the first block of `arr` (what c-blosc2 hands to LZ4)."""
    raw = arr.view(np.uint8).ravel()[:block]
    return [np.ascontiguousarray(s) for s in raw.reshape(-1, ts).T]


def inputs():
    rng = np.random.Generator(np.random.PCG64(20240611))
    out = []
    sizes_small = [1, 2, 4, 5, 11, 12, 13, 14, 15, 16, 17, 31, 64, 65, 255, 256, 270, 1000]
    for n in sizes_small:
        out.append((f"zeros{n}", np.zeros(n, np.uint8)))
        out.append((f"ramp{n}", (np.arange(n) & 255).astype(np.uint8)))
        out.append((f"rand{n}", rng.integers(0, 256, n, dtype=np.uint8)))
        out.append((f"low{n}", rng.integers(0, 3, n, dtype=np.uint8)))
    for n in [4096, 16384, 32768]:
        out.append((f"zeros{n}", np.zeros(n, np.uint8)))
        out.append((f"ramp{n}", (np.arange(n) & 255).astype(np.uint8)))
        out.append((f"rand{n}", rng.integers(0, 256, n, dtype=np.uint8)))
        out.append((f"low{n}", rng.integers(0, 4, n, dtype=np.uint8)))
        out.append((f"period7_{n}", np.resize(np.arange(7, dtype=np.uint8) * 31 + 3, n)))
        mix = rng.integers(0, 256, n, dtype=np.uint8)
        mix[n // 10:] = np.resize(mix[:97], n - n // 10)          # 90 % structured
        out.append((f"struct90_{n}", mix))
        words = [b"lorem ", b"ipsum ", b"dolor ", b"sit ", b"amet, ", b"consectetur ", b"chunk ", b"image "]
        txt = b"".join(words[i] for i in rng.integers(0, len(words), n // 3))[:n]
        out.append((f"text{n}", np.frombuffer(txt.ljust(n, b"."), np.uint8).copy()))
        runs = np.repeat(rng.integers(0, 256, n // 40 + 1, dtype=np.uint8), rng.integers(1, 80, n // 40 + 1))[:n]
        out.append((f"runs{n}", np.resize(runs, n)))
    for dt, ts in [(np.float16, 2), (np.uint16, 2), (np.float32, 4), (np.uint8, 1)]:
        t = synth.tiled_channel(dt, 4096, 8)
        for j, s in enumerate(streams_of(t, ts)):
            out.append((f"tiled_{np.dtype(dt).name}_s{j}", s))
        nat = synth.natural_channel(dt, 4096, 8)
        for j, s in enumerate(streams_of(nat, ts)):
            out.append((f"natural_{np.dtype(dt).name}_s{j}", s))
    # last-bytes / end-of-block edge cases
    e = np.zeros(64, np.uint8); e[-6:] = [1, 2, 3, 4, 5, 6]
    out.append(("tail_edge", e))
    e2 = np.resize(np.array([9, 9, 9, 9, 1, 2, 3, 4, 5, 6, 7, 8], np.uint8), 300)
    out.append(("period12", e2))
    out.append(("max64k", np.resize(rng.integers(0, 256, 4099, dtype=np.uint8), 65535 + 11)))
    return out


def main():
    lz4 = C.CDLL("liblz4.so.1")
    lz4.LZ4_versionString.restype = C.c_char_p
    ver = lz4.LZ4_versionString().decode()
    lz4.LZ4_compress_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lz4.LZ4_compress_fast.restype = C.c_int
    store = {"lz4_version": np.array(ver)}
    names = []
    total = 0
    for name, src in inputs():
        src = np.ascontiguousarray(src, dtype=np.uint8)
        n = src.size
        bound = n + n // 255 + 16
        caps = sorted({n, bound, max(1, n // 2), max(1, (3 * n) // 4)})
        for accel in (1, 5):
            for cap in caps:
                dst = np.zeros(bound + 64, np.uint8)
                r = lz4.LZ4_compress_fast(src.ctypes.data, dst.ctypes.data, n, cap, accel)
                key = f"{name}|a{accel}|c{cap}"
                names.append(key)
                store[f"out|{key}"] = dst[:max(r, 0)].copy()
                store[f"ret|{key}"] = np.int32(r)
                total += 1
        store[f"in|{name}"] = src
        # smallest capacity for which liblz4 still succeeds (success is monotone in cap): pins the
        # `need` output of the oracle / HIP encoder that the chunk assembler relies on.
        if n >= 13:
            for accel in (1,):
                lo, hi = 1, bound
                while lo < hi:
                    mid = (lo + hi) // 2
                    dst = np.zeros(bound + 64, np.uint8)
                    if lz4.LZ4_compress_fast(src.ctypes.data, dst.ctypes.data, n, mid, accel) > 0:
                        hi = mid
                    else:
                        lo = mid + 1
                store[f"need|{name}|a{accel}"] = np.int32(lo)
    store["cases"] = np.array(names)
    path = os.path.join(HERE, "lz4_kat.npz")
    np.savez_compressed(path, **store)
    print(f"liblz4 {ver}: {total} cases -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
