#!/usr/bin/env python3
"""Generate tests/golden/blosclz_kat.npz: per-stream BloscLZ payloads emitted by the system c-blosc *1* library
(1.21.0 at /opt/conda/lib/libblosc.so.1, which bundles BloscLZ 2.3.0), through its PUBLIC API only.

Why this library: the reference's codec (c-blosc2 >= 2.17) is an empty submodule and is not installed; this is
the only BloscLZ build in the image.  c-blosc2 vendors a later BloscLZ, so the vectors pin oracle/blosclz.c to
"c-blosc1's blosclz 2.3.0" -- the stream format is shared, the encoder's parameter tables are not.

How a stream payload is obtained: blosc_compress_ctx(clevel, doshuffle = 0, typesize = 1, n, ..., "blosclz",
blocksize = n, 1 thread) makes ONE block with ONE stream and calls blosclz_compress(clevel, src, n, dest,
maxout = n).  The frame is 16-byte header, int32 bstarts[1], int32 csize, payload; csize == n means the codec
returned 0 (or n) and the stream was stored raw.  n >= 128 (smaller buffers are memcpyed by c-blosc 1 itself).

Run:  python tests/golden/make_blosclz_golden.py      (output committed)
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "compressed-image_amd"))
from cimg import synth  # noqa: E402


def families(rng):
    """name -> callable(n) -> uint8[n]; mostly what byte planes / bit rows of images look like"""
    def shuf_plane(dtype, plane, natural):
        def f(n):
            fn = synth.natural_channel if natural else synth.tiled_channel
            a = fn(dtype, 4096, max(1, -(-n // 4096) + 1)).view(np.uint8).reshape(-1, np.dtype(dtype).itemsize)
            return np.ascontiguousarray(a[:n, plane])
        return f

    def pasted(n):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(4, 40))):
            ln = int(rng.integers(3, 300))
            if n <= ln + 2:
                continue
            s, d = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
            a[d:d + ln] = a[s:s + ln].copy()
        return a

    def walk(n):
        return (np.cumsum(rng.integers(-2, 3, n)) & 0xFF).astype(np.uint8)

    def period(n):
        p = int(rng.integers(1, 300))
        return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()

    def steps(n):
        return np.resize(np.repeat(rng.integers(0, 256, n // 8 + 1, dtype=np.uint8), int(rng.integers(1, 33))), n).astype(np.uint8)

    return {
        "tiled_u16_hi": shuf_plane(np.uint16, 1, False), "tiled_u16_lo": shuf_plane(np.uint16, 0, False),
        "natural_u16_hi": shuf_plane(np.uint16, 1, True), "natural_f32_b2": shuf_plane(np.float32, 2, True),
        "tiled_u8": shuf_plane(np.uint8, 0, False),
        "pasted": pasted, "walk": walk, "period": period, "steps": steps,
        "two_bits": lambda n: rng.integers(0, 4, n, dtype=np.uint8),
        "zeros_then_noise": lambda n: np.concatenate([np.zeros(n - n // 3, np.uint8), rng.integers(0, 256, n // 3, dtype=np.uint8)]),
        "random": lambda n: rng.integers(0, 256, n, dtype=np.uint8),
        "constant": lambda n: np.full(n, 7, np.uint8),
    }


def main():
    b = C.CDLL("/opt/conda/lib/libblosc.so.1")
    b.blosc_get_version_string.restype = C.c_char_p
    ver = b.blosc_get_version_string().decode()
    lz_name, lz_ver = C.c_char_p(), C.c_char_p()
    b.blosc_get_complib_info(b"blosclz", C.byref(lz_name), C.byref(lz_ver))
    b.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    b.blosc_compress_ctx.restype = C.c_int
    rng = np.random.Generator(np.random.PCG64(20))
    fams = families(rng)
    sizes = [128, 191, 1000, 4096, 8192, 16384, 32768, 65535]
    noisy = {"random", "two_bits", "zeros_then_noise", "pasted", "tiled_u16_lo", "tiled_u8", "walk"}
    store = {"blosc_version": np.array(ver), "blosclz_version": np.array(lz_ver.value.decode())}
    names, coded, raw = [], 0, 0
    k = 0
    for fname, gen in fams.items():
        for n in sizes:
            if fname in noisy and n > (16384 if fname in ("pasted", "walk") else 4096):
                continue                                   # high-entropy inputs bloat the fixture
            src = np.ascontiguousarray(gen(n)).astype(np.uint8)
            assert src.size == n
            for clevel in sorted({9, 5, 1 + k % 9}):
                k += 1
                dst = np.zeros(n + 4096, np.uint8)
                r = b.blosc_compress_ctx(clevel, 0, 1, n, src.ctypes.data, dst.ctypes.data, dst.size, b"blosclz", n, 1)
                assert r > 0
                flags = int(dst[2])
                nbytes, blocksize, cbytes = struct.unpack_from("<iii", dst.tobytes(), 4)
                assert nbytes == n and blocksize == n and cbytes == r and not flags & 0x2, (fname, n, clevel)
                (bstart,) = struct.unpack_from("<i", dst.tobytes(), 16)
                (csize,) = struct.unpack_from("<i", dst.tobytes(), bstart)
                payload = dst[bstart + 4:bstart + 4 + csize].copy()
                assert bstart + 4 + csize == r
                name = f"{fname}|{n}|{clevel}"
                names.append(name)
                store["in|" + f"{fname}|{n}"] = src
                if csize == n:                             # stored raw: the codec gave up or did not fit n bytes
                    store["out|" + name] = np.zeros(0, np.uint8)
                    raw += 1
                else:
                    store["out|" + name] = payload
                    coded += 1
    store["cases"] = np.array(names)
    path = os.path.join(HERE, "blosclz_kat.npz")
    np.savez_compressed(path, **store)
    print(f"c-blosc {ver} / BloscLZ {lz_ver.value.decode()}: {len(names)} vectors ({coded} coded, {raw} raw) -> {path} "
          f"({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
