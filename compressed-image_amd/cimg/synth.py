"""Seeded synthetic "tiled channel" inputs.

Per channel c: base[y, x] = ((x // 64) * 37 + (y // 64) * 101 + 17 * c) mod 2**k, shifted left by b
bits, plus b bits of PCG64(seed + c) noise; float types are assembled as integer bit patterns above
the smallest normal exponent so every value is a finite normal number.

    dtype    b (noise bits)   k (base bits)   bit pattern
    uint8    4                4               base << 4 | noise
    uint16   6                10              base << 6 | noise
    float16  6                8               0x0400 + (base << 6 | noise)
    uint32   10               22              base << 10 | noise
    float32  10               20              0x00800000 + (base << 10 | noise)
"""
import numpy as np

_SPEC = {
    "uint8": (np.uint8, 4, 4, 0),
    "uint16": (np.uint16, 6, 10, 0),
    "float16": (np.uint16, 6, 8, 0x0400),
    "uint32": (np.uint32, 10, 22, 0),
    "float32": (np.uint32, 10, 20, 0x00800000),
}


def tiled_channel(dtype, width, height, c=0, seed=1234):
    """One (height, width) channel of the tiled-gradient + noise family."""
    name = np.dtype(dtype).name
    store, b, k, bias = _SPEC[name]
    rng = np.random.Generator(np.random.PCG64(seed + c))
    x = (np.arange(width, dtype=np.uint32) // 64) * 37
    y = (np.arange(height, dtype=np.uint32) // 64) * 101
    base = (x[None, :] + y[:, None] + np.uint32(17 * c)) & np.uint32((1 << k) - 1)
    noise = rng.integers(0, 1 << b, size=(height, width), dtype=np.uint32)
    bits = ((base << np.uint32(b)) | noise) + np.uint32(bias)
    return np.ascontiguousarray(bits.astype(store)).view(np.dtype(dtype))


def zero_channel(dtype, width, height):
    """All-zero boundary set (run-token / special-zero chunk path)."""
    return np.zeros((height, width), dtype=dtype)


def random_channel(dtype, width, height, c=0, seed=4321):
    """Uniform-random bytes boundary set (store-raw / memcpyed chunk path)."""
    rng = np.random.Generator(np.random.PCG64(seed + c))
    it = np.dtype(dtype).itemsize
    raw = rng.integers(0, 256, size=(height, width * it), dtype=np.uint8)
    return raw.view(np.uint16 if it == 2 else (np.uint32 if it == 4 else np.uint8)).view(
        np.dtype(dtype) if np.dtype(dtype).kind != "f" else np.dtype(dtype)).reshape(height, width)


def natural_channel(dtype, width, height, c=0, seed=777):
    """Smooth ramps + edges + mild noise: many short LZ4 sequences (stress set, not a BASELINE config)."""
    name = np.dtype(dtype).name
    store, b, k, bias = _SPEC[name]
    rng = np.random.Generator(np.random.PCG64(seed + c))
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    f = (np.sin(xx / 37.0 + c) + np.cos(yy / 53.0) + ((xx // 97 + yy // 61) % 3)) * 0.2 + 0.5
    top = (1 << (k + b)) - 8
    vals = np.clip(f, 0, 1) * top
    noise = rng.integers(0, 4, size=(height, width), dtype=np.uint32) * (rng.random((height, width)) < 0.2)
    bits = (vals.astype(np.uint64).astype(np.uint32) & ~np.uint32(3)) + noise.astype(np.uint32) + np.uint32(bias)
    return np.ascontiguousarray(bits.astype(store)).view(np.dtype(dtype))
