"""ctypes binding of oracle/liborc_cblosc2.so -- the CPU restatement used as the parity checker.

Test infrastructure only (see oracle/orc.h): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(ORACLE_DIR, "liborc_cblosc2.so")

BLOSCLZ, LZ4, LZ4HC, ZLIB, ZSTD = 0, 1, 2, 4, 5
NOFILTER, SHUFFLE, BITSHUFFLE = 0, 1, 2
ALWAYS_SPLIT, NEVER_SPLIT, AUTO_SPLIT, FORWARD_COMPAT_SPLIT = 1, 2, 3, 4
HEADER_LEN = 32


class CParams(C.Structure):
    _fields_ = [("clevel", C.c_int32), ("typesize", C.c_int32), ("blocksize", C.c_int32),
                ("compcode", C.c_int32), ("splitmode", C.c_int32),
                ("filters", C.c_uint8 * 6), ("filters_meta", C.c_uint8 * 6)]


class Geometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("blocksize", "nblocks", "leftover", "split", "nstreams_total", "flags", "memcpyed")]


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(_LIB)
            for f in ("lz4_block.c", "filters.c", "chunk.c", "orc.h", "blosclz.c", "zstd_dl.c", "bench_cpu.c")
            if os.path.exists(os.path.join(ORACLE_DIR, f))):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        u8p = C.c_void_p
        L.orc_lz4_compress_fast.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_lz4_compress_fast.restype = C.c_int
        L.orc_lz4_decompress_safe.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.orc_lz4_decompress_safe.restype = C.c_int
        L.orc_blosclz_compress.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int)]
        L.orc_blosclz_compress.restype = C.c_int
        L.orc_blosclz_decompress.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.orc_blosclz_decompress.restype = C.c_int
        L.orc_blosclz_probe.argtypes = [u8p, C.c_int, C.c_int]
        L.orc_blosclz_probe.restype = C.c_int
        L.orc_blosclz_plan.argtypes = [C.c_int, u8p, C.c_int]
        L.orc_blosclz_plan.restype = C.c_int
        for n in ("orc_shuffle", "orc_unshuffle", "orc_bitshuffle", "orc_bitunshuffle"):
            getattr(L, n).argtypes = [C.c_int, C.c_int, u8p, u8p]
            getattr(L, n).restype = None
        L.orc_blosc2_compress.argtypes = [C.POINTER(CParams), u8p, C.c_int32, u8p, C.c_int32]
        L.orc_blosc2_compress.restype = C.c_int
        L.orc_blosc2_compress_2phase.argtypes = [C.POINTER(CParams), u8p, C.c_int32, u8p, C.c_int32, C.c_int]
        L.orc_blosc2_compress_2phase.restype = C.c_int
        L.orc_blosc2_decompress.argtypes = [u8p, C.c_int32, u8p, C.c_int32]
        L.orc_blosc2_decompress.restype = C.c_int
        L.orc_blosc2_cbuffer_sizes.argtypes = [u8p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_blosc2_cbuffer_sizes.restype = C.c_int
        L.orc_chunk_geometry.argtypes = [C.POINTER(CParams), C.c_int32, C.POINTER(Geometry)]
        L.orc_chunk_geometry.restype = C.c_int
        L.orc_zstd_available.restype = C.c_int
        L.orc_zstd_version.restype = C.c_char_p
        L.orc_zstd_level_of_clevel.argtypes = [C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _bytes_in(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b.view(np.uint8).ravel()
    return np.ascontiguousarray(a)


def cparams(typesize, clevel=9, blocksize=32768, compcode=LZ4, splitmode=AUTO_SPLIT,
            filters=(0, 0, 0, 0, 0, SHUFFLE)):
    """The cparams the reference builds (blosc2/wrapper.h:338-359) for a channel<T>."""
    p = CParams()
    p.clevel, p.typesize, p.blocksize, p.compcode, p.splitmode = clevel, typesize, blocksize, compcode, splitmode
    for i, f in enumerate(filters):
        p.filters[i] = f
    return p


def lz4_compress(src, cap=None, accel=1, want_need=False):
    s = _bytes_in(src)
    n = s.size
    cap = n if cap is None else cap
    out = np.zeros(max(cap, n + n // 255 + 32), dtype=np.uint8)
    need = C.c_int(0)
    r = lib().orc_lz4_compress_fast(_ptr(s), n, _ptr(out), cap, accel, C.byref(need))
    data = out[:max(r, 0)].tobytes()
    return (r, data, need.value) if want_need else (r, data)


def lz4_decompress(comp, n):
    c = _bytes_in(comp)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    r = lib().orc_lz4_decompress_safe(_ptr(c), c.size, _ptr(out), n)
    return r, out[:max(r, 0)].tobytes()


def blosclz_compress(src, clevel=9, cap=None, want_need=False):
    s = _bytes_in(src)
    n = s.size
    cap = n if cap is None else cap
    out = np.zeros(max(cap, n) + 64, dtype=np.uint8)
    need = C.c_int(0)
    r = lib().orc_blosclz_compress(clevel, _ptr(s), n, _ptr(out), cap, C.byref(need))
    data = out[:max(r, 0)].tobytes()
    return (r, data, need.value) if want_need else (r, data)


def blosclz_decompress(comp, n):
    c = _bytes_in(comp)
    out = np.zeros(max(n, 1) + 8, dtype=np.uint8)
    r = lib().orc_blosclz_decompress(_ptr(c), c.size, _ptr(out), n)
    return r, out[:max(r, 0)].tobytes()


def blosclz_plan(src, clevel=9):
    s = _bytes_in(src)
    return lib().orc_blosclz_plan(clevel, _ptr(s), s.size)


def _filter(fn, ts, data):
    s = _bytes_in(data)
    out = np.empty_like(s)
    getattr(lib(), fn)(ts, s.size, _ptr(s), _ptr(out))
    return out


def shuffle(ts, data): return _filter("orc_shuffle", ts, data)
def unshuffle(ts, data): return _filter("orc_unshuffle", ts, data)
def bitshuffle(ts, data): return _filter("orc_bitshuffle", ts, data)
def bitunshuffle(ts, data): return _filter("orc_bitunshuffle", ts, data)


def compress(p, src, destsize=None, two_phase=False, nthreads=1):
    s = _bytes_in(src)
    destsize = s.size + HEADER_LEN if destsize is None else destsize
    out = np.zeros(max(destsize, HEADER_LEN) + 64, dtype=np.uint8)
    if two_phase:
        r = lib().orc_blosc2_compress_2phase(C.byref(p), _ptr(s), s.size, _ptr(out), destsize, nthreads)
    else:
        r = lib().orc_blosc2_compress(C.byref(p), _ptr(s), s.size, _ptr(out), destsize)
    return r, out[:max(r, 0)].tobytes()


def decompress(chunk, nbytes=None):
    c = _bytes_in(chunk)
    if nbytes is None:
        nbytes = cbuffer_sizes(chunk)[0]
    out = np.zeros(max(nbytes, 1), dtype=np.uint8)
    r = lib().orc_blosc2_decompress(_ptr(c), c.size, _ptr(out), nbytes)
    return r, out[:max(r, 0)]


def zstd_available():
    """The box has a libzstd the checker can dlopen (oracle/zstd_dl.c): ORC_ZSTD chunks can be made and decoded."""
    return bool(lib().orc_zstd_available())


def zstd_version():
    return lib().orc_zstd_version().decode()


def cbuffer_sizes(chunk):
    c = _bytes_in(chunk)
    a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
    r = lib().orc_blosc2_cbuffer_sizes(_ptr(c), C.byref(a), C.byref(b), C.byref(d))
    if r < 0:
        raise ValueError(f"cbuffer_sizes error {r}")
    return a.value, b.value, d.value


def geometry(p, nbytes):
    g = Geometry()
    r = lib().orc_chunk_geometry(C.byref(p), nbytes, C.byref(g))
    if r < 0:
        raise ValueError(f"geometry error {r}")
    return g
