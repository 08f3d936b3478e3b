"""ctypes binding of tests/emu/libcimg_emu.so: the gfx950 kernel bodies run on the host lane emulator.

Test infrastructure only.  It checks kernel LOGIC on CPU; the `-m gpu` tests check the real kernels.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_CSRC = os.path.join(_ROOT, "compressed-image_amd", "csrc")
_SRC = os.path.join(_HERE, "emu", "emu.cpp")
_LIB = os.path.join(_HERE, "emu", "libcimg_emu.so")


class CParams(C.Structure):
    _fields_ = [("typesize", C.c_int32), ("clevel", C.c_int32), ("blocksize", C.c_int32),
                ("compcode", C.c_int32), ("splitmode", C.c_int32),
                ("filters", C.c_uint8 * 6), ("filters_meta", C.c_uint8 * 6)]


def build():
    if os.environ.get("CIMG_EMU_LIB"):                 # (a test that runs this module against a differently built emulator)
        return os.environ["CIMG_EMU_LIB"]
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
                               "-fno-strict-aliasing", "-I", _CSRC, _SRC, "-o", _LIB])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.emu_set_write_order.argtypes = [C.c_int]
        L.emu_compress_batch.argtypes = [C.POINTER(CParams), C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.emu_compress_batch.restype = C.c_int
        L.emu_decompress_batch.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.emu_decompress_batch.restype = C.c_int
        L.emu_lz4_encode.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.emu_lz4_encode.restype = C.c_int
        L.emu_lz4_decode.argtypes = [vp, C.c_int, vp, C.c_int]
        L.emu_lz4_decode.restype = C.c_int
        L.emu_blosclz_encode.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.emu_blosclz_encode.restype = C.c_int
        L.emu_blosclz_decode.argtypes = [vp, C.c_int, vp, C.c_int]
        L.emu_blosclz_decode.restype = C.c_int
        _lib = L
    return _lib


def cparams(typesize, clevel=9, blocksize=32768, compcode=1, splitmode=3, filters=(0, 0, 0, 0, 0, 1)):
    p = CParams()
    p.typesize, p.clevel, p.blocksize, p.compcode, p.splitmode = typesize, clevel, blocksize, compcode, splitmode
    for i, f in enumerate(filters):
        p.filters[i] = f
    return p


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x).view(np.uint8).ravel()
    return np.frombuffer(bytes(x), np.uint8).copy()


def set_write_order(o):
    lib().emu_set_write_order(o)


def set_zstd_lose_fused(on):
    """Test hook: the fused kernel does not run behind the replay (blocks the walk refused stay undecoded)."""
    lib().emu_set_zstd_lose_fused.argtypes = [C.c_int]
    lib().emu_set_zstd_lose_fused(int(on))


def set_zstd_plan(cap):
    """The zstd read path of the emulated batch decode: -1 = walk + replay launches with plans of the block area (the engine's
    default), 0 = the fused kernels only, n = plans of n bytes (small: plans overflow and their blocks go to the fused kernels)."""
    lib().emu_set_zstd_plan(int(cap))


def set_zstd_lanes(n):
    """Blocks a wave of the lane decoder (cimg_zstd_seq) takes side by side; 0: the walkers decode the sequences themselves."""
    lib().emu_set_zstd_lanes(int(n))


def zstd_refused():
    """Blocks whose plan did not fit its slot since the last call."""
    lib().emu_zstd_refused.restype = C.c_long
    return int(lib().emu_zstd_refused())


def set_lean(on):
    """Run (default) or skip the lean decode kernel in front of the general one."""
    lib().emu_set_lean(1 if on else 0)


def set_block_items(mode):
    """0: every work item of the split encode launch is one byte plane; 1: a whole block; 2: the first half of the blocks whole,
    the rest plane by plane (the mixed queue the engine builds for a small batch)"""
    lib().emu_set_block_items(int(mode))


def lean_blocks():
    """Blocks the lean decode kernel produced since the last call."""
    lib().emu_lean_blocks.restype = C.c_long
    return int(lib().emu_lean_blocks())


def set_enc_rt(on):
    """1: LZ4 streams through the register-table form of the encoder (csrc/encode_rt_kernel.h), 0: the LDS-table form (default)."""
    lib().emu_set_enc_rt.argtypes = [C.c_int]
    lib().emu_set_enc_rt(int(on))


def lz4_encode(src, cap=None, accel=1):
    s = _u8(src)
    n = s.size
    cap = n if cap is None else cap
    out = np.full(n + n // 255 + 64, 0xAB, np.uint8)
    need = C.c_int(0)
    r = lib().emu_lz4_encode(_p(s), n, _p(out), cap, accel, C.byref(need))
    return r, out[:max(r, 0)].tobytes(), need.value


def lz4_decode(comp, n):
    c = _u8(comp)
    out = np.zeros(max(n, 1), np.uint8)
    r = lib().emu_lz4_decode(_p(c), c.size, _p(out), n)
    return r, out[:n].tobytes()


def zstd_decode(frame, cap):
    """One zstd frame through csrc/zstd_decode.h: (return code = regenerated size or < 0, bytes)."""
    c = _u8(frame)
    out = np.zeros(max(cap, 1), np.uint8)
    r = lib().emu_zstd_decode(_p(c), c.size, _p(out), cap)
    return r, out[:max(r, 0)].tobytes()


def blosclz_encode(src, cap=None, clevel=9):
    s = _u8(src)
    n = s.size
    cap = n if cap is None else cap
    out = np.full(max(cap, n) + 64, 0xAB, np.uint8)
    need = C.c_int(0)
    r = lib().emu_blosclz_encode(_p(s), n, _p(out), cap, clevel, C.byref(need))
    return r, out[:max(r, 0)].tobytes(), need.value


def blosclz_decode(comp, n):
    c = _u8(comp)
    out = np.zeros(max(n, 1), np.uint8)
    r = lib().emu_blosclz_decode(_p(c), c.size, _p(out), n)
    return r, out[:n].tobytes()


def compress_batch(p, raw, nbytes_list, destsize_list, stride=None):
    """raw: the chunks' pixels back to back. Returns (rc, [cbytes], [chunk bytes])."""
    raw = _u8(raw)
    n = len(nbytes_list)
    nb = np.asarray(nbytes_list, np.int32)
    ds = np.asarray(destsize_list, np.int32)
    raw_off = np.concatenate([[0], np.cumsum(nb[:-1], dtype=np.int64)]).astype(np.int64)
    stride = int(max(ds.max(), 32)) + 64 if stride is None else stride
    comp_off = (np.arange(n, dtype=np.int64) * stride).astype(np.int64)
    comp = np.full(n * stride + 64, 0x5A, np.uint8)
    cbytes = np.zeros(n, np.int32)
    rc = lib().emu_compress_batch(C.byref(p), n, _p(raw), _p(raw_off), _p(nb), _p(comp), _p(comp_off), _p(ds), _p(cbytes))
    chunks = [comp[comp_off[i]:comp_off[i] + max(cbytes[i], 0)].tobytes() for i in range(n)]
    return rc, cbytes.tolist(), chunks


def decompress_batch(chunks, nbytes_list, blocksize_list, misalign=0):
    n = len(chunks)
    sizes = [len(c) for c in chunks]
    offs, o = [], misalign
    for s in sizes:
        offs.append(o)
        o += (s + 15) // 16 * 16 + 16 + misalign
    comp_off = np.asarray(offs, np.int64)
    comp = np.zeros(o + 64, np.uint8)
    for off, c in zip(comp_off, chunks):
        comp[off:off + len(c)] = np.frombuffer(c, np.uint8)
    nb = np.asarray(nbytes_list, np.int32)
    bs = np.asarray(blocksize_list, np.int32)
    raw_off = np.concatenate([[0], np.cumsum(nb[:-1], dtype=np.int64)]).astype(np.int64)
    raw = np.full(int(nb.sum()) + 64, 0x77, np.uint8)
    status = np.zeros(n, np.int32)
    rc = lib().emu_decompress_batch(n, _p(comp), _p(comp_off), _p(nb), _p(bs), _p(raw), _p(raw_off), _p(status))
    return rc, status.tolist(), [raw[raw_off[i]:raw_off[i] + nb[i]].copy() for i in range(n)]


def deinterleave(interleaved, nchannels, typesize):
    """interleaved: uint8 array of npixels * nchannels * typesize bytes.  Returns (nchannels, npixels * typesize) uint8 planes."""
    src = _u8(interleaved)
    npixels = src.size // (nchannels * typesize)
    stride = (npixels * typesize + 15) & ~15
    dst = np.full(max(stride * nchannels, 16), 0xEE, np.uint8)
    L = lib()
    L.emu_deinterleave.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64]
    rc = L.emu_deinterleave(_p(src), nchannels, typesize, npixels, _p(dst), stride)
    if rc < 0:
        raise ValueError("emu_deinterleave rejected the arguments (%d)" % rc)
    return np.stack([dst[c * stride:c * stride + npixels * typesize] for c in range(nchannels)]) if npixels else np.zeros((nchannels, 0), np.uint8)
