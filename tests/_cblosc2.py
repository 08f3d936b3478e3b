"""Find and bind a GENUINE c-blosc2 shared library, if this box has one (none exists in the image this was written on).

Test infrastructure, like tests/_oracle.py: used by tests/test_conformance_cblosc2.py -- the one test that can turn
"parity unpinned" into "pinned" -- and by bench.py's cpu_baseline_cblosc2 leg (kind "reference").  Everything is called the way
the reference calls it: blosc2_create_cctx with the cparams of blosc2/wrapper.h:338-359, blosc2_compress_ctx (wrapper.h:139),
blosc2_create_dctx / blosc2_decompress_ctx (wrapper.h:395,246).  Never loaded by the product.
"""
import ctypes as C
import os

BLOCK = 32768


def find_blosc2():
    """A genuine c-blosc2 shared library on this box, or None: $CIMG_BLOSC2_LIB, the loader's search path, or the one inside
    an importable python `blosc2` wheel.  (None exists in the image this was written on.)"""
    import ctypes.util
    import glob
    import importlib.util
    cands = [os.environ.get("CIMG_BLOSC2_LIB"), ctypes.util.find_library("blosc2")]
    try:
        spec = importlib.util.find_spec("blosc2")
        if spec and spec.submodule_search_locations:
            for d in spec.submodule_search_locations:
                cands += sorted(glob.glob(os.path.join(d, "**", "libblosc2*.so*"), recursive=True))
                cands += sorted(glob.glob(os.path.join(d, "blosc2_ext*.so")))
    except (ImportError, ValueError):
        pass
    return [c for c in cands if c]


class Blosc2CParamsReal(C.Structure):
    """blosc2_cparams of c-blosc2 2.1x (include/blosc2.h lists the same fields in the same order)."""
    _fields_ = [("compcode", C.c_uint8), ("compcode_meta", C.c_uint8), ("clevel", C.c_uint8), ("use_dict", C.c_int),
                ("typesize", C.c_int32), ("nthreads", C.c_int16), ("blocksize", C.c_int32), ("splitmode", C.c_int32),
                ("schunk", C.c_void_p), ("filters", C.c_uint8 * 6), ("filters_meta", C.c_uint8 * 6),
                ("prefilter", C.c_void_p), ("preparams", C.c_void_p), ("tuner_params", C.c_void_p), ("tuner_id", C.c_int),
                ("instr_codec", C.c_bool), ("codec_params", C.c_void_p), ("filter_params", C.c_void_p * 6)]


class Blosc2DParamsReal(C.Structure):
    _fields_ = [("nthreads", C.c_int16), ("schunk", C.c_void_p), ("postfilter", C.c_void_p), ("postparams", C.c_void_p)]


def open_blosc2():
    """(library, path) of the first candidate that loads and exports the context API the reference binds, else (None, None)."""
    for name in find_blosc2():
        try:
            B = C.CDLL(name)
            for sym in ("blosc2_create_cctx", "blosc2_create_dctx", "blosc2_compress_ctx", "blosc2_decompress_ctx", "blosc2_free_ctx"):
                getattr(B, sym)
        except (OSError, AttributeError):
            continue
        B.blosc2_create_cctx.restype = C.c_void_p
        B.blosc2_create_cctx.argtypes = [Blosc2CParamsReal]
        B.blosc2_create_dctx.restype = C.c_void_p
        B.blosc2_create_dctx.argtypes = [Blosc2DParamsReal]
        B.blosc2_free_ctx.argtypes = [C.c_void_p]
        B.blosc2_compress_ctx.restype = C.c_int
        B.blosc2_compress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        B.blosc2_decompress_ctx.restype = C.c_int
        B.blosc2_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        if hasattr(B, "blosc2_init"):
            B.blosc2_init()
        return B, name
    return None, None


def cctx(B, typesize, nthreads=1, compcode=1, clevel=9, blocksize=BLOCK, filt=1, splitmode=3):
    """The cparams the reference builds (blosc2/wrapper.h:338-359): defaults + blocksize, typesize, AUTO_SPLIT, clevel, nthreads, compcode."""
    cp = Blosc2CParamsReal()
    cp.compcode, cp.clevel, cp.typesize, cp.nthreads, cp.blocksize, cp.splitmode = compcode, clevel, typesize, nthreads, blocksize, splitmode
    cp.filters[5] = filt
    return B.blosc2_create_cctx(cp)


def dctx(B, nthreads=1):
    dp = Blosc2DParamsReal()
    dp.nthreads = nthreads
    return B.blosc2_create_dctx(dp)


def version(B):
    """BLOSC2_VERSION_STRING of the loaded library, if it exports blosc2_get_version_string (2.x does)."""
    try:
        f = B.blosc2_get_version_string
        f.restype = C.c_char_p
        return f().decode()
    except AttributeError:
        return "unknown"


def compress(B, ctx, src, destsize):
    """One blosc2_compress_ctx call: (return code, chunk bytes)."""
    import numpy as np
    s = np.ascontiguousarray(src).view(np.uint8).ravel()
    out = np.zeros(max(destsize, 32) + 64, np.uint8)
    r = B.blosc2_compress_ctx(ctx, s.ctypes.data, s.size, out.ctypes.data, destsize)
    return r, out[:max(r, 0)].tobytes()
