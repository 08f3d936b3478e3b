"""ctypes binding of libcimg_hip.so (include/cimg_hip.h + include/blosc2.h).

Plumbing for bench.py, __graft_entry__.py and the GPU tests: it only marshals pointers and sizes
into the C ABI.  There is no fallback: if the library is missing or no gfx950 device is present the
calls raise.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("CIMG_LIB") or os.path.join(_PKG, "libcimg_hip.so")   # CIMG_LIB: diagnostic builds

K_ENCODE, K_LAYOUT, K_EMIT, K_DECODE, K_DEINTERLEAVE, K_DECODE_ZSTD, K_ENCODE_ZSTD, K_ZSTD_WALK, K_ZSTD_REPLAY, K_ZSTD_FUSED, K_ZSTD_SEQ, K_ZSTD_LIT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
# names by timing id (cimg_kernel_name).  K_ENCODE times whichever of cimg_encode_streams / _blosclz the codec selects; K_DECODE
# times the pair cimg_decode_lean + cimg_decode_blocks (the second only runs for blocks the first left): bench.py reports it under
# the kernel that did the work
KERNELS = ("cimg_encode_streams", "cimg_layout_chunks", "cimg_emit_blocks", "cimg_decode_blocks", "cimg_deinterleave",
           "cimg_decode_zstd", "cimg_encode_streams_zstd", "cimg_zstd_walk", "cimg_zstd_replay", "cimg_decode_zstd_fused", "cimg_zstd_seq", "cimg_zstd_lit")
# (K_DECODE_ZSTD times the zstd read path of a batch as a whole -- cimg_zstd_walk + cimg_zstd_lit + cimg_zstd_seq + cimg_zstd_replay, and
# cimg_decode_zstd behind them for blocks the walk refused; the last five ids time those launches one by one)
BLOSCLZ, LZ4, LZ4HC, ZLIB, ZSTD = 0, 1, 2, 4, 5
NOFILTER, SHUFFLE, BITSHUFFLE = 0, 1, 2
MAX_OVERHEAD = 32

EXPORTS = (
    # include/cimg_hip.h
    "cimg_cparams_init", "cimg_engine_create", "cimg_engine_destroy", "cimg_last_error",
    "cimg_engine_synchronize", "cimg_engine_lock", "cimg_engine_unlock", "cimg_engine_stream", "cimg_compress_batch_device",
    "cimg_decompress_batch_device", "cimg_compress_batch_host", "cimg_decompress_batch_host", "cimg_decompress_batch_host_sized",
    "cimg_compress_batch_host_begin", "cimg_compress_batch_host_fetch", "cimg_compress_batch_host_packed",
    "cimg_deinterleave_device", "cimg_compress_batch_host_interleaved_begin",
    "cimg_compress_batch_device_begin", "cimg_compress_batch_device_fetch", "cimg_decompress_batch_device_begin", "cimg_decompress_batch_device_fetch",
    "cimg_decompress_batch_device_sized", "cimg_decompress_batch_device_begin_sized",
    "cimg_device_malloc", "cimg_device_free", "cimg_memcpy_h2d", "cimg_memcpy_d2h", "cimg_host_malloc", "cimg_host_free",
    "cimg_engine_enable_timing", "cimg_engine_reset_timing", "cimg_engine_kernel_time", "cimg_engine_kernel_samples", "cimg_engine_decode_stats", "cimg_engine_zstd_stats", "cimg_kernel_name",
    "cimg_engine_debug_stamps", "cimg_engine_read_stamps", "cimg_shared_engine", "cimg_context_cparams",
    # include/blosc2.h
    "blosc2_create_cctx", "blosc2_create_dctx", "blosc2_free_ctx", "blosc2_compress_ctx",
    "blosc2_decompress_ctx", "blosc2_cbuffer_sizes", "blosc2_schunk_new", "blosc2_schunk_free",
    "blosc2_schunk_append_chunk", "register_filters", "print_error",
)


class CodecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"blosc2 error {code}: {msg}")
        self.code = code


class CParams(C.Structure):
    _fields_ = [("typesize", C.c_int32), ("clevel", C.c_int32), ("blocksize", C.c_int32),
                ("compcode", C.c_int32), ("splitmode", C.c_int32),
                ("filters", C.c_uint8 * 6), ("filters_meta", C.c_uint8 * 6)]


class Blosc2CParams(C.Structure):
    _fields_ = [("compcode", C.c_uint8), ("compcode_meta", C.c_uint8), ("clevel", C.c_uint8),
                ("use_dict", C.c_int), ("typesize", C.c_int32), ("nthreads", C.c_int16),
                ("blocksize", C.c_int32), ("splitmode", C.c_int32), ("schunk", C.c_void_p),
                ("filters", C.c_uint8 * 6), ("filters_meta", C.c_uint8 * 6),
                ("prefilter", C.c_void_p), ("preparams", C.c_void_p), ("tuner_params", C.c_void_p),
                ("tuner_id", C.c_int), ("instr_codec", C.c_bool), ("codec_params", C.c_void_p),
                ("filter_params", C.c_void_p * 6)]


class Blosc2DParams(C.Structure):
    _fields_ = [("nthreads", C.c_int16), ("schunk", C.c_void_p), ("postfilter", C.c_void_p),
                ("postparams", C.c_void_p)]


_lib = None


def load():
    """dlopen the in-tree library (raises OSError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(or `make -C compressed-image_amd`)")
    L = C.CDLL(LIB_PATH)
    vp, i32p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.cimg_cparams_init.argtypes = [C.POINTER(CParams), C.c_int32]
    L.cimg_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.cimg_engine_destroy.argtypes = [vp]
    L.cimg_last_error.argtypes = [vp]
    L.cimg_last_error.restype = C.c_char_p
    L.cimg_engine_synchronize.argtypes = [vp]
    L.cimg_engine_stream.argtypes = [vp]
    L.cimg_engine_stream.restype = vp
    L.cimg_compress_batch_device.argtypes = [vp, C.POINTER(CParams), C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_device.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_device_sized.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_device_begin_sized.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_deinterleave_device.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int64]
    L.cimg_compress_batch_host_interleaved_begin.argtypes = [vp, C.POINTER(CParams), C.c_int32, C.c_int64, vp, C.c_int32, vp, vp, vp, vp]
    L.cimg_compress_batch_device_begin.argtypes = [vp, C.POINTER(CParams), C.c_int32, vp, vp, vp, vp, vp, vp]
    L.cimg_compress_batch_device_fetch.argtypes = [vp, C.c_int32, vp]
    L.cimg_decompress_batch_device_begin.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_device_fetch.argtypes = [vp, vp]
    L.cimg_compress_batch_host.argtypes = [vp, C.POINTER(CParams), C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_host.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.cimg_decompress_batch_host_sized.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.cimg_device_malloc.argtypes = [vp, C.c_size_t]
    L.cimg_device_malloc.restype = vp
    L.cimg_device_free.argtypes = [vp, vp]
    L.cimg_host_malloc.argtypes = [C.c_size_t]
    L.cimg_host_malloc.restype = vp
    L.cimg_host_free.argtypes = [vp]
    L.cimg_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.cimg_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.cimg_engine_enable_timing.argtypes = [vp, C.c_int]
    L.cimg_engine_reset_timing.argtypes = [vp]
    L.cimg_engine_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.cimg_engine_kernel_samples.argtypes = [vp, C.c_int, vp, C.c_int]
    L.cimg_engine_decode_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cimg_engine_decode_stats.restype = None
    L.cimg_engine_zstd_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cimg_engine_zstd_stats.restype = None
    L.cimg_kernel_name.argtypes = [C.c_int]
    L.cimg_kernel_name.restype = C.c_char_p
    L.cimg_engine_debug_stamps.argtypes = [vp, C.c_int]
    L.cimg_engine_read_stamps.argtypes = [vp, C.c_int, vp, C.c_int]
    L.blosc2_create_cctx.argtypes = [Blosc2CParams]
    L.blosc2_create_cctx.restype = vp
    L.blosc2_create_dctx.argtypes = [Blosc2DParams]
    L.blosc2_create_dctx.restype = vp
    L.blosc2_free_ctx.argtypes = [vp]
    L.blosc2_compress_ctx.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
    L.blosc2_decompress_ctx.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
    L.blosc2_cbuffer_sizes.argtypes = [vp, i32p, i32p, i32p]
    L.print_error.argtypes = [C.c_int]
    L.print_error.restype = C.c_char_p
    _lib = L
    return L


def cparams(typesize, clevel=9, blocksize=32768, compcode=LZ4, splitmode=3, filters=(0, 0, 0, 0, 0, SHUFFLE)):
    p = CParams()
    load().cimg_cparams_init(C.byref(p), typesize)
    p.clevel, p.blocksize, p.compcode, p.splitmode = clevel, blocksize, compcode, splitmode
    for i, f in enumerate(filters):
        p.filters[i] = f
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _i64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int64))


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


class DeviceBuffer:
    """A device allocation owned through the C ABI (no HIP / torch types involved)."""

    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        self.ptr = load().cimg_device_malloc(engine.handle, self.nbytes)
        if not self.ptr:
            raise CodecError(-4, engine.last_error())
        engine._buffers.add(self)

    def upload(self, host, offset=0):
        h = np.ascontiguousarray(host).view(np.uint8).ravel()
        self.engine._check(load().cimg_memcpy_h2d(self.engine.handle, self.ptr + offset, _ptr(h), h.size))

    def download(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(n, np.uint8)
        self.engine._check(load().cimg_memcpy_d2h(self.engine.handle, _ptr(out), self.ptr + offset, n))
        return out

    def free(self):
        # an engine that is closed first frees its buffers itself (Engine.close), so the handle is live here
        if self.ptr and self.engine.handle:
            load().cimg_device_free(self.engine.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One MI355X codec engine (stream + scratch) -- wraps cimg_engine_*."""

    def __init__(self, device=-1):
        import weakref
        self._buffers = weakref.WeakSet()
        self.handle = C.c_void_p()
        rc = load().cimg_engine_create(device, C.byref(self.handle))
        if rc != 0:
            raise CodecError(rc, load().cimg_last_error(None).decode())

    def close(self):
        if self.handle:
            for b in list(self._buffers):          # device memory goes before the engine, whatever the GC order
                b.free()
            load().cimg_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return load().cimg_last_error(self.handle).decode()

    def _check(self, rc):
        if rc < 0:
            raise CodecError(rc, self.last_error())
        return rc

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def synchronize(self):
        self._check(load().cimg_engine_synchronize(self.handle))

    # ---- device-resident batches (pointers are raw device addresses: DeviceBuffer.ptr or tensor.data_ptr()) ----
    def compress_device(self, p, d_raw, raw_off, nbytes, d_comp, comp_off, destsize):
        raw_off, comp_off, nbytes, destsize = _i64(raw_off), _i64(comp_off), _i32(nbytes), _i32(destsize)
        cbytes = np.zeros(nbytes.size, np.int32)
        self._check(load().cimg_compress_batch_device(self.handle, C.byref(p), nbytes.size, d_raw, _ptr(raw_off), _ptr(nbytes),
                                                      d_comp, _ptr(comp_off), _ptr(destsize), _ptr(cbytes)))
        return cbytes

    def decompress_device(self, d_comp, comp_off, nbytes, blocksize, d_raw, raw_off, check=True, comp_size=None):
        raw_off, comp_off, nbytes, blocksize = _i64(raw_off), _i64(comp_off), _i32(nbytes), _i32(blocksize)
        status = np.zeros(nbytes.size, np.int32)
        if comp_size is not None:       # the caller knows how many bytes each compressed buffer holds
            comp_size = _i32(comp_size)
            rc = load().cimg_decompress_batch_device_sized(self.handle, nbytes.size, d_comp, _ptr(comp_off), _ptr(comp_size), _ptr(nbytes),
                                                           _ptr(blocksize), d_raw, _ptr(raw_off), _ptr(status))
        else:
            rc = load().cimg_decompress_batch_device(self.handle, nbytes.size, d_comp, _ptr(comp_off), _ptr(nbytes), _ptr(blocksize),
                                                     d_raw, _ptr(raw_off), _ptr(status))
        if check:
            self._check(rc)
        return status

    def deinterleave_device(self, d_interleaved, nchannels, typesize, npixels, d_planar, plane_stride):
        """interleaved pixels -> one plane per channel (plane c at d_planar + c * plane_stride), enqueued on the engine's stream"""
        self._check(load().cimg_deinterleave_device(self.handle, d_interleaved, nchannels, typesize, npixels, d_planar, plane_stride))

    # the same in two steps (include/cimg_hip.h): the kernels of one compress and one decompress batch may be in flight together
    def compress_device_begin(self, p, d_raw, raw_off, nbytes, d_comp, comp_off, destsize):
        raw_off, comp_off, nbytes, destsize = _i64(raw_off), _i64(comp_off), _i32(nbytes), _i32(destsize)
        self._check(load().cimg_compress_batch_device_begin(self.handle, C.byref(p), nbytes.size, d_raw, _ptr(raw_off), _ptr(nbytes),
                                                            d_comp, _ptr(comp_off), _ptr(destsize)))
        return nbytes.size

    def compress_device_fetch(self, nchunks):
        cbytes = np.zeros(nchunks, np.int32)
        self._check(load().cimg_compress_batch_device_fetch(self.handle, nchunks, _ptr(cbytes)))
        return cbytes

    def decompress_device_begin(self, d_comp, comp_off, nbytes, blocksize, d_raw, raw_off):
        raw_off, comp_off, nbytes, blocksize = _i64(raw_off), _i64(comp_off), _i32(nbytes), _i32(blocksize)
        self._check(load().cimg_decompress_batch_device_begin(self.handle, nbytes.size, d_comp, _ptr(comp_off), _ptr(nbytes), _ptr(blocksize),
                                                              d_raw, _ptr(raw_off)))
        return nbytes.size

    def decompress_device_fetch(self, nchunks, check=True):
        status = np.zeros(nchunks, np.int32)
        rc = load().cimg_decompress_batch_device_fetch(self.handle, _ptr(status))
        if check:
            self._check(rc)
        return status

    def roundtrip_calls(self, p, d_raw, raw_off, nbytes, d_comp, comp_off, destsize, blocksize, d_out):
        """The four calls of a device-resident round trip (compress _begin, decompress _begin, both _fetch) with their arguments
        marshalled ONCE: returns step() -> cbytes.  For callers that repeat the same batch geometry (bench.py): the per-call numpy
        -> ctypes conversions of the methods above cost the host more than the C calls themselves."""
        L = load()
        raw_off, comp_off, nbytes, destsize, blocksize = _i64(raw_off), _i64(comp_off), _i32(nbytes), _i32(destsize), _i32(blocksize)
        n = int(nbytes.size)
        cbytes = np.zeros(n, np.int32)
        status = np.zeros(n, np.int32)
        keep = (raw_off, comp_off, nbytes, destsize, blocksize, cbytes, status, p)          # the pointers below point into these
        h, pp = self.handle, C.byref(p)
        a_raw, a_comp, a_nb, a_ds, a_bs = _ptr(raw_off), _ptr(comp_off), _ptr(nbytes), _ptr(destsize), _ptr(blocksize)
        a_cb, a_st = _ptr(cbytes), _ptr(status)
        vr, vc, vo = C.c_void_p(d_raw), C.c_void_p(d_comp), C.c_void_p(d_out)
        cb_begin, db_begin, cb_fetch, db_fetch = (L.cimg_compress_batch_device_begin, L.cimg_decompress_batch_device_begin,
                                                  L.cimg_compress_batch_device_fetch, L.cimg_decompress_batch_device_fetch)
        check = self._check

        def step(_keep=keep):
            check(cb_begin(h, pp, n, vr, a_raw, a_nb, vc, a_comp, a_ds))
            check(db_begin(h, n, vc, a_comp, a_nb, a_bs, vo, a_raw))
            check(cb_fetch(h, n, a_cb))
            check(db_fetch(h, a_st))
            return cbytes
        return step

    def stream_handle(self):
        """the engine's hipStream_t as an integer (torch.cuda.ExternalStream(handle) enqueues a caller's own kernels on it)"""
        return int(load().cimg_engine_stream(self.handle) or 0)

    def single_chunk_calls(self, p, d_work, d_comp, comp_off, chunk_bytes, destsize, blocksize):
        """get_chunk(i) / set_chunk(i) of a channel whose chunks live at d_comp + comp_off[i] (the random-access loop of the
        reference's examples/lazy_channels/main.cpp:35-52; channel.h:502-538): one device-resident decompress call of ONE chunk into
        d_work, one compress call of ONE chunk from d_work back into its slot.  Arguments marshalled once (a per-call numpy -> ctypes
        conversion costs the host more than a single-chunk launch).  set_chunk returns the chunk's new compressed size."""
        L = load()
        comp_off = _i64(comp_off)
        n = int(comp_off.size)
        zero = _i64([0]); nb = _i32([chunk_bytes]); ds = _i32([destsize]); bs = _i32([blocksize])
        cb = np.zeros(1, np.int32); st = np.zeros(1, np.int32)
        keep = (comp_off, zero, nb, ds, bs, cb, st, p)
        h, pp = self.handle, C.byref(p)
        base = comp_off.ctypes.data
        a_zero, a_nb, a_ds, a_bs, a_cb, a_st = _ptr(zero), _ptr(nb), _ptr(ds), _ptr(bs), _ptr(cb), _ptr(st)
        vw, vc = C.c_void_p(d_work), C.c_void_p(d_comp)
        dec, enc, check = L.cimg_decompress_batch_device, L.cimg_compress_batch_device, self._check

        def get_chunk(i, _keep=keep):
            check(dec(h, 1, vc, C.c_void_p(base + 8 * i), a_nb, a_bs, vw, a_zero, a_st))

        def set_chunk(i, _keep=keep):
            check(enc(h, pp, 1, vw, a_zero, a_nb, vc, C.c_void_p(base + 8 * i), a_ds, a_cb))
            return int(cb[0])
        assert n > 0
        return get_chunk, set_chunk

    # ---- host-resident batches ----
    def compress_host(self, p, raw, nbytes, destsize):
        """raw: numpy array holding the chunks back to back.  Returns list of chunk bytes (b'' = does not fit)."""
        raw = np.ascontiguousarray(raw).view(np.uint8).ravel()
        nbytes, destsize = _i32(nbytes), _i32(destsize)
        raw_off = _i64(np.concatenate([[0], np.cumsum(nbytes[:-1], dtype=np.int64)]))
        stride = int(destsize.max()) + 32
        comp_off = _i64(np.arange(nbytes.size, dtype=np.int64) * stride)
        comp = np.zeros(stride * nbytes.size, np.uint8)
        cbytes = np.zeros(nbytes.size, np.int32)
        self._check(load().cimg_compress_batch_host(self.handle, C.byref(p), nbytes.size, _ptr(raw), _ptr(raw_off), _ptr(nbytes),
                                                    _ptr(comp), _ptr(comp_off), _ptr(destsize), _ptr(cbytes)))
        return [comp[o:o + max(c, 0)].tobytes() for o, c in zip(comp_off, cbytes)]

    def decompress_host(self, chunks, check=True):
        """chunks: list of bytes.  Returns (list of uint8 arrays, status array)."""
        sizes = [len(c) for c in chunks]
        comp_off = _i64(np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.int64)]))
        comp = np.frombuffer(b"".join(chunks), np.uint8)
        nb = _i32([int.from_bytes(c[4:8], "little", signed=True) for c in chunks])
        raw_off = _i64(np.concatenate([[0], np.cumsum(nb[:-1], dtype=np.int64)]))
        raw = np.zeros(max(int(nb.sum()), 1), np.uint8)
        status = np.zeros(nb.size, np.int32)
        held = _i32(sizes)
        rc = load().cimg_decompress_batch_host_sized(self.handle, nb.size, _ptr(comp), _ptr(comp_off), _ptr(held), _ptr(raw), _ptr(raw_off),
                                                     _ptr(nb), _ptr(status))
        if check:
            self._check(rc)
        return [raw[o:o + n] for o, n in zip(raw_off, nb)], status

    # ---- timing ----
    def enable_timing(self, on=True):
        """on: False/0 off, True/1 every batch call, n > 1 every n-th batch call."""
        load().cimg_engine_enable_timing(self.handle, int(on))

    def reset_timing(self):
        load().cimg_engine_reset_timing(self.handle)

    def debug_stamps(self, on=True):
        load().cimg_engine_debug_stamps(self.handle, 1 if on else 0)

    def read_stamps(self, which, max_workgroups=1 << 20):
        out = np.zeros((max_workgroups, 16), np.uint64)
        n = self._check(load().cimg_engine_read_stamps(self.handle, which, _ptr(out), max_workgroups))
        return out[:n]

    def kernel_time(self, kernel):
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(load().cimg_engine_kernel_time(self.handle, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_samples(self, kernel, max_samples=65536):
        """milliseconds of every timed launch of `kernel` since reset_timing(), oldest first"""
        out = np.zeros(max_samples, np.float32)
        n = load().cimg_engine_kernel_samples(self.handle, kernel, _ptr(out), max_samples)
        if n < 0:
            self._check(n)
        return out[:n].copy()

    def decode_stats(self):
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        load().cimg_engine_decode_stats(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"lean_batches": a.value, "blocks_left_to_general": b.value, "blocks_total": c.value, "zstd_batches": d.value}

    def zstd_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        load().cimg_engine_zstd_stats(self.handle, C.byref(a), C.byref(b))
        return {"zstd_batches": a.value, "blocks_refused": b.value}


def cbuffer_sizes(chunk):
    c = np.frombuffer(bytes(chunk[:32]), np.uint8)
    a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
    rc = load().blosc2_cbuffer_sizes(_ptr(c), C.byref(a), C.byref(b), C.byref(d))
    if rc < 0:
        raise CodecError(rc, load().print_error(rc).decode())
    return a.value, b.value, d.value
