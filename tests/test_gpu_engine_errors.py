"""Error paths of the batch engine (csrc/engine.hip) on the GPU: what a rejected batch, an unreadable filter pipeline or a
truncated device-resident chunk must NOT do -- finish on another batch's state, come back with status 0, read past a buffer.

Reference behaviour being mirrored: blosc2_decompress_ctx is called with srcsize = INT32_MAX (blosc2/wrapper.h:249), so the
callee has to be the careful one; a codec / filter the library cannot read is BLOSC2_ERROR_CODEC_SUPPORT (-7), never pixels.
"""
import struct

import numpy as np
import pytest

import _oracle as O
from cimg import hip, synth

pytestmark = pytest.mark.gpu


def _lz4_chunk(eng, a):
    (c,) = eng.compress_host(hip.cparams(a.dtype.itemsize), a, [a.nbytes], [a.nbytes + 32])
    return c


def test_unsupported_compress_requests_on_a_fresh_engine_each():
    """compress_finish ran on a rejected batch and read a result area that was never reserved (null on a fresh engine)."""
    raw = synth.natural_channel(np.uint16, 256, 64)
    for p in (hip.cparams(2, compcode=3), hip.cparams(2, compcode=hip.ZLIB), hip.cparams(2, blocksize=0),
              hip.cparams(2, filters=(0, 0, 0, 0, hip.SHUFFLE, hip.BITSHUFFLE))):
        e = hip.Engine(0)
        with pytest.raises(hip.CodecError) as ei:
            e.compress_host(p, raw, [raw.nbytes], [raw.nbytes + 32])
        assert ei.value.code in (-7, -12), ei.value.code
        # the engine is still usable
        (c,) = e.compress_host(hip.cparams(2), raw, [raw.nbytes], [raw.nbytes + 32])
        assert c == O.compress(O.cparams(2), raw.view(np.uint8).ravel(), destsize=raw.nbytes + 32)[1]
        e.close()
    # many chunks (the multi-group pipeline) on a fresh engine, rejected
    big = synth.natural_channel(np.uint16, 2048, 6144)
    sizes = [4 << 20] * 6
    e = hip.Engine(0)
    with pytest.raises(hip.CodecError):
        e.compress_host(hip.cparams(2, compcode=hip.ZLIB), big, sizes, [s + 32 for s in sizes])
    e.close()


def test_blosc2_shim_rejects_unsupported_codec_on_first_use():
    """The single-chunk shim on the process-wide engine: a filter pipeline the planner refuses must come back as an error code."""
    L = hip.load()
    cp = hip.Blosc2CParams()
    cp.compcode, cp.clevel, cp.typesize, cp.nthreads, cp.blocksize, cp.splitmode = 1, 9, 2, 1, 32768, 3
    cp.filters[4] = 1
    cp.filters[5] = 2                                       # two filters: not on the GPU path
    cctx = L.blosc2_create_cctx(cp)
    if not cctx:
        return                                              # refused at context creation: equally loud
    src = synth.natural_channel(np.uint16, 256, 64).view(np.uint8).ravel()
    dst = np.zeros(src.size + 32, np.uint8)
    assert L.blosc2_compress_ctx(cctx, src.ctypes.data, src.size, dst.ctypes.data, dst.size) < 0
    L.blosc2_free_ctx(cctx)


def test_rejected_decode_batch_does_not_finish_on_the_previous_batch():
    """a planner rejection (header with nbytes < 0, blocks too large for LDS) used to run decompress_finish on the
    PREVIOUS batch's flight state: status words of the old chunk count written into the new, shorter vector."""
    e = hip.Engine(0)
    a = synth.tiled_channel(np.float16, 1024, 512)
    many = [_lz4_chunk(e, a[64 * k:64 * (k + 1)]) for k in range(8)]
    outs, status = e.decompress_host(many)                          # a batch of 8 goes first
    assert not status.any()
    good = many[0]
    # (1) nbytes < 0 in the header
    bad = bytearray(good)
    struct.pack_into("<i", bad, 4, -5)
    with pytest.raises(hip.CodecError) as ei:
        e.decompress_host([bytes(bad)])
    assert ei.value.code == -11
    # (2) a block size no workgroup's LDS holds: planner says CODEC_SUPPORT -- before anything is enqueued
    huge = bytearray(good)
    struct.pack_into("<i", huge, 4, 1 << 20)
    struct.pack_into("<i", huge, 8, 1 << 19)
    raw = np.zeros(1 << 20, np.uint8)
    comp = np.frombuffer(bytes(huge), np.uint8)
    st = np.zeros(1, np.int32)
    rc = hip.load().cimg_decompress_batch_host_sized(e.handle, 1, hip._ptr(comp), hip._ptr(hip._i64([0])), hip._ptr(hip._i32([comp.size])),
                                                     hip._ptr(raw), hip._ptr(hip._i64([0])), hip._ptr(hip._i32([raw.size])), hip._ptr(st))
    assert rc == -7
    # the engine still decodes, and the lean statistics were not double-counted into a crash
    outs, status = e.decompress_host(many[:3])
    assert not status.any() and outs[2].tobytes() == a[128:192].tobytes()
    e.close()


def test_own_codec_with_a_filter_nobody_reads_stays_an_error():
    """the zstd retry cleared EVERY ERR_CODEC_SUPPORT word; an lz4 chunk whose header names a delta filter then ended
    with status 0 and undecoded pixels.  Only zstd chunks are retried now."""
    e = hip.Engine(0)
    a = synth.tiled_channel(np.float16, 512, 64)
    good = _lz4_chunk(e, a)
    for slot, code in ((16 + 4, 3), (16 + 5, 3), (16 + 5, 4), (16 + 3, 1)):     # delta / trunc_prec / a second filter
        bad = bytearray(good)
        bad[slot] = code
        outs, status = e.decompress_host([bytes(bad), good], check=False)
        assert status[0] == -7, (slot, code, status)
        assert status[1] == 0 and outs[1].tobytes() == a.tobytes()
    # a zlib chunk (codec format 3): nobody reads it
    bad = bytearray(good)
    bad[2] = (bad[2] & 0x1F) | (3 << 5)
    outs, status = e.decompress_host([bytes(bad)], check=False)
    assert status[0] == -7
    e.close()


def test_truncated_device_resident_chunk_is_refused():
    """cimg_decompress_batch_device_sized: comp_size[] closes the device-resident hole -- a header that claims more bytes than the
    device buffer holds is READ_BUFFER for that chunk, its neighbours decode."""
    e = hip.Engine(0)
    a = synth.tiled_channel(np.float16, 2048, 1024)                 # one 4 MiB chunk
    good = _lz4_chunk(e, a)
    n = len(good)
    d_comp = e.alloc(2 * n + 256)
    d_comp.upload(np.frombuffer(good, np.uint8), 0)
    off2 = (n + 63) & ~63
    d_comp.upload(np.frombuffer(good, np.uint8), off2)
    d_raw = e.alloc(2 * a.nbytes)
    nb, _, bs = hip.cbuffer_sizes(good)
    # both chunks as they are
    st = e.decompress_device(d_comp.ptr, [0, off2], [nb, nb], [bs, bs], d_raw.ptr, [0, a.nbytes], comp_size=[n, n])
    assert not st.any()
    assert d_raw.download(a.nbytes, a.nbytes).tobytes() == a.tobytes()
    # the first one "truncated": the caller's buffer holds fewer bytes than the header says
    for held in (n - 1, n // 2, 600, 32):
        st = e.decompress_device(d_comp.ptr, [0, off2], [nb, nb], [bs, bs], d_raw.ptr, [0, a.nbytes], check=False, comp_size=[held, n])
        assert st[0] == -5 and st[1] == 0, (held, st)
    with pytest.raises(hip.CodecError):                             # fewer than a header's worth of bytes: refused by the planner
        e.decompress_device(d_comp.ptr, [0], [nb], [bs], d_raw.ptr, [0], comp_size=[31])
    # two-step form
    L = hip.load()
    args = [hip._i64([0, off2]), hip._i32([n - 1, n]), hip._i32([nb, nb]), hip._i32([bs, bs]), hip._i64([0, a.nbytes])]
    rc = L.cimg_decompress_batch_device_begin_sized(e.handle, 2, d_comp.ptr, hip._ptr(args[0]), hip._ptr(args[1]), hip._ptr(args[2]),
                                                    hip._ptr(args[3]), d_raw.ptr, hip._ptr(args[4]))
    assert rc == 0
    st = e.decompress_device_fetch(2, check=False)
    assert st[0] == -5 and st[1] == 0
    e.close()
