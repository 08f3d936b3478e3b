"""Parity tests proper: the HIP kernels, called through the C ABI (libcimg_hip.so), against the oracle.

Bit-exact compressed bytes AND bit-exact pixels on the same seeded inputs, the liblz4 golden vectors
pushed through the GPU encoder, the reference's own known-answer cases, edge cases, and the BASELINE
configs[1] geometry at full size.  Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

import _oracle as O
from cimg import hip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = hip.Engine(0)
    yield e
    e.close()


def _roundtrip(eng, dtype, arr, chunk, blocksize=32768, destsize=None, clevel=9, filters=(0, 0, 0, 0, 0, 1), splitmode=3, compcode=1):
    it = np.dtype(dtype).itemsize
    raw = np.ascontiguousarray(arr).view(np.uint8).ravel()
    sizes = [min(chunk, raw.size - o) for o in range(0, raw.size, chunk)]
    dsz = [chunk + 32 if destsize is None else destsize] * len(sizes)
    chunks = eng.compress_host(hip.cparams(it, clevel=clevel, blocksize=blocksize, filters=filters, splitmode=splitmode, compcode=compcode), raw, sizes, dsz)
    po = O.cparams(it, clevel=clevel, blocksize=blocksize, filters=filters, splitmode=splitmode, compcode=compcode)
    off = 0
    for i, s in enumerate(sizes):
        r, want = O.compress(po, raw[off:off + s], destsize=dsz[i])
        assert len(chunks[i]) == r, (np.dtype(dtype).name, i, len(chunks[i]), r)
        if chunks[i] != want:
            k = next(j for j, (a, b) in enumerate(zip(chunks[i], want)) if a != b)
            raise AssertionError(f"{np.dtype(dtype).name} chunk {i}: first differing byte at {k} of {r}")
        off += s
    live = [c for c in chunks if c]
    if live:
        outs, status = eng.decompress_host(live)
        assert not status.any()
        want = b"".join(raw[sum(sizes[:i]):sum(sizes[:i + 1])].tobytes() for i, c in enumerate(chunks) if c)
        assert b"".join(o.tobytes() for o in outs) == want
    return chunks


def _oracle_all_chunks(po, host, chunk, destsize=None):
    """Every chunk of `host` through the oracle, C-driven on the box's cores (oracle/bench_cpu.c): (cbytes[], bytes of chunk i)."""
    L = O.lib()
    L.orc_bench_compress.argtypes = [C.POINTER(O.CParams), C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_int]
    L.orc_bench_compress.restype = C.c_int64
    n = host.size // chunk
    stride = chunk + 64
    comp = np.zeros(n * stride, np.uint8)
    cb = np.zeros(n, np.int32)
    threads = min(len(os.sched_getaffinity(0)), 16, n)
    r = L.orc_bench_compress(C.byref(po), host.ctypes.data, n, chunk, comp.ctypes.data, stride, chunk + 32 if destsize is None else destsize,
                             cb.ctypes.data, threads, 1)
    assert r > 0
    return cb, (lambda i: comp[i * stride:i * stride + cb[i]])


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float16, np.uint32, np.float32])
@pytest.mark.parametrize("family", ["tiled", "zero", "random", "natural"])
def test_bytes_and_pixels_equal_oracle(eng, dtype, family):
    arr = getattr(synth, family + "_channel")(dtype, 1024, 300)
    it = np.dtype(dtype).itemsize
    _roundtrip(eng, dtype, arr, 4 * 1024 * 1024 // (1024 * it) * 1024 * it)      # scanline-aligned 4 MiB chunks
    _roundtrip(eng, dtype, arr[:100], 40000 // it * it)                          # ragged: leftover blocks, short last chunk


def test_liblz4_golden_vectors_through_the_gpu_encoder(eng, golden_dir):
    """typesize 1, no filter, block = whole chunk: the chunk's single stream IS the LZ4 call."""
    kat = np.load(os.path.join(golden_dir, "lz4_kat.npz"))
    names = sorted({str(k).split("|")[0] for k in kat["cases"]})
    names = [n for n in names if 32 <= kat["in|" + n].size <= 65536]
    for clevel, accel in ((9, 1), (5, 5)):
        raw = np.concatenate([kat["in|" + n] for n in names])
        sizes = [int(kat["in|" + n].size) for n in names]
        p = hip.cparams(1, clevel=clevel, blocksize=65536, filters=(0, 0, 0, 0, 0, 0), splitmode=2)
        chunks = eng.compress_host(p, raw, sizes, [s + 4096 for s in sizes])
        checked = 0
        for n, c, s in zip(names, chunks, sizes):
            src = kat["in|" + n]
            if (src == src[0]).all():
                continue                                            # run token, LZ4 not called
            key = f"{n}|a{accel}|c{s}"
            ret = int(kat["ret|" + key])
            (cs,) = struct.unpack_from("<i", c, 32 + 4)
            if ret == 0 or ret == s:
                assert cs == s and c[40:40 + s] == src.tobytes(), key
            else:
                assert cs == ret and c[40:40 + cs] == kat["out|" + key].tobytes(), key
            checked += 1
        assert checked > 50
        outs, status = eng.decompress_host(chunks)
        assert not status.any() and b"".join(o.tobytes() for o in outs) == raw.tobytes()


def test_reference_known_answers(eng):
    for dt in (np.uint8, np.uint16, np.uint32, np.float32):                     # test_schunk.cpp:39-75
        chunks = _roundtrip(eng, dt, np.arange(4096).astype(dt), 256, blocksize=64)
        assert len(chunks) == 4096 * np.dtype(dt).itemsize // 256
        assert hip.cbuffer_sizes(chunks[0])[0] == 256
    _roundtrip(eng, np.uint8, np.arange(50, dtype=np.uint8), 4194304)            # test_channel.cpp:46-55
    _roundtrip(eng, np.uint8, (np.arange(8192) & 255).astype(np.uint8), 4096, blocksize=128)   # :60-69
    for v in (255, 0, 199, 12, 13, 14, 25, 90, 100):                             # test_image.cpp / python tests
        for dt in (np.uint8, np.uint16, np.float32, np.float16):
            _roundtrip(eng, dt, np.full(64 * 16, v).astype(dt), 768 // np.dtype(dt).itemsize // 64 * 64 * np.dtype(dt).itemsize or 64 * np.dtype(dt).itemsize, blocksize=256)


def test_edge_geometries(eng):
    rng = np.random.default_rng(3)
    a = (rng.integers(0, 40, 6000, dtype=np.uint64) * 0x0101010101).astype(np.uint64)
    _roundtrip(eng, np.uint64, a, 16384, blocksize=4096)                         # typesize 8: generic plane path
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 300, 41), 5000, blocksize=256)
    _roundtrip(eng, np.uint8, synth.natural_channel(np.uint8, 333, 77), 9999, blocksize=1000)
    _roundtrip(eng, np.uint8, np.arange(20, dtype=np.uint8), 4096)               # < 32 bytes -> memcpyed chunk
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 512, 64), 65536, clevel=0)   # clevel 0 -> memcpyed
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 512, 64), 65536, clevel=5)   # LZ4 acceleration 5
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 512, 256), 262144, blocksize=65536)
    noisy = rng.integers(0, 65536, 20000, dtype=np.uint16)
    noisy[3000:9000] = 7
    for destsize in (40000 + 32, 39000, 36000, 30000, 20100, 200):               # blosc2's running-destsize rule
        _roundtrip(eng, np.uint16, noisy, 40000, blocksize=4096, destsize=destsize)


def test_empty_batch_and_unsupported_requests_fail_loudly(eng):
    assert eng.compress_device(hip.cparams(2), 0, [], [], 0, [], []).size == 0
    raw = synth.natural_channel(np.uint16, 256, 64)
    with pytest.raises(hip.CodecError) as ei:                                 # zlib: the one blosc2 codec nobody here writes
        eng.compress_host(hip.cparams(2, compcode=hip.ZLIB), raw, [raw.nbytes], [raw.nbytes + 32])
    assert ei.value.code == -7
    with pytest.raises(hip.CodecError):                                       # a second filter in the pipeline
        eng.compress_host(hip.cparams(2, filters=(0, 0, 0, 0, hip.SHUFFLE, hip.BITSHUFFLE)), raw, [raw.nbytes], [raw.nbytes + 32])


def test_corrupt_chunks_are_reported_not_crashed(eng):
    a = synth.natural_channel(np.uint16, 1024, 64)
    (good,) = eng.compress_host(hip.cparams(2), a, [a.nbytes], [a.nbytes + 32])
    bad = bytearray(good)
    struct.pack_into("<i", bad, 32, len(good) + 1000)            # bstart outside the chunk
    outs, status = eng.decompress_host([bytes(bad), good], check=False)
    assert status[0] < 0 and status[1] == 0
    assert outs[1].tobytes() == a.tobytes()
    bad = bytearray(good)
    first = struct.unpack_from("<i", good, 32)[0]
    for k in range(first + 4, first + 40):
        bad[k] ^= 0x5A                                            # garbage inside an LZ4 stream
    outs, status = eng.decompress_host([bytes(bad)], check=False)
    assert status[0] < 0 or outs[0].tobytes() != a.tobytes()


def test_blosc2_shim_single_chunk_calls(eng):
    L = hip.load()
    cp = hip.Blosc2CParams()
    cp.compcode, cp.clevel, cp.typesize, cp.nthreads, cp.blocksize, cp.splitmode = 1, 9, 2, 4, 32768, 3
    cp.filters[5] = 1
    cctx = L.blosc2_create_cctx(cp)
    dp = hip.Blosc2DParams()
    dp.nthreads = 1
    dctx = L.blosc2_create_dctx(dp)
    a = synth.tiled_channel(np.float16, 2048, 100)
    src = a.view(np.uint8).ravel()
    dst = np.zeros(src.size + 32, np.uint8)
    r = L.blosc2_compress_ctx(cctx, src.ctypes.data, src.size, dst.ctypes.data, dst.size)
    ro, want = O.compress(O.cparams(2), src, destsize=src.size + 32)
    assert r == ro and dst[:r].tobytes() == want
    out = np.zeros(src.size, np.uint8)
    assert L.blosc2_decompress_ctx(dctx, dst.ctypes.data, 2**31 - 1, out.ctypes.data, out.size) == src.size
    assert out.tobytes() == src.tobytes()
    assert L.blosc2_decompress_ctx(dctx, dst.ctypes.data, 2**31 - 1, out.ctypes.data, out.size - 1) == -6
    L.blosc2_free_ctx(cctx)
    L.blosc2_free_ctx(dctx)


def test_full_size_config2_device_resident(eng):
    """BASELINE configs[1]: 4 x 4096^2 float16, 32 chunks of 4 MiB, device-resident batch calls."""
    chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    n, chunk = host.size, 4 * 1024 * 1024
    nchunks, stride = n // chunk, chunk + 64
    d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
    d_raw.upload(host)
    raw_off = np.arange(nchunks) * chunk
    comp_off = np.arange(nchunks) * stride
    cbytes = eng.compress_device(hip.cparams(2), d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    comp = d_comp.download()
    po = O.cparams(2)
    for i in range(nchunks):
        r, want = O.compress(po, host[i * chunk:(i + 1) * chunk], destsize=chunk + 32)
        assert cbytes[i] == r and comp[comp_off[i]:comp_off[i] + r].tobytes() == want, i
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    assert d_out.download().tobytes() == host.tobytes()
    # idempotence: a second pass over the same buffers gives the same bytes
    cbytes2 = eng.compress_device(hip.cparams(2), d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    assert (cbytes2 == cbytes).all() and d_comp.download().tobytes() == comp.tobytes()
    for b in (d_raw, d_out, d_comp):
        b.free()


def test_descriptor_cache_follows_the_geometry(eng):
    """Batches alternate between two geometries and between fresh data in the same geometry: the engine's cached
    chunk descriptors must never be used for a batch they do not describe."""
    a = synth.natural_channel(np.uint16, 512, 256)                 # 256 KiB
    b = synth.tiled_channel(np.uint16, 512, 256, c=2)
    for arr, chunk in ((a, 65536), (a, 32768), (b, 65536), (a, 65536), (b, 32768), (b, 32768)):
        _roundtrip(eng, np.uint16, arr, chunk)
    # same sizes, different device offsets
    raw = a.view(np.uint8).ravel()
    n, chunk = raw.size, 65536
    d_raw, d_comp, d_out = eng.alloc(2 * n), eng.alloc(2 * (n + 4096)), eng.alloc(2 * n)
    d_raw.upload(raw)
    d_raw.upload(b.view(np.uint8).ravel(), offset=n)
    po = O.cparams(2)
    for base, src in ((0, raw), (n, b.view(np.uint8).ravel()), (0, raw)):
        raw_off = base + np.arange(n // chunk) * chunk
        comp_off = (base // chunk) * (chunk + 64) + np.arange(n // chunk) * (chunk + 64)
        cb = eng.compress_device(hip.cparams(2), d_raw.ptr, raw_off, [chunk] * len(raw_off), d_comp.ptr, comp_off, [chunk + 32] * len(raw_off))
        comp = d_comp.download()
        for i in range(len(raw_off)):
            r, want = O.compress(po, src[i * chunk:(i + 1) * chunk], destsize=chunk + 32)
            assert cb[i] == r and comp[comp_off[i]:comp_off[i] + r].tobytes() == want
        eng.decompress_device(d_comp.ptr, comp_off, [chunk] * len(raw_off), [32768] * len(raw_off), d_out.ptr, raw_off)
        assert d_out.download(n, offset=base).tobytes() == src.tobytes()
    for buf in (d_raw, d_comp, d_out):
        buf.free()


def test_config1_geometry_remainder_chunk_in_a_nominal_buffer(eng):
    """BASELINE configs[0] geometry with the GPU codec (lz4): one 1024^2 uint8 channel is a single 1 MiB remainder
    chunk compressed into the nominal 4 MiB + 32 buffer (schunk.h:73), so incompressible data stays block-framed
    and cbytes > nbytes (SURVEY.md N7 ii)."""
    rng = np.random.default_rng(11)
    for arr in (synth.tiled_channel(np.uint8, 1024, 1024), rng.integers(0, 256, 1024 * 1024, dtype=np.uint8)):
        (c,) = _roundtrip(eng, np.uint8, arr, 1024 * 1024, destsize=4 * 1024 * 1024 + 32)
        assert struct.unpack_from("<i", c, 4)[0] == 1024 * 1024
    assert len(c) > 1024 * 1024 + 32 and not (c[2] & 0x02)        # random bytes: framed, not memcpyed


def test_config3_geometry_random_access_get_set_chunk(eng):
    """BASELINE configs[2] access pattern (with the GPU codec, lz4 + byte shuffle): 8192^2 uint16 channels as 32
    chunks of 4 MiB, chunks visited in a seeded random permutation: decode one, add 1, encode it back."""
    rng = np.random.default_rng(99)
    chan = synth.tiled_channel(np.uint16, 8192, 8192, c=1)
    host = chan.view(np.uint8).ravel()
    n, chunk = host.size, 4 * 1024 * 1024
    nchunks, stride = n // chunk, chunk + 64
    d_raw, d_comp, d_one = eng.alloc(n), eng.alloc(nchunks * stride), eng.alloc(chunk)
    d_raw.upload(host)
    raw_off, comp_off = np.arange(nchunks) * chunk, np.arange(nchunks) * stride
    p = hip.cparams(2)
    cbytes = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    po = O.cparams(2)
    order = rng.permutation(nchunks)
    for i in order:
        eng.decompress_device(d_comp.ptr + int(comp_off[i]), [0], [chunk], [32768], d_one.ptr, [0])       # get_chunk
        px = d_one.download().view(np.uint16) + np.uint16(1)
        d_one.upload(px)
        cbytes[i] = eng.compress_device(p, d_one.ptr, [0], [chunk], d_comp.ptr + int(comp_off[i]), [0], [chunk + 32])[0]   # set_chunk
    comp = d_comp.download()
    want_px = chan.ravel() + np.uint16(1)
    ocb, ochunk = _oracle_all_chunks(po, np.ascontiguousarray(want_px.view(np.uint8)), chunk)
    for i in range(nchunks):                                       # bytes against the oracle on EVERY chunk (round 3: was a sample of 6) ...
        assert cbytes[i] == ocb[i] and comp[comp_off[i]:comp_off[i] + cbytes[i]].tobytes() == ochunk(i).tobytes(), i
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_raw.ptr, raw_off)
    assert d_raw.download().tobytes() == want_px.tobytes()         # ... pixels everywhere
    for buf in (d_raw, d_comp, d_one):
        buf.free()


def test_config4_one_rank_share_many_images(eng):
    """BASELINE configs[3], the share of one of 8 ranks: 8 images x 4 channels x 4096^2 float16 = 256 chunks (1 GiB)
    in ONE batch call.  Round trip everywhere, bytes against the oracle on every chunk."""
    imgs, chunk = 8, 4 * 1024 * 1024
    host = np.concatenate([synth.tiled_channel(np.float16, 4096, 4096, c=c, seed=1234 + 4 * img).view(np.uint8).ravel()
                           for img in range(imgs) for c in range(4)])
    n = host.size
    nchunks, stride = n // chunk, chunk + 64
    d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
    d_raw.upload(host)
    raw_off, comp_off = np.arange(nchunks) * chunk, np.arange(nchunks) * stride
    cbytes = eng.compress_device(hip.cparams(2), d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    assert (cbytes > 32).all() and (cbytes < chunk).all()
    ocb, ochunk = _oracle_all_chunks(O.cparams(2), host, chunk)
    comp = d_comp.download()
    for i in range(nchunks):                                       # all 256 chunks byte for byte (round 3: was a sample of 6)
        assert cbytes[i] == ocb[i] and comp[comp_off[i]:comp_off[i] + cbytes[i]].tobytes() == ochunk(i).tobytes(), i
    del comp
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    assert d_out.download().tobytes() == host.tobytes()
    for buf in (d_raw, d_out, d_comp):
        buf.free()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float16, np.uint32, np.float32, np.uint64])
def test_bitshuffle_filter(eng, dtype):
    """lz4 + bitshuffle (BASELINE configs[2]'s filter with the GPU codec): bytes and pixels against the oracle."""
    bs = (0, 0, 0, 0, 0, hip.BITSHUFFLE)
    it = np.dtype(dtype).itemsize
    arr = synth.natural_channel(dtype, 2048, 128) if it <= 4 else (np.arange(2048 * 128, dtype=np.uint64) // 7 * 0x0101).astype(np.uint64)
    _roundtrip(eng, dtype, arr, 262144, filters=bs)                          # 8 blocks of 32 KiB per chunk
    _roundtrip(eng, dtype, arr, 40960, blocksize=8192, filters=bs)
    _roundtrip(eng, dtype, arr.ravel()[:5003], 4096 * it, blocksize=1024 * it, filters=bs)   # ne % 8 != 0 in the leftover block


def test_config3_with_bitshuffle_full_size(eng):
    """One 8192^2 uint16 channel (32 chunks of 4 MiB), lz4 + bitshuffle, device-resident; oracle bytes on every chunk."""
    chan = synth.tiled_channel(np.uint16, 8192, 8192, c=2)
    host = chan.view(np.uint8).ravel()
    n, chunk = host.size, 4 * 1024 * 1024
    nchunks, stride = n // chunk, chunk + 64
    d_raw, d_out, d_comp = eng.alloc(n), eng.alloc(n), eng.alloc(nchunks * stride)
    d_raw.upload(host)
    raw_off, comp_off = np.arange(nchunks) * chunk, np.arange(nchunks) * stride
    p = hip.cparams(2, filters=(0, 0, 0, 0, 0, hip.BITSHUFFLE))
    cbytes = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    po = O.cparams(2, filters=(0, 0, 0, 0, 0, O.BITSHUFFLE))
    ocb, ochunk = _oracle_all_chunks(po, host, chunk)
    comp = d_comp.download()
    for i in range(nchunks):                                       # every chunk (round 3: was three)
        assert cbytes[i] == ocb[i] and comp[comp_off[i]:comp_off[i] + cbytes[i]].tobytes() == ochunk(i).tobytes(), i
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
    assert d_out.download().tobytes() == host.tobytes()
    for buf in (d_raw, d_out, d_comp):
        buf.free()


def test_split_modes(eng):
    a = synth.natural_channel(np.uint16, 2048, 64)
    f = synth.tiled_channel(np.float32, 1024, 64)
    _roundtrip(eng, np.uint16, a, 131072, splitmode=2)                                # never split: shuffled blocks as one stream
    _roundtrip(eng, np.float32, f, 65536, blocksize=8192, splitmode=2)
    _roundtrip(eng, np.uint16, a, 40000, splitmode=1, filters=(0, 0, 0, 0, 0, 0))     # always split, no filter
    _roundtrip(eng, np.uint16, a, 131072, splitmode=4)
    _roundtrip(eng, np.uint8, synth.natural_channel(np.uint8, 1024, 64), 20000, splitmode=1)


def test_general_decode_kernel_alone_still_covers_every_block():
    """The lean decode launch takes most image blocks in the other tests; with it switched off (CIMG_NO_LEAN, read
    when an engine is created) the general kernel must produce the same pixels for the same chunks."""
    os.environ["CIMG_NO_LEAN"] = "1"
    try:
        e2 = hip.Engine(0)
    finally:
        del os.environ["CIMG_NO_LEAN"]
    try:
        for dtype, arr in ((np.float16, synth.tiled_channel(np.float16, 2048, 256)), (np.float32, synth.tiled_channel(np.float32, 1024, 128)),
                           (np.uint16, synth.natural_channel(np.uint16, 1024, 100)), (np.uint8, synth.natural_channel(np.uint8, 1024, 64))):
            _roundtrip(e2, dtype, arr, 262144)
            _roundtrip(e2, dtype, arr, 40000 // np.dtype(dtype).itemsize * np.dtype(dtype).itemsize)
    finally:
        e2.close()


def test_assembly_inside_the_encode_launch_and_behind_it_give_the_same_chunks(eng):
    """Round 3: chunks whose streams all belong to one encode launch are laid out and copied into place by the waves of that launch
    (the wave that finishes a chunk's last stream lays it out; every wave copies the streams it encoded); memcpyed-up-front chunks
    and chunks with split planes PLUS an unsplit leftover block go through cimg_layout_chunks / cimg_emit_blocks behind it.
    CIMG_NO_ASSEMBLE_IN_LAUNCH=1 (read when an engine is created) sends everything the second way: same bytes, and both equal
    the oracle's.  Run several times in a row: the queue heads and the per-chunk generation count on from batch to batch."""
    os.environ["CIMG_NO_ASSEMBLE_IN_LAUNCH"] = "1"
    try:
        e2 = hip.Engine(0)
    finally:
        del os.environ["CIMG_NO_ASSEMBLE_IN_LAUNCH"]
    try:
        a = synth.tiled_channel(np.float16, 2048, 700)                       # 2.8 MB
        raw = a.view(np.uint8).ravel()
        for sizes, dest in (([262144] * 10 + [raw.size - 10 * 262144], 262144 + 32),      # whole blocks everywhere (last chunk: 245760 B... with a leftover?)
                            ([100000] * 20, 100000 + 32),                                 # every chunk: split planes + an unsplit leftover block
                            ([262144, 20, 262144, 31, 500000], 500000 + 32)):             # tiny chunks are memcpyed up front
            src = np.concatenate([raw[:s] for s in sizes])
            for rep in range(3):
                got = eng.compress_host(hip.cparams(2), src, sizes, [dest] * len(sizes))
                ref = e2.compress_host(hip.cparams(2), src, sizes, [dest] * len(sizes))
                assert got == ref, (sizes[:3], rep)
            off = 0
            for c, n in zip(got, sizes):
                assert c == O.compress(O.cparams(2), src[off:off + n], destsize=dest)[1]
                off += n
            outs, status = eng.decompress_host(got)
            assert not status.any() and b"".join(o.tobytes() for o in outs) == src.tobytes()
        # blosclz goes through the same launch shape
        sizes = [131072] * 8
        src = raw[:sum(sizes)]
        got = eng.compress_host(hip.cparams(2, compcode=hip.BLOSCLZ), src, sizes, [131072 + 32] * 8)
        assert got == e2.compress_host(hip.cparams(2, compcode=hip.BLOSCLZ), src, sizes, [131072 + 32] * 8)
    finally:
        e2.close()


@pytest.mark.parametrize("force", ["1", "0"])
def test_encode_block_items_and_plane_items_give_the_same_bytes(force):
    """The split encode launch hands out whole blocks (each read from HBM once, waiting planes in registers) on large
    batches, and on small ones for the rounds every chain takes anyway with the last partial round plane by plane
    (CIMG_ENC_BLOCK_ITEMS forces all / none, read when an engine is created; the mixed queue runs in
    test_encode_automatic_gang_on_a_batch_that_fills_the_device and in the emulator, tests/test_emu_kernels.py):
    both must produce the oracle's bytes, for lz4 and blosclz, 2- and 4-byte types, with leftover blocks in the batch."""
    os.environ["CIMG_ENC_BLOCK_ITEMS"] = force
    try:
        e2 = hip.Engine(0)
    finally:
        del os.environ["CIMG_ENC_BLOCK_ITEMS"]
    try:
        for dtype, arr in ((np.float16, synth.tiled_channel(np.float16, 2048, 256)), (np.float32, synth.tiled_channel(np.float32, 1024, 128)),
                           (np.uint16, synth.natural_channel(np.uint16, 1024, 100)), (np.float32, synth.natural_channel(np.float32, 1024, 67))):
            for code in (hip.LZ4, hip.BLOSCLZ):
                _roundtrip(e2, dtype, arr, 262144, compcode=code)
                _roundtrip(e2, dtype, arr, 100000 // np.dtype(dtype).itemsize * np.dtype(dtype).itemsize, compcode=code)   # leftover blocks
    finally:
        e2.close()


@pytest.mark.parametrize("gang", ["1", "3", "5", "8"])
def test_encode_gangs_give_the_same_bytes(gang):
    """The encode launch packs independent chains into workgroups of 1..8 waves so that the CU's LDS granules come out even
    (engine.hip: encode_gang; five 32 KiB chains only fit a CU as ONE 160 KiB workgroup).  CIMG_ENC_GANG forces the size,
    read when an engine is created; a batch large enough to take the automatic choice runs as well.  Same bytes as the
    oracle every way, for lz4 and blosclz, split and unsplit blocks, with a leftover block in the batch."""
    os.environ["CIMG_ENC_GANG"] = gang
    try:
        e2 = hip.Engine(0)
    finally:
        del os.environ["CIMG_ENC_GANG"]
    try:
        for dtype, arr in ((np.float16, synth.tiled_channel(np.float16, 2048, 256)), (np.float32, synth.natural_channel(np.float32, 1024, 67)),
                           (np.uint8, synth.tiled_channel(np.uint8, 1024, 200))):
            for code in (hip.LZ4, hip.BLOSCLZ):
                _roundtrip(e2, dtype, arr, 262144, compcode=code)
                _roundtrip(e2, dtype, arr, 100000 // np.dtype(dtype).itemsize * np.dtype(dtype).itemsize, compcode=code)
    finally:
        e2.close()


def test_encode_automatic_gang_on_a_batch_that_fills_the_device(eng):
    """1536 blocks x 2 planes is more work items than single-wave workgroups fit the device (4 per CU at 32 KiB): the engine
    switches to gangs of five on its own.  Oracle bytes on a sample of the chunks, pixels on all of them."""
    a = synth.tiled_channel(np.float16, 4096, 6144)                     # 48 MiB = 12 chunks of 4 MiB = 1536 blocks
    chunk = 4 * 1024 * 1024
    sizes = [chunk] * (a.nbytes // chunk)
    chunks = eng.compress_host(hip.cparams(2), a, sizes, [chunk + 32] * len(sizes))
    raw = a.view(np.uint8).ravel()
    for k in (0, 5, len(sizes) - 1):
        rc, want = O.compress(O.cparams(2), raw[k * chunk:(k + 1) * chunk])
        assert rc == len(chunks[k]) and want == bytes(chunks[k])
    outs, status = eng.decompress_host(chunks)
    assert not status.any()
    assert np.array_equal(np.concatenate(outs), raw)


def test_device_batches_in_two_steps_overlap_and_refuse_misuse(eng):
    """cimg_*_batch_device_begin / _fetch: a decompress batch is enqueued right behind the compress batch whose chunks it
    reads, before anything of the compress results has reached the host; sizes, bytes and pixels are those of the plain
    calls.  A _fetch without its _begin, or after a plain call of the same kind, is an error, not a hang."""
    a = synth.tiled_channel(np.float16, 2048, 1024)                     # 4 MiB -> 4 chunks of 1 MiB
    raw = a.view(np.uint8).ravel()
    chunk, n = 1 << 20, 4
    stride = chunk + 64
    d_raw, d_comp, d_out = eng.alloc(raw.size), eng.alloc(n * stride), eng.alloc(raw.size)
    d_raw.upload(raw)
    raw_off, comp_off = np.arange(n) * chunk, np.arange(n) * stride
    p = hip.cparams(2)
    plain = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * n, d_comp.ptr, comp_off, [chunk + 32] * n)
    for rep in range(3):
        k = eng.compress_device_begin(p, d_raw.ptr, raw_off, [chunk] * n, d_comp.ptr, comp_off, [chunk + 32] * n)
        eng.decompress_device_begin(d_comp.ptr, comp_off, [chunk] * n, [32768] * n, d_out.ptr, raw_off)
        cb = eng.compress_device_fetch(k)
        st = eng.decompress_device_fetch(k)
        assert np.array_equal(cb, plain) and not st.any()
        assert np.array_equal(d_out.download(), raw)
    comp = d_comp.download()
    for i in range(n):
        rc, want = O.compress(O.cparams(2), raw[i * chunk:(i + 1) * chunk])
        assert rc == cb[i] and want == comp[i * stride:i * stride + rc].tobytes()
    with pytest.raises(hip.CodecError):
        eng.compress_device_fetch(n)                                     # nothing in flight
    with pytest.raises(hip.CodecError):
        eng.decompress_device_fetch(n)
    eng.compress_device_begin(p, d_raw.ptr, raw_off, [chunk] * n, d_comp.ptr, comp_off, [chunk + 32] * n)
    eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * n, d_comp.ptr, comp_off, [chunk + 32] * n)   # a plain call in between
    with pytest.raises(hip.CodecError):
        eng.compress_device_fetch(n)
    with pytest.raises(hip.CodecError):
        eng.compress_device_begin(p, d_raw.ptr, raw_off, [chunk] * n, d_comp.ptr, comp_off, [chunk + 32] * n) and eng.compress_device_fetch(n + 1)
    eng.synchronize()


def test_lean_and_general_decode_agree_on_mixed_batches(eng):
    """One batch holding chunks the lean kernel takes (tiled float16), chunks it must leave (both planes coded, ragged
    leftover blocks, typesize 1) and a damaged chunk: every good chunk decodes, the damaged one is reported."""
    t = synth.tiled_channel(np.float16, 1024, 64).view(np.uint8).ravel()           # 128 KiB
    nat = synth.natural_channel(np.uint16, 1024, 50).view(np.uint8).ravel()         # 100 KiB: ragged
    u8 = synth.natural_channel(np.uint8, 1024, 64).ravel()
    chunks = []
    for ts, raw in ((2, t), (2, nat), (1, u8)):
        chunks += eng.compress_host(hip.cparams(ts), raw, [raw.size], [raw.size + 32])
    bad = bytearray(chunks[0])
    first = struct.unpack_from("<i", chunks[0], 32)[0]
    for k in range(first + 8, first + 60):
        bad[k] ^= 0xA5
    outs, status = eng.decompress_host(chunks + [bytes(bad)], check=False)
    assert list(status[:3]) == [0, 0, 0]
    assert outs[0].tobytes() == t.tobytes() and outs[1].tobytes() == nat.tobytes() and outs[2].tobytes() == u8.tobytes()
    assert status[3] < 0 or outs[3].tobytes() != t.tobytes()
    # lean blocks WITHOUT a coded plane: 16-bit pixels below 256 with a noisy low byte (low plane stored raw, high plane a zero run),
    # decoded right behind blocks that do have one -- so that whatever those left in LDS would show (a round-3 build un-shuffled the
    # stored plane against it); lz4 and blosclz, which store different blocks raw
    rng = np.random.default_rng(11)
    quiet = ((np.arange(300 * 200).reshape(200, 300) // 37 % 251) + rng.integers(0, 3, (200, 300))).astype(np.uint16).view(np.uint8).ravel()
    noisy = rng.integers(0, 256, (512, 1024)).astype(np.uint16).view(np.uint8).ravel()
    for codec in (hip.LZ4, hip.BLOSCLZ):
        cs = []
        for raw in (t, quiet, noisy, t, quiet):
            cs += eng.compress_host(hip.cparams(2, compcode=codec), raw, [raw.size], [raw.size + 32])
        outs, status = eng.decompress_host(cs)
        assert not status.any()
        for o, raw in zip(outs, (t, quiet, noisy, t, quiet)):
            assert o.tobytes() == raw.tobytes(), codec


def test_unusual_typesizes(eng):
    """Element sizes 3 (packed RGB), 16 (largest that still splits) and 20 (one shuffled stream per block) through the C ABI."""
    rng = np.random.default_rng(21)
    for ts in (3, 16, 20):
        n = 40000
        base = (np.arange(n)[:, None] // 9 * (np.arange(ts)[None, :] + 1)).astype(np.uint8)
        base[:, 0] = rng.integers(0, 256, n)
        raw = np.ascontiguousarray(base).ravel()
        for blocksize in (32768 // ts * ts, 4096 // ts * ts):
            (c,) = eng.compress_host(hip.cparams(ts, blocksize=blocksize), raw, [raw.size], [raw.size + 32])
            r, want = O.compress(O.cparams(ts, blocksize=blocksize), raw, destsize=raw.size + 32)
            assert len(c) == r and c == want, (ts, blocksize)
            outs, st = eng.decompress_host([c])
            assert not st.any() and outs[0].tobytes() == raw.tobytes()


def _mixed_data(rng, n, ts):
    """n bytes of ts-byte elements: smooth ramps, flat runs, noise and repeats glued together at random."""
    out = np.empty(n, np.uint8)
    ne = n // ts
    e = np.zeros((ne, ts), np.uint8)
    pos = 0
    while pos < ne:
        ln = int(rng.integers(1, 4000))
        kind = int(rng.integers(0, 5))
        seg = e[pos:pos + ln]
        m = seg.shape[0]
        if kind == 0:
            seg[:] = rng.integers(0, 256, (m, ts), dtype=np.uint8)
        elif kind == 1:
            seg[:] = rng.integers(0, 256, (1, ts), dtype=np.uint8)
        elif kind == 2:
            ramp = (np.arange(m) // int(rng.integers(1, 40))).astype(np.uint32) * int(rng.integers(1, 300))
            for b in range(ts):
                seg[:, b] = (ramp >> (8 * (b % 4))) & 0xFF
        elif kind == 3 and pos > 0:
            back = int(rng.integers(1, min(pos, 5000) + 1))
            src = e[pos - back:pos - back + m]
            seg[:src.shape[0]] = src
        else:
            seg[:, 0] = rng.integers(0, 4, m, dtype=np.uint8)
            seg[:, 1:] = 7
        pos += ln
    out[:ne * ts] = e.ravel()
    out[ne * ts:] = 1
    return out


@pytest.mark.parametrize("compcode", [hip.LZ4, hip.BLOSCLZ])
def test_randomized_geometries_against_the_oracle(eng, compcode):
    """Seeded differential test: random element size, block size, chunk size, clevel, filter, dest capacity and data
    make-up; every chunk must equal the oracle's byte for byte and decode back to the input.  lz4 and blosclz (whose streams end
    at 65535 bytes: a one-byte 64 KiB block is taken as 32 KiB there)."""
    rng = np.random.default_rng(20260101 + int(os.environ.get("CIMG_TEST_SEED", "0")) + 7919 * (compcode != hip.LZ4))       # CIMG_TEST_SEED: soak runs with other seeds
    for it in range(int(os.environ.get("CIMG_TEST_ROUNDS", "400")) // (1 if compcode == hip.LZ4 else 2)):
        ts = int(rng.choice([1, 2, 2, 4, 4, 8, 3]))
        blocksize = int(rng.choice([256, 1024, 4096, 8192, 32768, 65536])) // ts * ts
        if compcode == hip.BLOSCLZ and blocksize > 65535:
            blocksize = 32768
        nchunks = int(rng.integers(1, 5))
        chunk = int(rng.integers(1, 9)) * blocksize + (int(rng.integers(0, blocksize)) // ts * ts if rng.random() < 0.4 else 0)
        chunk = min(chunk, 300000) // ts * ts or ts
        total = chunk * (nchunks - 1) + int(rng.integers(1, chunk + 1)) // ts * ts
        total = max(total, ts)
        raw = _mixed_data(rng, total, ts)
        clevel = int(rng.choice([1, 5, 9, 9, 9]))
        filt = int(rng.choice([0, 1, 1, 1, 2]))
        if filt == 2 and blocksize > 65536 // 1:
            filt = 1
        dest = chunk + 32 if rng.random() < 0.7 else max(40, int(chunk * rng.uniform(0.3, 1.0)))
        sizes = [min(chunk, total - o) for o in range(0, total, chunk)]
        if compcode == hip.BLOSCLZ:
            clevel = int(rng.integers(1, 10))
        p = hip.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
        po = O.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
        chunks = eng.compress_host(p, raw, sizes, [dest] * len(sizes))
        off = 0
        for i, s in enumerate(sizes):
            r, want = O.compress(po, raw[off:off + s], destsize=dest)
            assert len(chunks[i]) == max(r, 0) and chunks[i] == want, (compcode, it, i, ts, blocksize, chunk, clevel, filt, dest, len(chunks[i]), r)
            off += s
        live = [(c, s) for c, s in zip(chunks, sizes) if c]
        if live:
            outs, status = eng.decompress_host([c for c, _ in live])
            assert not status.any(), (it, status)
            off = 0
            k = 0
            for i, s in enumerate(sizes):
                if chunks[i]:
                    assert outs[k].tobytes() == raw[off:off + s].tobytes(), (it, i)
                    k += 1
                off += s


# ---- BloscLZ (enums::codec::blosclz) and lz4hc chunks ---------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float16, np.float32])
@pytest.mark.parametrize("family", ["tiled", "zero", "random", "natural"])
def test_blosclz_bytes_and_pixels_equal_oracle(eng, dtype, family):
    arr = getattr(synth, family + "_channel")(dtype, 1024, 200)
    it = np.dtype(dtype).itemsize
    for clevel in (9, 5, 2):
        _roundtrip(eng, dtype, arr, 4 * 1024 * 1024 // (1024 * it) * 1024 * it, clevel=clevel, compcode=hip.BLOSCLZ)
    _roundtrip(eng, dtype, arr[:100], 40000 // it * it, compcode=hip.BLOSCLZ)                  # leftover blocks, short last chunk
    _roundtrip(eng, dtype, arr[:64], 65536, blocksize=4096, compcode=hip.BLOSCLZ, clevel=7)


def test_blosclz_golden_vectors_through_the_gpu_encoder(eng, golden_dir):
    """typesize 1, no filter, block = whole chunk: the chunk's single stream IS the blosclz_compress call, and the
    vectors are what c-blosc 1.21's BloscLZ 2.3.0 emitted (tests/golden/make_blosclz_golden.py)."""
    kat = np.load(os.path.join(golden_dir, "blosclz_kat.npz"))
    by_level = {}
    for name in kat["cases"]:
        fam, n, clevel = str(name).rsplit("|", 2)
        by_level.setdefault(int(clevel), []).append((str(name), fam, int(n)))
    checked = 0
    for clevel, cases in sorted(by_level.items()):
        raw = np.concatenate([kat[f"in|{fam}|{n}"] for _, fam, n in cases])
        sizes = [n for _, _, n in cases]
        p = hip.cparams(1, clevel=clevel, blocksize=65535, filters=(0, 0, 0, 0, 0, 0), splitmode=2, compcode=hip.BLOSCLZ)
        chunks = eng.compress_host(p, raw, sizes, [s + 4096 for s in sizes])
        for (name, fam, n), c in zip(cases, chunks):
            src = kat[f"in|{fam}|{n}"]
            if (src == src[0]).all():
                continue                                            # run token, the codec is not called
            want = kat["out|" + name].tobytes()
            (cs,) = struct.unpack_from("<i", c, 32 + 4)
            if want:
                assert cs == len(want) and c[40:40 + cs] == want, name
            else:
                assert cs == n and c[40:40 + n] == src.tobytes(), name
            checked += 1
        outs, status = eng.decompress_host(chunks)
        assert not status.any() and b"".join(o.tobytes() for o in outs) == raw.tobytes()
    assert checked > 200


def test_config1_as_named_blosclz_u8(eng):
    """BASELINE configs[0] as named: one 1024^2 uint8 channel, blosclz level 9 -> a single 1 MiB remainder chunk in the
    nominal 4 MiB + 32 buffer (schunk.h:73)."""
    rng = np.random.default_rng(11)
    for arr in (synth.tiled_channel(np.uint8, 1024, 1024), synth.natural_channel(np.uint8, 1024, 1024),
                rng.integers(0, 256, 1024 * 1024, dtype=np.uint8)):
        (c,) = _roundtrip(eng, np.uint8, arr, 1024 * 1024, destsize=4 * 1024 * 1024 + 32, compcode=hip.BLOSCLZ)
        assert struct.unpack_from("<i", c, 4)[0] == 1024 * 1024 and (c[2] >> 5) == 0
    assert len(c) > 1024 * 1024 + 32 and not (c[2] & 0x02)        # random bytes: framed, not memcpyed


@pytest.mark.parametrize("filt", ["shuffle", "bitshuffle"])
def test_config3_as_named_blosclz_full_size_random_access(eng, filt):
    """BASELINE configs[2] as named: an 8192^2 uint16 channel, blosclz level 9 (+ bitshuffle; the reference's own filter
    is the byte shuffle: both run), 32 chunks of 4 MiB, visited in a seeded random permutation: get_chunk, +1, set_chunk."""
    filters = (0, 0, 0, 0, 0, hip.BITSHUFFLE if filt == "bitshuffle" else hip.SHUFFLE)
    rng = np.random.default_rng(99)
    chan = synth.tiled_channel(np.uint16, 8192, 8192, c=1)
    host = chan.view(np.uint8).ravel()
    n, chunk = host.size, 4 * 1024 * 1024
    nchunks, stride = n // chunk, chunk + 64
    d_raw, d_comp, d_one = eng.alloc(n), eng.alloc(nchunks * stride), eng.alloc(chunk)
    d_raw.upload(host)
    raw_off, comp_off = np.arange(nchunks) * chunk, np.arange(nchunks) * stride
    p = hip.cparams(2, compcode=hip.BLOSCLZ, filters=filters)
    cbytes = eng.compress_device(p, d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
    # (bit rows: BloscLZ 2.3.0 probes the first eighth of a stream -- the noise bits -- gives up, and the chunks come out
    # memcpyed; the byte-shuffled planes compress)
    assert (cbytes > 32).all() and ((cbytes < chunk).all() or filt == "bitshuffle")
    po = O.cparams(2, compcode=O.BLOSCLZ, filters=filters)
    order = rng.permutation(nchunks)
    for i in order:                                                # every chunk of the channel (round 3: was 12 of 32)
        eng.decompress_device(d_comp.ptr + int(comp_off[i]), [0], [chunk], [32768], d_one.ptr, [0])       # get_chunk
        px = d_one.download().view(np.uint16) + np.uint16(1)
        d_one.upload(px)
        cbytes[i] = eng.compress_device(p, d_one.ptr, [0], [chunk], d_comp.ptr + int(comp_off[i]), [0], [chunk + 32])[0]   # set_chunk
    comp = d_comp.download()
    want_px = chan.ravel() + np.uint16(1)
    ocb, ochunk = _oracle_all_chunks(po, np.ascontiguousarray(want_px.view(np.uint8)), chunk)
    for i in range(nchunks):                                       # bytes against the oracle on every chunk ...
        assert cbytes[i] == ocb[i] and comp[comp_off[i]:comp_off[i] + cbytes[i]].tobytes() == ochunk(i).tobytes(), i
    eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_raw.ptr, raw_off)
    assert d_raw.download().tobytes() == want_px.tobytes()         # ... pixels everywhere
    for buf in (d_raw, d_comp, d_one):
        buf.free()


def test_blosclz_edge_geometries_and_destsize_rule(eng):
    rng = np.random.default_rng(4)
    _roundtrip(eng, np.uint8, np.arange(50, dtype=np.uint8), 4194304, compcode=hip.BLOSCLZ)     # test_channel.cpp:74-87 (one tiny chunk)
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 300, 41), 5000, blocksize=256, compcode=hip.BLOSCLZ)
    _roundtrip(eng, np.uint64, (rng.integers(0, 40, 6000, dtype=np.uint64) * 0x0101010101).astype(np.uint64), 16384, blocksize=4096, compcode=hip.BLOSCLZ)
    _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 512, 256), 262144, blocksize=65534, compcode=hip.BLOSCLZ)
    noisy = rng.integers(0, 65536, 20000, dtype=np.uint16)
    noisy[3000:9000] = 7
    noisy[9000:15000] = (np.arange(6000) // 50).astype(np.uint16)
    for destsize in (40000 + 32, 39000, 36000, 30000, 20100, 200):               # blosc2's running-destsize rule
        _roundtrip(eng, np.uint16, noisy, 40000, blocksize=4096, destsize=destsize, compcode=hip.BLOSCLZ)
    for splitmode, filters in ((2, (0, 0, 0, 0, 0, 1)), (1, (0, 0, 0, 0, 0, 0)), (4, (0, 0, 0, 0, 0, 1))):
        _roundtrip(eng, np.uint16, synth.natural_channel(np.uint16, 2048, 64), 131072, splitmode=splitmode, filters=filters, compcode=hip.BLOSCLZ)


def test_damaged_blosclz_chunks_are_reported_not_crashed(eng):
    a = synth.natural_channel(np.uint16, 1024, 64)
    (good,) = eng.compress_host(hip.cparams(2, compcode=hip.BLOSCLZ), a, [a.nbytes], [a.nbytes + 32])
    rng = np.random.default_rng(8)
    agree = 0
    for _ in range(60):
        bad = bytearray(good)
        k = int(rng.integers(32 + 16, len(bad)))
        bad[k] ^= 1 << int(rng.integers(0, 8))
        r, pix = O.decompress(bytes(bad), a.nbytes)
        try:
            outs, status = eng.decompress_host([bytes(bad)])
            ok = not status.any()
        except hip.CodecError:
            ok = False
        assert ok == (r == a.nbytes)
        if ok:
            assert outs[0].tobytes() == pix.tobytes()
        agree += 1
    assert agree == 60


def test_truncated_chunk_buffer_is_refused_before_it_is_read(eng):
    """cimg_decompress_batch_host_sized: the header claims more compressed bytes than the buffer holds -> READ_BUFFER."""
    a = synth.natural_channel(np.uint16, 512, 64)
    (good,) = eng.compress_host(hip.cparams(2), a, [a.nbytes], [a.nbytes + 32])
    for cut in (len(good) // 2, 40, 31, 8):
        with pytest.raises(hip.CodecError) as ei:
            eng.decompress_host([good[:cut]])
        assert ei.value.code == -5
    outs, status = eng.decompress_host([good])
    assert outs[0].tobytes() == a.tobytes()


def test_lz4hc_chunks_decode_on_the_gpu(eng, golden_dir):
    """enums::codec::lz4hc chunks (coded by liblz4's LZ4_compress_HC, tests/golden/make_lz4hc_golden.py) are codec
    format 1: they decode through the ordinary kernels (what this engine writes under that name: test_zstd_and_lz4hc_encoders...)."""
    kat = np.load(os.path.join(golden_dir, "lz4hc_kat.npz"))
    chunks = [kat["chunk|" + str(n)].tobytes() for n in kat["cases"]]
    outs, status = eng.decompress_host(chunks)
    assert not status.any()
    for n, o in zip(kat["cases"], outs):
        assert o.tobytes() == kat["in|" + str(n)].tobytes(), n


def test_zstd_chunks_decode_on_the_gpu(eng, golden_dir):
    """enums::codec::zstd chunks (streams = frames of the system libzstd, tests/golden/make_zstd_golden.py) are codec format 4:
    cimg_decode_blocks hands them to cimg_decode_zstd (csrc/zstd_kernel.h, the slow path).  Alone, and mixed into a batch with
    LZ4 chunks; a damaged one fails by itself.  (The engine's own zstd chunks: test_zstd_and_lz4hc_encoders_write_valid_chunks.)"""
    kat = np.load(os.path.join(golden_dir, "zstd_kat.npz"))
    names = [str(n) for n in kat["chunks"]]
    chunks = [kat["chunk|" + n].tobytes() for n in names]
    outs, status = eng.decompress_host(chunks)
    assert not status.any()
    for n, o in zip(names, outs):
        assert o.tobytes() == kat["cin|" + n].tobytes(), n
    # mixed with chunks of this engine's own codecs
    a = synth.tiled_channel(np.float16, 512, 64)
    (lz,) = eng.compress_host(hip.cparams(2), a, [a.nbytes], [a.nbytes + 32])
    outs, status = eng.decompress_host([lz, chunks[0], lz, chunks[2]])
    assert not status.any()
    assert outs[0].tobytes() == a.tobytes() and outs[2].tobytes() == a.tobytes()
    assert outs[1].tobytes() == kat["cin|" + names[0]].tobytes() and outs[3].tobytes() == kat["cin|" + names[2]].tobytes()
    # damage inside the first zstd frame of a chunk: that chunk fails, its neighbour is decoded
    bad = bytearray(chunks[0])
    at = chunks[0].find(b"\x28\xb5\x2f\xfd")
    for k in range(at + 5, at + 40):
        bad[k] ^= 0x5A
    outs, status = eng.decompress_host([chunks[1], bytes(bad)], check=False)
    assert status[0] == 0 and status[1] < 0
    assert outs[0].tobytes() == kat["cin|" + names[1]].tobytes()
    # ... and through the reference's own entry point, one chunk per call (blosc2/wrapper.h:236-259 -> blosc2_decompress_ctx)
    L = hip.load()
    dp = hip.Blosc2DParams()
    dp.nthreads = 1
    dctx = L.blosc2_create_dctx(dp)
    for n, c in zip(names, chunks):
        want = kat["cin|" + n]
        buf = np.frombuffer(c, np.uint8).copy()
        out = np.zeros(want.size, np.uint8)
        assert L.blosc2_decompress_ctx(dctx, buf.ctypes.data, buf.size, out.ctypes.data, out.size) == want.size, n
        assert out.tobytes() == want.tobytes(), n
    L.blosc2_free_ctx(dctx)


@pytest.mark.parametrize("form", ["lanes_8", "lanes_3", "walkers_decode_sequences", "fused", "plans_overflow", "groups_of_9_blocks", "no_memory_for_plans"])
def test_every_form_of_the_zstd_read_path_on_the_gpu(form, golden_dir, monkeypatch):
    """The read path of zstd chunks is several launches (engine.hip: decompress_finish -- cimg_zstd_walk, cimg_zstd_lit beside
    cimg_zstd_seq, cimg_zstd_replay) with cimg_decode_zstd behind them for blocks whose plan does not fit its slot.  Every form the
    environment can select -- lanes per wave of the sequence decoder, sequences decoded by the walkers themselves, the fused
    kernels alone, plans of 256 bytes (most blocks refused) -- decodes the golden chunks and a batch of local ones bit-exactly,
    fails a damaged chunk by itself, and says how many plans were refused."""
    env = {"lanes_8": {}, "lanes_3": {"CIMG_ZSTD_LANES": "3"}, "walkers_decode_sequences": {"CIMG_ZSTD_LANES": "0"},
           "fused": {"CIMG_ZSTD_FUSED": "1"}, "plans_overflow": {"CIMG_ZSTD_PLAN_CAP": "256"},
           "groups_of_9_blocks": {"CIMG_ZSTD_PLAN_MIB": "1"},                   # (a batch whose plans exceed the plan memory goes in groups of blocks)
           "no_memory_for_plans": {"CIMG_ZSTD_PLAN_FAIL": "1"}}[form]           # (the plans' allocation fails: the fused kernel reads the batch -- ADVICE r4)
    for k in ("CIMG_ZSTD_LANES", "CIMG_ZSTD_FUSED", "CIMG_ZSTD_PLAN_CAP", "CIMG_ZSTD_PLAN_MIB", "CIMG_ZSTD_PLAN_FAIL"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = hip.Engine(0)
    try:
        kat = np.load(os.path.join(golden_dir, "zstd_kat.npz"))
        names = [str(n) for n in kat["chunks"]]
        chunks = [kat["chunk|" + n].tobytes() for n in names]
        bad = bytearray(chunks[0])
        at = chunks[0].find(b"\x28\xb5\x2f\xfd")
        for k in range(at + 5, at + 40):
            bad[k] ^= 0x5A
        batch = chunks + [bytes(bad)] + chunks[::-1]
        outs, status = e.decompress_host(batch, check=False)
        want = names + [None] + names[::-1]
        for i, n in enumerate(want):
            if n is None:
                assert status[i] < 0
            else:
                assert status[i] == 0 and outs[i].tobytes() == kat["cin|" + n].tobytes(), (form, i, n)
        if O.zstd_available():
            # 8 MiB of each family as libzstd writes it at the reference's default level (one frame per block) and at a split one
            for fam, clevel in ((synth.tiled_channel, 9), (synth.natural_channel, 9), (synth.natural_channel, 5)):
                a = fam(np.float32, 2048, 1024)
                raw = np.ascontiguousarray(a).view(np.uint8).ravel()
                p = O.cparams(4, clevel=clevel, blocksize=32768, compcode=O.ZSTD)
                made = [O.compress(p, raw[o:o + (4 << 20)])[1] for o in range(0, raw.size, 4 << 20)]
                outs, status = e.decompress_host(made)
                assert not status.any()
                assert b"".join(o.tobytes() for o in outs) == raw.tobytes(), (form, fam.__name__, clevel)
        st = e.zstd_stats()
        assert st["zstd_batches"] >= 1
        assert (st["blocks_refused"] > 0) == (form == "plans_overflow"), (form, st)
    finally:
        e.close()


def test_zstd_chunks_made_on_this_box(eng, golden_dir):
    """Beyond the committed vectors: chunks of every element size, split and unsplit, made here with the box's own libzstd
    (skipped where there is none) and framed as c-blosc2 frames them (tests/golden/make_zstd_golden.py) -- 1 MiB each, ragged
    last block included -- decode to their pixels."""
    import ctypes as C
    import ctypes.util
    name = ctypes.util.find_library("zstd")
    if not name:
        pytest.skip("no libzstd on this box")
    sys.path.insert(0, golden_dir)
    import make_zstd_golden as G
    z = C.CDLL(name)
    z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t; z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    chunks, want = [], []
    for dtype, fam in ((np.uint8, synth.natural_channel), (np.uint16, synth.tiled_channel), (np.float16, synth.natural_channel),
                       (np.float32, synth.tiled_channel), (np.float32, synth.natural_channel)):
        it = np.dtype(dtype).itemsize
        a = fam(dtype, 1024, 1024 // it + 3)                       # a little over 1 MiB: the last block is a leftover
        src = np.ascontiguousarray(a).view(np.uint8).ravel()
        for clevel in (1, 5, 9):
            for filt in ("shuffle", "bitshuffle", "none"):
                chunks.append(G.frame(z, src, it, 32768, clevel, filt))
                want.append(src)
    outs, status = eng.decompress_host(chunks)
    assert not status.any()
    for k, (o, w) in enumerate(zip(outs, want)):
        assert o.tobytes() == w.tobytes(), k
    # other block sizes: 128 and 64 KiB (the kernel's LDS follows the batch's largest block; unsplit streams above 64 KiB), 4 KiB
    a = synth.natural_channel(np.uint16, 1024, 600)
    src = np.ascontiguousarray(a).view(np.uint8).ravel()
    # (144 KiB: longer than zstd's largest block -- frames of two blocks, i.e. two jobs per frame for the lane decoders, the second
    # with repeated tables / treeless literals and the repeat offsets carried over)
    # (... which only the planned path holds: under the switches that send blocks to cimg_decode_zstd -- tools/switch_matrix.sh -- that
    # size is the documented ERR_CODEC_SUPPORT, tests/test_emu_zstd.py has the case)
    to_fused = os.environ.get("CIMG_ZSTD_FUSED") or os.environ.get("CIMG_ZSTD_PLAN_CAP") or os.environ.get("CIMG_ZSTD_PLAN_FAIL")
    for bs in ((131072, 65536, 4096) if to_fused else (147456, 131072, 65536, 4096)):
        chunks = [G.frame(z, src, 2, bs, clevel) for clevel in (3, 9)]
        outs, status = eng.decompress_host(chunks)
        assert not status.any(), bs
        assert all(o.tobytes() == src.tobytes() for o in outs), bs


def test_mutated_zstd_chunks_fail_or_decode_but_never_hang(eng, golden_dir):
    """240 bit-flipped copies of the golden zstd chunks in batches of 60: every chunk ends with a status (0 = the flip hit
    bytes that do not matter, or a stored stream; < 0 = rejected), the undamaged chunk riding along in each batch still decodes."""
    kat = np.load(os.path.join(golden_dir, "zstd_kat.npz"))
    rng = np.random.default_rng(99)
    names = [str(n) for n in kat["chunks"]]
    good = kat["chunk|" + names[0]].tobytes()
    want = kat["cin|" + names[0]].tobytes()
    for rnd in range(4):
        batch = [good]
        for k in range(60):
            c = bytearray(kat["chunk|" + names[(rnd + k) % len(names)]].tobytes())
            for _ in range(1 + k % 3):
                c[int(rng.integers(32, len(c)))] ^= 1 << int(rng.integers(0, 8))
            batch.append(bytes(c))
        outs, status = eng.decompress_host(batch, check=False)
        assert status[0] == 0 and outs[0].tobytes() == want
        assert (status <= 0).all()


def test_config5_zstd_full_size_one_rank_share(eng):
    """BASELINE configs[4] as named, one rank's share: ONE 16384 x 16384 float32 channel = 256 chunks x 4 MiB = 32768 blocks of
    32 KiB, zstd + byte shuffle (enums.h:18-24; blosc2/wrapper.h:236-259).  The chunks are made here by the box's libzstd under
    the checker's chunk layer (oracle/zstd_dl.c) the way c-blosc2 frames them: the whole GiB at a fast level (clevel 3: split
    planes, 131072 frames) in ONE device-resident decode call, and a 32-chunk slice at the reference's default clevel 9
    (= ZSTD_maxCLevel(), one unsplit frame per block).  Pixels must come back bit-exact; the ratio is printed."""
    if not O.zstd_available():
        pytest.skip("no libzstd on this box")
    import ctypes as C
    W = H = 16384
    chan = synth.tiled_channel(np.float32, W, H)
    host = chan.view(np.uint8).ravel()
    CH = 4 << 20
    L = O.lib()
    L.orc_bench_compress.argtypes = [C.POINTER(O.CParams), C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_int]
    L.orc_bench_compress.restype = C.c_int64
    threads = min(len(os.sched_getaffinity(0)), 16)

    def make(first, count, clevel):
        stride = CH + 64
        comp = np.zeros(count * stride, np.uint8)
        cb = np.zeros(count, np.int32)
        p = O.cparams(4, clevel=clevel, blocksize=32768, compcode=O.ZSTD)
        r = L.orc_bench_compress(C.byref(p), host[first * CH:].ctypes.data, count, CH, comp.ctypes.data, stride, CH + 32, cb.ctypes.data,
                                 min(threads, count), max(1, threads // min(threads, count)))
        assert r > 0
        return comp, cb, stride

    for first, count, clevel in ((0, 256, 3), (96, 32, 9)):
        comp, cb, stride = make(first, count, clevel)
        n = count * CH
        d_comp = eng.alloc(comp.size)
        d_comp.upload(comp)
        d_raw = eng.alloc(n)
        st = eng.decompress_device(d_comp.ptr, np.arange(count, dtype=np.int64) * stride, [CH] * count, [32768] * count,
                                   d_raw.ptr, np.arange(count, dtype=np.int64) * CH, comp_size=cb)
        assert not st.any()
        got = d_raw.download(n)
        assert got.tobytes() == host[first * CH:first * CH + n].tobytes(), clevel
        flags = comp[2]
        assert (flags >> 5) == 4 and bool(flags & 0x10) == (clevel > 5)          # codec format 4; unsplit above clevel 5
        print(f"configs[4] share: {count} chunks at zstd clevel {clevel}: ratio {n / float(cb.sum()):.3f}")
        d_comp.free(); d_raw.free()


def test_zstd_and_lz4hc_encoders_write_valid_chunks(eng):
    """enums::codec::zstd / lz4hc (enums.h:18-24) construct and compress.  Their chunks are FORMAT-VALID, NOT BYTE-PINNED (DESIGN.md
    section 2): zstd streams are frames built by csrc/zstd_encode.h (raw literals + predefined FSE tables over the wave's LZ4
    matches), lz4hc streams are LZ4 blocks from the fast match finder.  Checked here: (1) the box's own libzstd (through the
    checker's chunk layer) and liblz4's format twin decode every chunk to the pixels, (2) so do the GPU decoders, (3) the bytes
    equal what the kernel sources produce on the host lane emulator, (4) headers say what c-blosc2 would say (compcode, codec
    format, the split rule of the level), (5) lz4hc chunks equal the checker's twin byte for byte.  Ratios are printed."""
    import _emu as Em
    cases = []
    for dtype, fam in ((np.float16, "tiled"), (np.uint16, "natural"), (np.float32, "tiled"), (np.uint8, "natural"), (np.uint16, "zero"), (np.uint16, "random")):
        cases.append((dtype, getattr(synth, fam + "_channel")(dtype, 1024, 200), fam))
    for codec, name in ((hip.ZSTD, "zstd"), (hip.LZ4HC, "lz4hc")):
        for clevel in (9, 3):
            for dtype, arr, fam in cases:
                it = np.dtype(dtype).itemsize
                raw = np.ascontiguousarray(arr).view(np.uint8).ravel()
                chunk = 131072
                sizes = [min(chunk, raw.size - o) for o in range(0, raw.size, chunk)]
                got = eng.compress_host(hip.cparams(it, clevel=clevel, compcode=codec), raw, sizes, [chunk + 32] * len(sizes))
                rc, cb, emu = Em.compress_batch(Em.cparams(it, clevel=clevel, compcode=codec), raw, sizes, [chunk + 32] * len(sizes))
                assert rc == 0 and got == emu, (name, clevel, np.dtype(dtype).name, fam)
                off = 0
                for c, n in zip(got, sizes):
                    assert 0 < len(c) <= chunk + 32                                  # (a remainder chunk of noise in a nominal buffer stays block-framed: SURVEY N7)
                    assert c[22] == codec and (c[2] >> 5) == (4 if codec == hip.ZSTD else 1) or (c[2] & 0x02)      # (memcpyed chunks keep the bits too)
                    if not (c[2] & 0x02) and it > 1:
                        assert bool(c[2] & 0x10) == (not (codec == hip.ZSTD and clevel <= 5)), (name, clevel)      # zstd splits up to level 5, lz4hc never
                    if codec == hip.ZSTD and not O.zstd_available():
                        pass
                    else:
                        r, px = O.decompress(c)
                        assert r == n and px.tobytes() == raw[off:off + n].tobytes(), (name, clevel, fam)
                    if codec == hip.LZ4HC:
                        assert c == O.compress(O.cparams(it, clevel=clevel, compcode=O.LZ4HC), raw[off:off + n], destsize=chunk + 32)[1]
                    off += n
                outs, status = eng.decompress_host(got)
                assert not status.any() and b"".join(o.tobytes() for o in outs) == raw.tobytes(), (name, clevel, fam)
                print(f"{name} clevel {clevel} {np.dtype(dtype).name} {fam}: ratio {raw.size / sum(map(len, got)):.3f}")


def test_leftover_blocks_have_a_launch_of_their_own(eng):
    """Chunk sizes that are no multiple of the block size (every image whose row size does not divide 4 MiB): the last block of
    every chunk is shorter and never split, the lean decode kernel leaves it, and from the second such batch on the general kernel
    is launched over exactly those blocks.  Decoded several times in a row (the launch shape follows the previous batch), then
    with chunks whose blocks the lean kernel cannot take for their CONTENT (both planes coded: the late full launch), then again."""
    chunk = 7 * 32768 + 9000                                              # seven full blocks + a leftover
    tiled = np.ascontiguousarray(synth.tiled_channel(np.float16, 1024, 400)).view(np.uint8).ravel()
    natural = np.ascontiguousarray(synth.natural_channel(np.float16, 1024, 400)).view(np.uint8).ravel()
    def batch(raw, n):
        sizes = [chunk] * n
        chunks = eng.compress_host(hip.cparams(2), raw[:chunk * n], sizes, [chunk + 32] * n)
        for i, c in enumerate(chunks):
            assert c == O.compress(O.cparams(2), raw[i * chunk:(i + 1) * chunk], destsize=chunk + 32)[1]
        return chunks, raw[:chunk * n]
    t_chunks, t_raw = batch(tiled, 3)
    n_chunks, n_raw = batch(natural, 3)
    mixed = [t_chunks[0], n_chunks[1], t_chunks[2]]
    mixed_raw = np.concatenate([t_raw[:chunk], n_raw[chunk:2 * chunk], t_raw[2 * chunk:3 * chunk]])
    # chunks of different sizes in one batch (an image's remainder chunks), one of them without a leftover block
    sizes = [chunk, 3 * 32768 + 100, 4 * 32768, 2 * 32768 + 30000]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ragged_raw = tiled[:offs[-1]]
    ragged = eng.compress_host(hip.cparams(2), ragged_raw, sizes, [s + 32 for s in sizes])
    for i, c in enumerate(ragged):
        assert c == O.compress(O.cparams(2), ragged_raw[offs[i]:offs[i + 1]], destsize=sizes[i] + 32)[1]
    for chunks, raw in ((t_chunks, t_raw), (t_chunks, t_raw), (t_chunks, t_raw), (n_chunks, n_raw), (t_chunks, t_raw), (mixed, mixed_raw), (mixed, mixed_raw),
                        (t_chunks, t_raw), (t_chunks, t_raw), (ragged, ragged_raw), (ragged, ragged_raw), (ragged, ragged_raw), (n_chunks, n_raw), (ragged, ragged_raw)):
        outs, status = eng.decompress_host(chunks)
        assert not status.any()
        got = b"".join(o.tobytes() for o in outs)
        assert got == raw.tobytes()


def test_leftover_blocks_device_resident_side_by_side(eng):
    """The same geometry through the device-resident calls, several batches in a row: from the second one on nothing of a batch
    is on the engine's stream before its launches, so the leftover blocks' launch is enqueued on the side stream without waiting
    for anything (engine.hip: compress_launch) and runs beside the large one.  Bytes against the oracle every time; lz4 and
    blosclz; then decoded (leftover blocks: the general kernel over exactly them, beside the lean launch)."""
    chunk = 31 * 32768 + 31744                                            # a 1920-pixel float16 row size would give 127 + 31744
    nchunks = 12
    raws = [np.ascontiguousarray(f(np.float16, 1024, chunk * nchunks // 2048 + 1)).view(np.uint8).ravel()[:chunk * nchunks] for f in (synth.tiled_channel, synth.natural_channel)]
    stride = (chunk + 32 + 63) // 64 * 64
    d_raw, d_comp, d_out = eng.alloc(chunk * nchunks), eng.alloc(stride * nchunks), eng.alloc(chunk * nchunks)
    raw_off = np.arange(nchunks, dtype=np.int64) * chunk
    comp_off = np.arange(nchunks, dtype=np.int64) * stride
    for compcode in (hip.LZ4, hip.BLOSCLZ):
        for raw in raws:
            d_raw.upload(raw)
            want = [O.compress(O.cparams(2, compcode=compcode), raw[i * chunk:(i + 1) * chunk], destsize=chunk + 32)[1] for i in range(nchunks)]
            for rep in range(4):
                cb = eng.compress_device(hip.cparams(2, compcode=compcode), d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
                blob = d_comp.download(stride * nchunks)
                for i in range(nchunks):
                    assert cb[i] == len(want[i]) and blob[i * stride:i * stride + cb[i]].tobytes() == want[i], (compcode, rep, i)
                eng.decompress_device(d_comp.ptr, comp_off, [chunk] * nchunks, [32768] * nchunks, d_out.ptr, raw_off)
                assert d_out.download(chunk * nchunks).tobytes() == raw.tobytes(), (compcode, rep)


def test_randomized_geometries_zstd_and_lz4hc_round_trips(eng):
    """The write side of the two format-valid codecs over random element size, block size (up to 64 KiB: a zstd stream's limit),
    chunk size, level (split / unsplit), filter, dest capacity and data make-up: every chunk the GPU writes is decoded (1) by the
    GPU decoders and (2) by the checker -- libzstd / liblz4's twin under the oracle's chunk layer -- to the input, and equals what
    the kernel sources write on the host lane emulator (every 8th case: the emulator is slow)."""
    import _emu as Em
    rng = np.random.default_rng(20260303 + int(os.environ.get("CIMG_TEST_SEED", "0")))
    have_zstd = O.zstd_available()
    for it in range(int(os.environ.get("CIMG_TEST_ROUNDS", "400")) // 2):
        codec = hip.ZSTD if it % 3 else hip.LZ4HC
        ts = int(rng.choice([1, 2, 2, 4, 4, 8, 3]))
        blocksize = int(rng.choice([256, 1024, 4096, 8192, 32768, 65536])) // ts * ts
        nchunks = int(rng.integers(1, 4))
        chunk = int(rng.integers(1, 7)) * blocksize + (int(rng.integers(0, blocksize)) // ts * ts if rng.random() < 0.4 else 0)
        chunk = min(chunk, 300000) // ts * ts or ts
        total = max(chunk * (nchunks - 1) + int(rng.integers(1, chunk + 1)) // ts * ts, ts)
        raw = _mixed_data(rng, total, ts)
        clevel = int(rng.choice([1, 5, 9, 9]))
        filt = int(rng.choice([0, 1, 1, 1, 2]))
        dest = chunk + 32 if rng.random() < 0.7 else max(40, int(chunk * rng.uniform(0.3, 1.0)))
        sizes = [min(chunk, total - o) for o in range(0, total, chunk)]
        p = hip.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=codec, filters=(0, 0, 0, 0, 0, filt))
        chunks = eng.compress_host(p, raw, sizes, [dest] * len(sizes))
        what = (it, codec, ts, blocksize, chunk, clevel, filt, dest)
        if it % 8 == 0:
            rc, cb, emu = Em.compress_batch(Em.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=codec, filters=(0, 0, 0, 0, 0, filt)), raw, sizes, [dest] * len(sizes))
            assert rc == 0 and [c for c in emu] == [c for c in chunks], what
        off = 0
        for c, n in zip(chunks, sizes):
            if c and (codec == hip.LZ4HC or have_zstd):
                r, px = O.decompress(c)
                assert r == n and px.tobytes() == raw[off:off + n].tobytes(), what
            off += n
        live = [c for c in chunks if c]                                      # (b"": the chunk did not fit dest -- the reference's error)
        if live:
            outs, status = eng.decompress_host(live)
            assert not status.any(), (what, status)
            off = k = 0
            for c, n in zip(chunks, sizes):
                if c:
                    assert outs[k].tobytes() == raw[off:off + n].tobytes(), what
                    k += 1
                off += n


@pytest.mark.parametrize("compcode", [hip.LZ4, hip.BLOSCLZ])
def test_every_combination_of_plane_kinds_decodes_on_the_gpu(eng, compcode):
    """All 9 / 81 combinations of plane kinds (coded / stored / run) of a 32 KiB block of 2- / 4-byte elements, back to back in
    one chunk and again with the blocks in reverse order (tests/test_emu_kernels.py: plane_kind_blocks): the lean kernel's shapes,
    each behind every other kind of LDS content.  Bytes against the oracle, pixels back."""
    from test_emu_kernels import plane_kind_blocks
    rng = np.random.default_rng(77 + compcode)
    for ts in (2, 4):
        fwd = plane_kind_blocks(ts, rng)
        rev = np.ascontiguousarray(fwd.reshape(-1, 32768)[::-1]).ravel()
        for raw in (fwd, rev):
            (c,) = eng.compress_host(hip.cparams(ts, compcode=compcode), raw, [raw.size], [raw.size + 32])
            assert c == O.compress(O.cparams(ts, compcode=compcode), raw, destsize=raw.size + 32)[1]
            outs, status = eng.decompress_host([c, c, c])                  # three copies: blocks also meet across chunk boundaries
            assert not status.any()
            for o in outs:
                bad = np.nonzero(o != raw)[0]
                assert bad.size == 0, (ts, compcode, "first wrong byte %d = block %d" % (bad[0], bad[0] // 32768))


def test_two_engines_encode_concurrently_with_partly_resident_launches(eng):
    """Two engines, two threads, sixty configs[1] compress batches each, at the same time (encode_kernel.h:
    EncodeArgs::claim, EncodeStream::run).  Each encode launch is sized to fill the device, so the workgroups of one are partly
    NOT resident while the other runs -- and a wave that is out of queue work waits, inside the launch, for its chunks to close.
    First items are claimed, and waiting waves take over the unclaimed ones, so every launch completes with whatever is resident:
    no batch may fail (cbytes -1 = "a chunk was never published"), and every batch's bytes are the oracle's."""
    import threading
    chans = [synth.tiled_channel(np.float16, 4096, 4096, c=c) for c in range(4)]
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    n, chunk = host.size, 4 * 1024 * 1024
    nchunks, stride = n // chunk, chunk + 64
    raw_off = np.arange(nchunks, dtype=np.int64) * chunk
    comp_off = np.arange(nchunks, dtype=np.int64) * stride
    cb_want, chunk_want = _oracle_all_chunks(O.cparams(2), host, chunk)
    want = np.zeros(nchunks * stride, np.uint8)
    for i in range(nchunks):
        want[i * stride:i * stride + cb_want[i]] = chunk_want(i)
    engines = [eng, hip.Engine(0)]
    bufs = []
    for e in engines:
        d_raw, d_comp = e.alloc(n), e.alloc(nchunks * stride)
        d_raw.upload(host)
        bufs.append((d_raw, d_comp))
    errors = []
    start = threading.Barrier(2)

    def worker(k):
        e, (d_raw, d_comp) = engines[k], bufs[k]
        try:
            start.wait()
            for rep in range(60):
                cb = e.compress_device(hip.cparams(2), d_raw.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
                assert (np.asarray(cb) == cb_want).all(), (k, rep, "sizes")
                if rep % 10 == 9 or rep == 0:
                    blob = d_comp.download()
                    for i in range(nchunks):
                        assert blob[i * stride:i * stride + cb[i]].tobytes() == want[i * stride:i * stride + cb[i]].tobytes(), (k, rep, i)
        except BaseException as ex:                        # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    alive = [t.is_alive() for t in threads]
    for (d_raw, d_comp) in bufs:
        d_raw.free(); d_comp.free()
    engines[1].close()
    assert not any(alive), "an encode batch did not come back"
    assert not errors, errors


def test_decode_batches_of_alternating_geometry_with_leftover_blocks(eng):
    """the leftover blocks' decode launch runs on the side stream; it reads the batch's descriptors, so it must
    be ordered behind their upload.  Two geometries (chunk counts, chunk sizes, offsets) ALTERNATE through the device-resident
    call -- every batch uploads new descriptors -- and every batch's pixels are compared."""
    geos = []
    for chunk, nchunks, fam in ((31 * 32768 + 31744, 12, synth.tiled_channel), (17 * 32768 + 2048, 7, synth.tiled_channel), (9 * 32768 + 30000, 20, synth.natural_channel)):
        raw = np.ascontiguousarray(fam(np.float16, 1024, chunk * nchunks // 2048 + 1)).view(np.uint8).ravel()[:chunk * nchunks]
        stride = (chunk + 32 + 63) // 64 * 64
        chunks = [O.compress(O.cparams(2), raw[i * chunk:(i + 1) * chunk], destsize=chunk + 32)[1] for i in range(nchunks)]
        blob = np.zeros(stride * nchunks, np.uint8)
        for i, c in enumerate(chunks):
            blob[i * stride:i * stride + len(c)] = np.frombuffer(c, np.uint8)
        d_comp, d_out = eng.alloc(blob.size), eng.alloc(raw.size)
        d_comp.upload(blob)
        geos.append((chunk, nchunks, raw, stride, d_comp, d_out))
    for rep in range(12):
        chunk, nchunks, raw, stride, d_comp, d_out = geos[rep % len(geos)]
        d_out.upload(np.zeros(raw.size, np.uint8))
        eng.decompress_device(d_comp.ptr, np.arange(nchunks, dtype=np.int64) * stride, [chunk] * nchunks, [32768] * nchunks, d_out.ptr,
                              np.arange(nchunks, dtype=np.int64) * chunk)
        assert d_out.download(raw.size).tobytes() == raw.tobytes(), rep
    for g in geos:
        g[4].free(); g[5].free()


def test_deinterleave_on_the_device_then_compress_with_leftover_blocks(eng):
    """cimg_deinterleave_device returns with its kernel enqueued on the engine's stream; a compress batch right
    behind it whose chunks have leftover blocks (their launch runs on the side stream) must still see the planes the kernel
    writes.  Alternating pictures, so that stale planes would show: chunks against the oracle's for the planar pixels."""
    nch, width, rows = 3, 1920, 24                                         # float16 rows of 3840 bytes: chunks with a leftover block
    npix = width * rows
    plane_bytes = npix * 2
    plane_stride = (plane_bytes + 15) & ~15
    chunk = plane_bytes // 2                                              # two chunks per plane: 46080 bytes = 1 block + 13312
    assert chunk % 32768 != 0 and plane_bytes % chunk == 0
    nchunks = nch * 2
    raw_off = np.array([c * plane_stride + k * chunk for c in range(nch) for k in range(2)], dtype=np.int64)
    stride = (chunk + 32 + 63) // 64 * 64
    comp_off = np.arange(nchunks, dtype=np.int64) * stride
    d_il, d_planar, d_comp = eng.alloc(npix * nch * 2), eng.alloc(plane_stride * nch), eng.alloc(stride * nchunks)
    for rep in range(6):
        planes = [np.ascontiguousarray(synth.tiled_channel(np.float16, width, rows, c=c, seed=100 + rep)).ravel() for c in range(nch)]
        il = np.stack(planes, axis=1).ravel()                             # R G B R G B ...
        d_il.upload(il.view(np.uint8))
        eng.deinterleave_device(d_il.ptr, nch, 2, npix, d_planar.ptr, plane_stride)
        cb = eng.compress_device(hip.cparams(2), d_planar.ptr, raw_off, [chunk] * nchunks, d_comp.ptr, comp_off, [chunk + 32] * nchunks)
        blob = d_comp.download(stride * nchunks)
        for c in range(nch):
            pb = planes[c].view(np.uint8)
            for k in range(2):
                i = 2 * c + k
                r, want = O.compress(O.cparams(2), pb[k * chunk:(k + 1) * chunk], destsize=chunk + 32)
                assert cb[i] == r and blob[i * stride:i * stride + r].tobytes() == want, (rep, c, k)
    for b in (d_il, d_planar, d_comp):
        b.free()
