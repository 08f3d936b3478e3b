"""The zstd frame decoder (csrc/zstd_decode.h: the read side of enums::codec::zstd, enums.h:18-24) on the host, against
frames libzstd made (tests/golden/make_zstd_golden.py): the answer of every vector is its input."""
import os

import numpy as np
import pytest

import _emu as E


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "zstd_kat.npz"))


def test_every_golden_frame_decodes_to_its_input(kat):
    n = 0
    for key in kat["frames"]:
        key = str(key)
        src = kat["in|" + key.split("|")[0]]
        r, out = E.zstd_decode(kat["frame|" + key], src.size + 8)
        assert r == src.size, (key, r)
        assert out == src.tobytes(), key
        n += 1
    assert n >= 100


def test_exact_capacity_and_short_capacity(kat):
    src = kat["in|text"]
    fr = kat["frame|text|L3"]
    r, out = E.zstd_decode(fr, src.size)
    assert r == src.size and out == src.tobytes()
    r, _ = E.zstd_decode(fr, src.size - 1)
    assert r < 0


def test_damaged_frames_fail_cleanly(kat):
    """Truncations and bit flips: an error or (for a flipped literal) wrong bytes -- never a crash, never a read or write
    outside the buffers (the emulator build of CI runs this under ASAN, tests/test_emu_fuzz.py)."""
    rng = np.random.default_rng(5)
    for key in ("text|L3", "natural_plane|L19", "tiled_hi_plane|L1", "skewed_16k|L5", "two_blocks_150k|L1"):
        fr = kat["frame|" + key].copy()
        n = kat["in|" + key.split("|")[0]].size
        for cut in (0, 3, 5, 9, fr.size // 2, fr.size - 1):
            r, _ = E.zstd_decode(fr[:cut], n)
            assert r < 0, (key, cut)
        for _ in range(60):
            bad = fr.copy()
            i = int(rng.integers(0, bad.size))
            bad[i] ^= 1 << int(rng.integers(0, 8))
            r, out = E.zstd_decode(bad, n)
            assert r <= n


def test_blosc2_zstd_chunks_decode_through_the_kernels(kat):
    """Chunks framed the way c-blosc2 frames enums::codec::zstd (codec format 4; split and unsplit, run tokens, stored
    streams, leftover blocks) go lean kernel -> general kernel (ERR_CODEC_SUPPORT) -> zstd kernel, as in the engine."""
    for name in kat["chunks"]:
        name = str(name)
        src = kat["cin|" + name]
        chunk = kat["chunk|" + name]
        bs = int(np.frombuffer(chunk[8:12].tobytes(), "<i4")[0])
        rc, status, out = E.decompress_batch([chunk.tobytes()], [src.size], [bs])
        assert rc == 0 and status == [0], (name, rc, status)
        assert out[0].tobytes() == src.tobytes(), name


def test_a_damaged_zstd_chunk_reports_an_error_and_leaves_its_neighbours_alone(kat):
    good = kat["chunk|tiled_u16_split"]
    src = kat["cin|tiled_u16_split"]
    bad = good.copy()
    at = good.tobytes().find(b"\x28\xb5\x2f\xfd")            # the first stream that is a zstd frame
    assert at > 32
    bad[at + 5:at + 40] ^= 0x5A
    bs = int(np.frombuffer(good[8:12].tobytes(), "<i4")[0])
    rc, status, out = E.decompress_batch([good.tobytes(), bad.tobytes(), good.tobytes()], [src.size] * 3, [bs] * 3)
    assert status[0] == 0 and status[2] == 0 and status[1] < 0
    assert out[0].tobytes() == src.tobytes() and out[2].tobytes() == src.tobytes()


def test_chunks_made_with_the_local_libzstd(golden_dir):
    """Beyond the committed vectors: every element size, split and unsplit, ragged last block -- made here with the system
    libzstd (skipped where there is none), decoded by the emulated kernels."""
    import ctypes as C
    import ctypes.util
    import sys
    name = ctypes.util.find_library("zstd")
    if not name:
        pytest.skip("no libzstd here")
    sys.path.insert(0, golden_dir)
    import make_zstd_golden as G
    from cimg import synth
    z = C.CDLL(name)
    z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t; z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    for dtype, fam in ((np.uint8, synth.natural_channel), (np.uint16, synth.tiled_channel), (np.float16, synth.natural_channel),
                       (np.float32, synth.tiled_channel), (np.float32, synth.natural_channel)):
        it = np.dtype(dtype).itemsize
        a = fam(dtype, 512, 256 // it + 3)
        src = np.ascontiguousarray(a).view(np.uint8).ravel()
        for clevel in (1, 5, 9):
            for filt in ("shuffle", "bitshuffle", "none"):
                chunk = G.frame(z, src, it, 32768, clevel, filt)
                rc, status, out = E.decompress_batch([chunk], [src.size], [32768])
                assert rc == 0 and status == [0], (dtype, clevel, filt, status)
                assert out[0].tobytes() == src.tobytes(), (dtype, clevel, filt)
    # other block sizes: 64 KiB (the kernel's LDS areas follow the batch's largest block), 4 KiB
    a = synth.natural_channel(np.uint16, 512, 300)
    src = np.ascontiguousarray(a).view(np.uint8).ravel()
    for bs in (65536, 4096):
        for clevel in (3, 9):
            chunk = G.frame(z, src, 2, bs, clevel)
            rc, status, out = E.decompress_batch([chunk], [src.size], [bs])
            assert rc == 0 and status == [0], (bs, clevel, status)
            assert out[0].tobytes() == src.tobytes(), (bs, clevel)


def test_zstd_chunks_through_the_reference_entry_point_on_the_mock_library(kat):
    """blosc2_decompress_ctx of tests/emu/libcimg_hip_mock.so (the C ABI served by the emulator): the call the reference makes
    per chunk (blosc2/wrapper.h:236-259) reads zstd chunks -- shim -> batch call -> general kernel -> zstd kernel."""
    import ctypes as C
    from cimg import hip
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libcimg_hip_mock.so")
    if not os.path.exists(path):
        pytest.skip("mock C ABI not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(path)
    L.blosc2_create_dctx.restype = C.c_void_p
    L.blosc2_create_dctx.argtypes = [hip.Blosc2DParams]
    L.blosc2_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    L.blosc2_free_ctx.argtypes = [C.c_void_p]
    dp = hip.Blosc2DParams()
    dp.nthreads = 1
    d = L.blosc2_create_dctx(dp)
    try:
        for n in kat["chunks"]:
            n = str(n)
            c = kat["chunk|" + n].copy()
            want = kat["cin|" + n]
            out = np.zeros(want.size, np.uint8)
            assert L.blosc2_decompress_ctx(d, c.ctypes.data, c.size, out.ctypes.data, out.size) == want.size, n
            assert out.tobytes() == want.tobytes(), n
            assert L.blosc2_decompress_ctx(d, c.ctypes.data, c.size, out.ctypes.data, out.size - 1) < 0
    finally:
        L.blosc2_free_ctx(d)
