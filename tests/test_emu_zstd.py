"""The zstd frame decoder (csrc/zstd_decode.h: the read side of enums::codec::zstd, enums.h:18-24) on the host, against
frames libzstd made (tests/golden/make_zstd_golden.py): the answer of every vector is its input."""
import os

import numpy as np
import pytest

import _emu as E
import _oracle as O
from cimg import synth


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


@pytest.fixture(params=["planned", "planned_3_lanes", "walkers_decode_sequences", "fused", "plans_overflow"], autouse=True)
def zstd_read_path(request):
    """Every test of this file in every form of the read path: the walk + lane decoder + replay launches (engine.hip:
    decompress_finish; eight blocks a wave, and three -- workgroups that end inside a chunk), the walkers decoding the sequences
    themselves (CIMG_ZSTD_LANES=0), the fused kernels alone (CIMG_ZSTD_FUSED=1), and plans so small (256 bytes of records, of
    literals) that most blocks are refused by the walk and decoded by the fused kernels behind the launches."""
    E.set_zstd_plan({"fused": 0, "plans_overflow": 256}.get(request.param, -1))
    E.set_zstd_lanes({"planned_3_lanes": 3, "walkers_decode_sequences": 0}.get(request.param, 8))
    E.zstd_refused()
    yield request.param
    E.set_zstd_plan(-1)
    E.set_zstd_lanes(8)


def test_plans_that_do_not_fit_are_counted_and_their_blocks_still_decode(kat, zstd_read_path):
    name = "natural_f32_split"
    chunk = kat["chunk|" + name]; src = kat["cin|" + name]
    bs = int(np.frombuffer(chunk[8:12].tobytes(), "<i4")[0])
    E.zstd_refused()
    rc, status, out = E.decompress_batch([chunk.tobytes()], [src.size], [bs])
    assert rc == 0 and status == [0] and out[0].tobytes() == src.tobytes()
    refused = E.zstd_refused()
    assert (refused > 0) == (zstd_read_path == "plans_overflow"), (zstd_read_path, refused)


def test_a_block_nobody_decoded_fails_the_batch(kat, zstd_read_path):
    """a block whose plan the walk refused used to hang on ONE counter -- had that count been lost, the fused kernel would
    not have run and the call would have reported success over pixels that were never written.  Every such block now carries its own
    pending word until cimg_decode_zstd has taken it: with the fused pass suppressed, the batch FAILS."""
    if zstd_read_path != "plans_overflow":
        pytest.skip("only plans that do not fit leave blocks to the fused kernel")
    name = "natural_f32_split"
    chunk = kat["chunk|" + name]; src = kat["cin|" + name]
    bs = int(np.frombuffer(chunk[8:12].tobytes(), "<i4")[0])
    E.set_zstd_lose_fused(1)
    try:
        rc, status, out = E.decompress_batch([chunk.tobytes()], [src.size], [bs])
        assert rc != 0, "blocks left pending went unnoticed"
    finally:
        E.set_zstd_lose_fused(0)
    rc, status, out = E.decompress_batch([chunk.tobytes()], [src.size], [bs])
    assert rc == 0 and out[0].tobytes() == src.tobytes()


def test_split_and_unsplit_zstd_chunks_in_one_batch(kat):
    """Split chunks go to the two-waves-per-block launch, chunks of one stream per block to the one-wave launch (engine.hip:
    decompress_finish; each launch reads its own kind only): one batch with both, and LZ4 chunks between them."""
    names = ["tiled_u16_split", "tiled_f16_unsplit", "natural_f32_split", "u8_small_blocks", "mixed_u16", "natural_f32_split"]
    chunks = [kat["chunk|" + n].tobytes() for n in names]
    srcs = [kat["cin|" + n] for n in names]
    bss = [int(np.frombuffer(c[8:12], "<i4")[0]) for c in chunks]
    lz4 = O.compress(O.cparams(2, clevel=5, blocksize=32768), srcs[0])[1]
    chunks.insert(2, lz4); srcs.insert(2, srcs[0]); bss.insert(2, 32768)
    rc, status, out = E.decompress_batch(chunks, [s.size for s in srcs], bss)
    assert rc == 0 and status == [0] * len(chunks), (rc, status)
    for o, s_, n in zip(out, srcs, range(len(srcs))):
        assert o.tobytes() == s_.tobytes(), n


def test_the_lowest_damaged_stream_of_a_split_block_is_the_one_reported(kat):
    """Two waves share the streams of a block; what the chunk reports is what decoding the streams in order reports: damage in a
    LATER stream only (a stream header that points outside the chunk = ERR_READ_BUFFER) against damage in the first frame (ERR_DATA
    class) AND the later header."""
    good = kat["chunk|natural_f32_split"]
    src = kat["cin|natural_f32_split"]
    bs = int(np.frombuffer(good[8:12].tobytes(), "<i4")[0])
    raw = good.tobytes()
    nblocks = -(-src.size // bs)
    b0 = int(np.frombuffer(raw[32:36], "<i4")[0])
    cs0 = int(np.frombuffer(raw[b0:b0 + 4], "<i4")[0])
    assert 0 < cs0 < bs // 4
    late = good.copy()
    late[b0 + 4 + cs0:b0 + 8 + cs0] = np.frombuffer(np.int32(1 << 24).tobytes(), np.uint8)      # stream 1's size word
    rc, st_late, _ = E.decompress_batch([late.tobytes()], [src.size], [bs])
    both = late.copy()
    both[b0 + 4 + 6:b0 + 4 + 40] ^= 0x5A                                                          # inside stream 0's frame
    rc, st_both, _ = E.decompress_batch([both.tobytes()], [src.size], [bs])
    first = good.copy()
    first[b0 + 4 + 6:b0 + 4 + 40] ^= 0x5A
    rc, st_first, _ = E.decompress_batch([first.tobytes()], [src.size], [bs])
    assert st_late[0] < 0 and st_first[0] < 0
    assert st_both == st_first, (st_both, st_first, st_late)
    assert nblocks >= 1


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


def test_chunks_made_with_the_local_libzstd(golden_dir, zstd_read_path):
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
    # other block sizes: 128 and 64 KiB (the kernel's LDS follows the batch's largest block; unsplit streams above 64 KiB), 4 KiB
    a = synth.natural_channel(np.uint16, 512, 300)
    src = np.ascontiguousarray(a).view(np.uint8).ravel()
    # (147456 = 144 KiB: a stream longer than zstd's largest block -- frames of two blocks, the second with treeless literals or
    # repeated tables and the history of repeat offsets carried over; through the lane decoders that is two jobs per frame)
    for bs in (147456, 131072, 65536, 4096):
        for clevel in (3, 9):
            chunk = G.frame(z, src, 2, bs, clevel)
            rc, status, out = E.decompress_batch([chunk], [src.size], [bs])
            if bs == 147456 and zstd_read_path == "plans_overflow":
                # (blocks the walk refuses go to cimg_decode_zstd, whose LDS holds output AND tables: 144 KiB blocks do not fit there --
                # the chunk, and only the chunk, says so)
                assert rc == 0 and status == [-7], (bs, clevel, status)
                continue
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


def test_zstd_encoder_frames_decode_with_libzstd_and_with_the_own_decoder():
    """csrc/zstd_encode.h on the host lane emulator: one stream -> one zstd frame (raw literals, predefined FSE tables, offsets never
    as repeat codes).  FORMAT-VALID, NOT BYTE-PINNED: every frame must decode to its input with the box's libzstd (skipped where
    there is none) AND with csrc/zstd_decode.h; a stream that does not shrink gives 0 (c-blosc2 then stores it raw).  Sizes land
    between LZ4's and libzstd's."""
    import ctypes as C
    import _oracle as O
    L = E.lib()
    L.emu_zstd_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.emu_zstd_encode.restype = C.c_int
    have = O.zstd_available()
    OL = O.lib()
    OL.orc_zstd_decompress_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    OL.orc_zstd_decompress_stream.restype = C.c_int
    rng = np.random.default_rng(5)
    plane = lambda a, ts, s: np.ascontiguousarray(a.view(np.uint8).reshape(-1, ts)[:, s])
    cases = {
        "tiled_hi": plane(synth.tiled_channel(np.float16, 4096, 4), 2, 1),
        "tiled_lo_noise": plane(synth.tiled_channel(np.float16, 4096, 4), 2, 0),
        "natural_hi": plane(synth.natural_channel(np.uint16, 4096, 4), 2, 1),
        "natural_lo": plane(synth.natural_channel(np.uint16, 4096, 4), 2, 0),
        "text": np.frombuffer((b"the quick brown fox jumps over the lazy dog, " * 400)[:16000], np.uint8),
        "runs": np.concatenate([np.zeros(5000, np.uint8), rng.integers(0, 4, 3000, dtype=np.uint8), np.full(8000, 7, np.uint8)]),
        "unsplit_32k_f32": synth.tiled_channel(np.float32, 4096, 2).view(np.uint8).ravel()[:32768],
        "period_10": np.tile(np.arange(10, dtype=np.uint8), 20),
        "far_matches_60k": np.tile(rng.integers(0, 256, 3000, dtype=np.uint8), 20)[:60000],
        "max_65535": np.tile(rng.integers(0, 7, 257, dtype=np.uint8), 256)[:65535],
        "max_65536": np.tile(rng.integers(0, 7, 259, dtype=np.uint8), 256)[:65536],
        "max_65536_one_match": np.zeros(65536, np.uint8),            # the longest match and, in the mirror case, the longest literal run
        "max_65536_literals_then_match": np.concatenate([rng.integers(0, 256, 65000, dtype=np.uint8), np.zeros(536, np.uint8)]),
        "max_65536_far_offset": (lambda head: np.concatenate([head, rng.integers(0, 256, 64936, dtype=np.uint8), head]))(rng.integers(0, 256, 300, dtype=np.uint8)),
        "one_long_literal_run_then_match": np.concatenate([rng.integers(0, 256, 20000, dtype=np.uint8), np.zeros(3000, np.uint8)]),
        "tiny_40": np.zeros(40, np.uint8) + np.arange(40, dtype=np.uint8) % 3,
        "random": rng.integers(0, 256, 16384, dtype=np.uint8),
    }
    for k in range(40):                                           # many sequences with long literal / match lengths of every code
        n = int(rng.integers(64, 40000))
        pieces, left = [], n
        while left > 0:
            m = int(min(left, rng.choice([1, 3, 15, 16, 40, 70, 130, 300, 2000])))
            pieces.append(rng.integers(0, 256, m, dtype=np.uint8) if rng.random() < 0.5 else np.full(m, rng.integers(0, 256), np.uint8))
            left -= m
        cases["mix%d" % k] = np.concatenate(pieces)
    coded = 0
    for name, src in cases.items():
        n = src.size
        dst = np.full(n + 64, 0xEE, np.uint8)
        r = L.emu_zstd_encode(src.ctypes.data, n, dst.ctypes.data, n)
        assert 0 <= r < n, name
        assert (dst[n:] == 0xEE).all(), name                      # nothing is written past the stream's budget
        if r == 0:
            continue
        coded += 1
        assert bytes(dst[:4]) == b"\x28\xb5\x2f\xfd"
        r2, own = E.zstd_decode(dst[:r].tobytes(), n)
        assert r2 == n and own == src.tobytes(), name
        if have:
            out = np.zeros(n, np.uint8)
            assert OL.orc_zstd_decompress_stream(dst.ctypes.data, r, out.ctypes.data, n) == n, name
            assert out.tobytes() == src.tobytes(), name
    assert coded >= 40
    assert L.emu_zstd_encode(cases["random"].ctypes.data, 16384, np.zeros(16500, np.uint8).ctypes.data, 16384) == 0   # noise: no frame


def test_zstd_and_lz4hc_chunks_from_the_emulated_kernels():
    """Whole chunks with codec::zstd / codec::lz4hc through the emulated encode kernels + in-launch assembly: the checker's chunk
    layer (libzstd / the LZ4 block decoder) and the emulated decode kernels both return the pixels; lz4hc chunks equal the checker's
    twin (LZ4 fast at acceleration 1 under compcode 2) byte for byte; zstd splits planes up to clevel 5 like c-blosc2."""
    import _oracle as O
    for codec in (O.ZSTD, O.LZ4HC):
        for clevel in (9, 5, 1):
            for dtype, arr in ((np.float16, synth.tiled_channel(np.float16, 1024, 100)), (np.uint16, synth.natural_channel(np.uint16, 512, 130)),
                               (np.float32, synth.tiled_channel(np.float32, 512, 70)), (np.uint8, synth.natural_channel(np.uint8, 700, 99)),
                               # (the last 16 KiB chunk of this one builds a frame for its low plane that does NOT pay: the plane must
                               # still be intact for the raw store -- a round-3 build staged the bit stream over it)
                               (np.uint16, synth.natural_channel(np.uint16, 1024, 200))):
                it = np.dtype(dtype).itemsize
                raw = np.ascontiguousarray(arr).view(np.uint8).ravel()
                chunk = 131072 if raw.size == 409600 else 65536
                sizes = [min(chunk, raw.size - o) for o in range(0, raw.size, chunk)]
                rc, cb, chunks = E.compress_batch(E.cparams(it, clevel=clevel, compcode=codec), raw, sizes, [chunk + 32] * len(sizes))
                assert rc == 0
                off = 0
                for c, n in zip(chunks, sizes):
                    assert c[22] == codec
                    if codec == O.ZSTD and not O.zstd_available():
                        off += n
                        continue
                    r, px = O.decompress(c)
                    assert r == n and px.tobytes() == raw[off:off + n].tobytes(), (codec, clevel, np.dtype(dtype).name)
                    if codec == O.LZ4HC:
                        assert c == O.compress(O.cparams(it, clevel=clevel, compcode=O.LZ4HC), raw[off:off + n], destsize=chunk + 32)[1]
                    off += n
                rc, status, outs = E.decompress_batch(chunks, sizes, [O.cbuffer_sizes(c)[2] for c in chunks])
                assert rc == 0 and not any(status)
                assert b"".join(o.tobytes() for o in outs) == raw.tobytes()


def test_randomized_geometries_zstd_and_lz4hc_on_the_emulated_kernels():
    """The GPU test of the same name (tests/test_gpu_parity.py) at sizes the lane emulator finishes in seconds: random element size,
    block size, chunk size, level, filter, dest capacity and data make-up through the emulated encode kernels; the checker
    (libzstd / the LZ4 block decoder under the oracle's chunk layer) and the emulated decode kernels both return the input."""
    import _oracle as O
    from test_gpu_parity import _mixed_data
    rng = np.random.default_rng(20260304 + int(os.environ.get("CIMG_TEST_SEED", "0")))
    have_zstd = O.zstd_available()
    for it in range(int(os.environ.get("CIMG_TEST_ROUNDS", "200"))):
        codec = O.ZSTD if it % 3 else O.LZ4HC
        ts = int(rng.choice([1, 2, 2, 4, 4, 8, 3]))
        blocksize = int(rng.choice([256, 1024, 4096, 8192, 32768, 65536])) // ts * ts
        nchunks = int(rng.integers(1, 3))
        chunk = int(rng.integers(1, 4)) * blocksize + (int(rng.integers(0, blocksize)) // ts * ts if rng.random() < 0.4 else 0)
        chunk = min(chunk, 100000) // ts * ts or ts
        total = max(chunk * (nchunks - 1) + int(rng.integers(1, chunk + 1)) // ts * ts, ts)
        raw = _mixed_data(rng, total, ts)
        clevel = int(rng.choice([1, 5, 9, 9]))
        filt = int(rng.choice([0, 1, 1, 1, 2]))
        dest = chunk + 32 if rng.random() < 0.7 else max(40, int(chunk * rng.uniform(0.3, 1.0)))
        sizes = [min(chunk, total - o) for o in range(0, total, chunk)]
        what = (it, codec, ts, blocksize, chunk, clevel, filt, dest)
        rc, cb, chunks = E.compress_batch(E.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=codec, filters=(0, 0, 0, 0, 0, filt)), raw, sizes, [dest] * len(sizes))
        assert rc == 0, what
        off = 0
        live, live_sizes = [], []
        for c, n in zip(chunks, sizes):
            if c:
                if codec == O.LZ4HC or have_zstd:
                    r, px = O.decompress(c)
                    assert r == n and px.tobytes() == raw[off:off + n].tobytes(), what
                if codec == O.LZ4HC:
                    assert c == O.compress(O.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=O.LZ4HC, filters=(0, 0, 0, 0, 0, filt)), raw[off:off + n], destsize=dest)[1], what
                live.append((c, n, off))
            off += n
        if live:
            rc, status, outs = E.decompress_batch([c for c, _, _ in live], [n for _, n, _ in live], [O.cbuffer_sizes(c)[2] for c, _, _ in live])
            assert rc == 0 and not any(status), (what, status)
            for o, (_, n, at) in zip(outs, live):
                assert o.tobytes() == raw[at:at + n].tobytes(), what


def test_sequence_bit_streams_longer_than_a_lanes_lds(golden_dir, zstd_read_path):
    """cimg_zstd_seq keeps 4 KiB of a job's bit stream in the lane's LDS and brings the next 4 KiB over when the reader reaches the
    bottom (zstd_seq_kernel.h: the refill step).  The emulator built once more with 256 bytes instead -- every stream of the chunk
    tests is refilled many times -- runs the chunk tests of this file again."""
    import subprocess
    import sys
    if zstd_read_path != "planned":
        pytest.skip("once is enough: the child runs the forms with lanes itself")
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "compressed-image_amd", "csrc")
    lib = os.path.join(here, "emu", "libcimg_emu_stream256.so")
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-w", "-fno-strict-aliasing", "-DCIMG_ZSTD_SEQ_STREAM=256",
                           "-I", csrc, os.path.join(here, "emu", "emu.cpp"), "-o", lib])
    env = dict(os.environ, CIMG_EMU_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "(blosc2_zstd_chunks_decode or local_libzstd or split_and_unsplit or damaged) and (planned or lanes)"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
