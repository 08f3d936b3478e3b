"""Kernel LOGIC on CPU: the gfx950 kernel bodies (csrc/*_kernel.h) compiled for the host lane emulator
(tests/emu) must reproduce the oracle bit for bit -- compressed bytes and pixels.  These tests do not
replace the `-m gpu` parity tests; they make sure the windowed LZ4 search, the in-place LZ4 decode
and the layout walk are right before any GPU time is spent.
"""
import os

import numpy as np
import pytest

import _emu as E
import _oracle as O
from cimg import synth


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "lz4_kat.npz"))


@pytest.mark.parametrize("order", [0, 1, 2])
def test_wave_lz4_encoder_matches_liblz4(kat, order):
    """Every limited-output golden vector, under three orders of resolving colliding LDS writes."""
    E.set_write_order(order)
    try:
        n = 0
        for key in kat["cases"]:
            key = str(key)
            name, a, c = key.split("|")
            src = kat["in|" + name]
            cap = int(c[1:])
            if cap >= src.size + src.size // 255 + 16:
                continue                       # the wave encoder is the limited-output form blosc2 uses
            r, out, _ = E.lz4_encode(src, cap, int(a[1:]))
            assert r == int(kat["ret|" + key]), key
            assert out == kat["out|" + key].tobytes(), key
            n += 1
        assert n > 600
    finally:
        E.set_write_order(0)


def _dense_streams(seed, count):
    """Sequence-dense streams (what photographs shuffle into): tiny alphabets, slow ramps, sparse noise, periodic rows with
    mutations, smooth 2-D gradients -- the post-match probe hits on most sequences, searches of 2 .. 20 probes on the rest."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        n = int(rng.integers(13, 33000)) if i % 7 else int(rng.choice([13, 14, 64, 77, 4096, 16384, 32768]))
        kind = i % 6
        if kind == 0:
            s = rng.integers(0, int(rng.integers(2, 6)), n)
        elif kind == 1:
            s = np.cumsum(rng.integers(0, 2, n)) // int(rng.integers(1, 4))
        elif kind == 2:
            s = rng.integers(0, 256, n) * (rng.random(n) < rng.choice([0.05, 0.3]))
        elif kind == 3:
            row = rng.integers(0, 256, int(rng.integers(5, 300)))
            s = np.resize(row, n).copy()
            flips = rng.integers(0, n, max(1, n // int(rng.integers(20, 400))))
            s[flips] = rng.integers(0, 256, flips.size)
        elif kind == 4:
            w = int(rng.integers(16, 512))
            y, x = np.divmod(np.arange(n), w)
            s = (np.sin(x / 37.0) * 40 + np.cos(y / 11.0) * 30 + 128).astype(np.int64) + (rng.random(n) < 0.1) * rng.integers(0, 3, n)
        else:
            s = np.repeat(rng.integers(0, 256, n // 3 + 1), rng.integers(1, 9, n // 3 + 1))[:n]
            if s.size < n:
                s = np.resize(s, n)
        yield (s.astype(np.int64) & 255).astype(np.uint8)


@pytest.mark.parametrize("rt", [0, 1])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_dense_streams_match_liblz4_pinned_oracle(rt, order):
    """Both forms of the LZ4 encoder (LDS table with its zero-literal chain / head / windows; register table) against the oracle
    (pinned to liblz4 by tests/test_oracle_lz4.py) on sequence-dense streams: bytes, return value and `need`, acceleration 1 and 5,
    capacity = the stream (c-blosc2's call) and a capacity that cuts the stream off."""
    E.set_write_order(order)
    E.set_enc_rt(rt)
    try:
        for k, src in enumerate(_dense_streams(1000 + order, 70)):
            for accel in ((1, 5) if k % 5 == 0 else (1,)):
                want_r, want, want_need = O.lz4_compress(src, cap=src.size, accel=accel, want_need=True)
                r, out, need = E.lz4_encode(src, src.size, accel)
                assert r == want_r, (k, src.size, accel)
                assert out == want[:max(want_r, 0)], (k, src.size, accel)
                if want_r > 0:
                    assert need == want_need, (k, src.size, accel)
                    cut = max(1, want_need - 1 - (k % 9))             # one short of what the encoder needs: both say 0
                    assert E.lz4_encode(src, cut, accel)[0] == O.lz4_compress(src, cap=cut, accel=accel)[0] == 0, (k, cut)
    finally:
        E.set_write_order(0)
        E.set_enc_rt(0)


def test_noise_planes_match_the_oracle():
    """Noise planes go through the no-match walk of the encoder; one in forty has a table slot that three probes of a window
    share, one in eighty a real four-byte repeat that LZ4 finds -- both hand the plane to the full encoder.  Same verdict, same
    bytes as the oracle in every write order, also with repeats planted inside the first windows."""
    rng = np.random.default_rng(77)
    planes = [rng.integers(0, 256, 16384, dtype=np.uint8) for _ in range(240)]
    for k in range(40):                                   # near-noise: a few repeats planted, some inside the first windows
        p = rng.integers(0, 256, 16384, dtype=np.uint8)
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, 2000)); b = a + int(rng.integers(4, 400)); n = int(rng.integers(4, 24))
            p[b:b + n] = p[a:a + n]
        planes.append(p)
    try:
        for order in (0, 1, 2):
            E.set_write_order(order)
            for i, p in enumerate(planes):
                cap = p.size + p.size // 255 + 16 - 1 if i % 3 == 0 else p.size
                r, out, need = E.lz4_encode(p, cap)
                ro, oo, no = O.lz4_compress(p, cap, want_need=True)
                assert r == ro and out == oo and (r <= 0 or need == no), (order, i)
    finally:
        E.set_write_order(0)


def test_wave_lz4_need_matches_liblz4(kat):
    for key in (k for k in kat.files if k.startswith("need|")):
        _, name, a = key.split("|")
        src = kat["in|" + name]
        r, _, need = E.lz4_encode(src, src.size, int(a[1:]))
        if r > 0:
            assert need == int(kat[key]), key


def test_wave_lz4_decoder_inverts_golden_streams(kat):
    n = 0
    for key in kat["cases"]:
        key = str(key)
        if int(kat["ret|" + key]) <= 0:
            continue
        src = kat["in|" + key.split("|")[0]]
        rc, dec = E.lz4_decode(kat["out|" + key], src.size)
        assert rc == 0 and dec == src.tobytes(), key
        n += 1
    assert n > 400


def test_wave_lz4_decoder_rejects_damage():
    src = np.resize(np.arange(50, dtype=np.uint8), 4000)
    r, comp = O.lz4_compress(src)
    assert E.lz4_decode(comp, 4000)[0] == 0
    assert E.lz4_decode(comp[:-3], 4000)[0] < 0
    assert E.lz4_decode(comp, 3999)[0] < 0
    assert E.lz4_decode(comp, 4001)[0] < 0


def _check_batch(dtype, arr, chunk, blocksize=32768, destsize=None, order=0, filters=(0, 0, 0, 0, 0, 1), splitmode=3):
    it = np.dtype(dtype).itemsize
    raw = arr.view(np.uint8).ravel()
    sizes = [min(chunk, raw.size - o) for o in range(0, raw.size, chunk)]
    dsz = [chunk + 32 if destsize is None else destsize] * len(sizes)
    E.set_write_order(order)
    try:
        rc, cb, chunks = E.compress_batch(E.cparams(it, blocksize=blocksize, filters=filters, splitmode=splitmode), raw, sizes, dsz)
    finally:
        E.set_write_order(0)
    assert rc == 0
    po = O.cparams(it, blocksize=blocksize, filters=filters, splitmode=splitmode)
    off = 0
    for i, s in enumerate(sizes):
        r, c = O.compress(po, raw[off:off + s], destsize=dsz[i])
        assert cb[i] == r and chunks[i] == c, (np.dtype(dtype).name, i)
        off += s
    live = [(c, s) for c, s in zip(chunks, sizes) if len(c)]
    if live:
        rc, st, outs = E.decompress_batch([c for c, _ in live], [s for _, s in live],
                                          [O.cbuffer_sizes(c)[2] for c, _ in live], misalign=3)
        assert rc == 0 and not any(st)
        assert b"".join(o.tobytes() for o in outs) == b"".join(
            raw[sum(sizes[:i]):sum(sizes[:i + 1])].tobytes() for i, c in enumerate(chunks) if len(c))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float16, np.uint32, np.float32])
@pytest.mark.parametrize("family", ["tiled", "zero", "random", "natural"])
def test_chunk_pipeline_equals_oracle(dtype, family):
    arr = getattr(synth, family + "_channel")(dtype, 1024, 100)
    it = np.dtype(dtype).itemsize
    _check_batch(dtype, arr, 40000 // it * it)         # ragged: 1 full + leftover block per chunk, short last chunk


def test_chunk_pipeline_other_typesizes_and_blocks():
    rng = np.random.default_rng(3)
    a = (rng.integers(0, 40, 6000, dtype=np.uint64) * 0x0101010101).astype(np.uint64)      # typesize 8
    _check_batch(np.uint64, a, 16384, blocksize=4096)
    _check_batch(np.uint16, synth.natural_channel(np.uint16, 300, 41), 5000, blocksize=256)   # tiny blocks
    _check_batch(np.uint8, synth.natural_channel(np.uint8, 333, 77), 9999, blocksize=1000)     # odd sizes
    _check_batch(np.uint32, np.arange(50, dtype=np.uint32), 4096)                            # single tiny chunk
    _check_batch(np.uint8, np.arange(20, dtype=np.uint8), 4096)                              # < 32 bytes: memcpyed


def test_chunk_pipeline_destsize_rules_collision_orders():
    rng = np.random.default_rng(8)
    noisy = rng.integers(0, 65536, 20000, dtype=np.uint16)
    noisy[3000:9000] = 7
    for destsize in (40000 + 32, 39000, 36000, 30000, 20100, 200):
        _check_batch(np.uint16, noisy, 40000, blocksize=4096, destsize=destsize, order=destsize % 3)


def test_reference_known_answers_through_kernels():
    """iota / constant planes of the reference's OIIO-free tests."""
    for dt in (np.uint8, np.uint16, np.uint32, np.float32):
        _check_batch(dt, np.arange(4096).astype(dt), 256, blocksize=64)       # test_schunk.cpp:39-75
    for v in (255, 0, 199, 12):
        _check_batch(np.uint16, np.full(64 * 16, v, np.uint16), 768, blocksize=256)   # test_image.cpp chunk 768


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float16, np.uint32, np.float32, np.uint64])
def test_bitshuffle_filter_equals_oracle(dtype):
    """lz4 + bitshuffle (the config-3 extension): bit rows are one unsplit stream per block.  Sizes keep
    ne % 8 == 0 in every block (the layout confirmed against c-blosc 1.21) except the
    last case, which exercises the copied tail of the restated rule."""
    bs = (0, 0, 0, 0, 0, 2)
    it = np.dtype(dtype).itemsize
    fam = synth.natural_channel if it <= 4 else (lambda d, w, h: (np.arange(w * h, dtype=np.uint64) // 7 * 0x0101).astype(np.uint64))
    arr = fam(dtype, 1024, 96)
    _check_batch(dtype, arr, 65536, filters=bs)                               # two 32 KiB blocks per chunk
    _check_batch(dtype, arr, 40960, blocksize=8192, filters=bs, order=1)      # short last chunk, still multiples of 8 elements
    _check_batch(dtype, synth.tiled_channel(dtype, 512, 40) if it <= 4 else arr, 16384, blocksize=4096, filters=bs, order=2)
    _check_batch(dtype, arr.ravel()[:5003], 4096 * it, blocksize=1024 * it, filters=bs)   # ragged tail: ne % 8 != 0 in the leftover block


def test_lean_decode_kernel_takes_the_single_coded_plane_blocks():
    """The lean decode kernel (one LZ4-coded plane per block, the rest stored raw or as run tokens) must produce the
    same pixels as the general kernel, must actually take the blocks it is meant for, and must leave the others."""
    E.lean_blocks()
    tiled = synth.tiled_channel(np.float16, 1024, 128)               # high byte plane codes, low byte plane is stored raw
    _check_batch(np.float16, tiled, 65536)
    assert E.lean_blocks() == tiled.nbytes // 32768                   # every block
    _check_batch(np.float32, synth.tiled_channel(np.float32, 512, 128), 65536)   # typesize 4: 8 KiB planes
    f32 = E.lean_blocks()
    zero = np.zeros((64, 1024), np.uint16)
    zero[::7, ::5] = 3                                                # low plane codes, high plane is a run token
    _check_batch(np.uint16, zero, 65536)
    assert E.lean_blocks() > 0
    # 16-bit pixels below 256 with a noisy low byte: low plane stored raw, high plane a zero run, NO coded plane (a round-3 build
    # requested the stored plane into registers here and un-shuffled it against whatever the last block left in LDS)
    rng = np.random.default_rng(11)
    quiet = ((np.arange(300 * 200).reshape(200, 300) // 37 % 251) + rng.integers(0, 3, (200, 300))).astype(np.uint16)
    _check_batch(np.uint16, quiet, 120000)
    _check_batch(np.uint16, rng.integers(0, 256, (64, 1024)).astype(np.uint16), 65536)
    assert E.lean_blocks() > 0
    _check_batch(np.uint8, synth.natural_channel(np.uint8, 512, 64), 16384)      # typesize 1: never lean
    assert E.lean_blocks() == 0
    _check_batch(np.uint16, synth.natural_channel(np.uint16, 1024, 64), 40000)   # ragged: leftover blocks stay general
    E.lean_blocks()
    E.set_lean(False)
    try:
        _check_batch(np.float16, tiled, 65536)                        # the general kernel alone still does everything
        assert E.lean_blocks() == 0
    finally:
        E.set_lean(True)
    assert f32 >= 0


def test_split_modes_follow_the_oracle():
    """blosc2 split modes (1 always, 2 never, 3 auto, 4 forward-compatible): forced splitting of unfiltered data and
    unsplit shuffled blocks are layouts the reference never asks for but a chunk from elsewhere may have."""
    a = synth.natural_channel(np.uint16, 512, 80)
    f = synth.tiled_channel(np.float32, 256, 64)
    _check_batch(np.uint16, a, 40000, splitmode=2)                                  # never: shuffled 32 KiB blocks as one stream
    _check_batch(np.float32, f, 32768, blocksize=8192, splitmode=2, order=1)
    _check_batch(np.uint16, a, 40000, splitmode=1, filters=(0, 0, 0, 0, 0, 0))      # always: planes are plain slices
    _check_batch(np.uint16, a, 40000, splitmode=4, order=2)
    _check_batch(np.uint8, synth.natural_channel(np.uint8, 512, 64), 20000, splitmode=1)


def test_unusual_typesizes():
    """Element sizes the Python surface never produces but the C ABI accepts: 3 (packed RGB), 16 (the largest that
    still splits into byte planes), 20 (too wide to split: one shuffled stream per block)."""
    rng = np.random.default_rng(21)
    for ts in (3, 16, 20):
        n = 3000
        base = (np.arange(n)[:, None] // 9 * (np.arange(ts)[None, :] + 1)).astype(np.uint8)
        base[:, 0] = rng.integers(0, 256, n)                               # one noisy byte plane
        raw = np.ascontiguousarray(base).ravel()
        sizes = [raw.size]
        for blocksize in (4096 // ts * ts, 960):
            rc, cb, chunks = E.compress_batch(E.cparams(ts, blocksize=blocksize), raw, sizes, [raw.size + 32])
            assert rc == 0
            r, c = O.compress(O.cparams(ts, blocksize=blocksize), raw, destsize=raw.size + 32)
            assert cb[0] == r and chunks[0] == c, (ts, blocksize)
            rc, st, outs = E.decompress_batch([chunks[0]], sizes, [O.cbuffer_sizes(c)[2]], misalign=1)
            assert rc == 0 and not any(st) and outs[0].tobytes() == raw.tobytes()


@pytest.mark.parametrize("compcode", [1, 0])
def test_randomized_geometries_against_the_oracle(compcode):
    """The seeded differential test of tests/test_gpu_parity.py on the emulated kernels (fewer rounds), cycling
    through the three colliding-write orders; lz4 (compcode 1) and blosclz (0, every level)."""
    from test_gpu_parity import _mixed_data
    rng = np.random.default_rng(20260102 + compcode)
    for it in range(40):
        ts = int(rng.choice([1, 2, 2, 4, 4, 8, 3]))
        clevel = 9 if compcode == 1 else int(rng.integers(1, 10))
        blocksize = int(rng.choice([256, 1024, 4096, 8192, 32768])) // ts * ts
        chunk = min(int(rng.integers(1, 6)) * blocksize + (int(rng.integers(0, blocksize)) // ts * ts if rng.random() < 0.4 else 0), 120000) // ts * ts or ts
        total = max(chunk * int(rng.integers(0, 3)) + int(rng.integers(1, chunk + 1)) // ts * ts, ts)
        raw = _mixed_data(rng, total, ts)
        filt = int(rng.choice([0, 1, 1, 1, 2]))
        dest = chunk + 32 if rng.random() < 0.7 else max(40, int(chunk * rng.uniform(0.3, 1.0)))
        sizes = [min(chunk, total - o) for o in range(0, total, chunk)]
        E.set_write_order(it % 3)
        try:
            rc, cb, chunks = E.compress_batch(E.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=compcode, filters=(0, 0, 0, 0, 0, filt)), raw, sizes, [dest] * len(sizes))
        finally:
            E.set_write_order(0)
        assert rc == 0
        po = O.cparams(ts, clevel=clevel, blocksize=blocksize, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
        off = 0
        for i, s in enumerate(sizes):
            r, want = O.compress(po, raw[off:off + s], destsize=dest)
            assert cb[i] == r and chunks[i] == want, (compcode, it, i, ts, blocksize, chunk, clevel, filt, dest)
            off += s
        live = [(c, s) for c, s in zip(chunks, sizes) if len(c)]
        if live:
            rc, st, outs = E.decompress_batch([c for c, _ in live], [s for _, s in live], [O.cbuffer_sizes(c)[2] for c, _ in live], misalign=it % 4)
            assert rc == 0 and not any(st)
            want = b"".join(raw[sum(sizes[:i]):sum(sizes[:i + 1])].tobytes() for i, c in enumerate(chunks) if len(c))
            assert b"".join(o.tobytes() for o in outs) == want


def test_encode_work_item_shapes_give_the_same_bytes():
    """The split encode launch deals blocks out whole, plane by plane, or -- on a small batch -- the full rounds whole and the
    last partial round plane by plane (encode_kernel.h: encode_items / item_place).  Whatever the queue looks like, the chunks
    are the oracle's, for 2- and 4-byte types, lz4 and blosclz, with a leftover block in the batch."""
    import _oracle as O
    from cimg import synth
    try:
        for dtype, ts in ((np.float16, 2), (np.float32, 4)):
            raw = synth.tiled_channel(dtype, 1024, 130).view(np.uint8).ravel()
            sizes = [131072, 131072, raw.size - 262144] if raw.size > 262144 else [131072, raw.size - 131072]
            for code in (1, 0):
                want = []
                off = 0
                for n in sizes:
                    want.append(O.compress(O.cparams(ts, compcode=code), raw[off:off + n])[1])
                    off += n
                for mode in (0, 1, 2):
                    E.set_block_items(mode)
                    rc, cb, chunks = E.compress_batch(E.cparams(ts, compcode=code), raw, sizes, [n + 32 for n in sizes])
                    assert rc == 0 and chunks == want, (dtype, code, mode)
    finally:
        E.set_block_items(1)


def test_own_codec_with_an_unreadable_filter_is_an_error_not_a_zstd_retry():
    """An lz4 chunk whose header names a filter the kernels do not implement (delta, trunc_prec, a second filter) is
    BLOSC2_ERROR_CODEC_SUPPORT (-7).  The engine retries chunks of codec format 4 (zstd) with cimg_decode_zstd; a round-2 build
    cleared EVERY -7 before that launch and such chunks ended with status 0 and no pixels."""
    a = synth.tiled_channel(np.float16, 512, 64)
    raw = a.view(np.uint8).ravel()
    rc, cb, chunks = E.compress_batch(E.cparams(2), raw, [raw.size], [raw.size + 32])
    good = chunks[0]
    for slot, code in ((20, 3), (21, 3), (21, 4), (19, 1)):
        bad = bytearray(good)
        bad[slot] = code
        rc, status, outs = E.decompress_batch([bytes(bad), good], [raw.size] * 2, [32768] * 2)
        assert status == [-7, 0], (slot, code, status)
        assert outs[1].tobytes() == raw.tobytes()
    bad = bytearray(good)
    bad[2] = (bad[2] & 0x1F) | (3 << 5)                                # codec format 3 (zlib): nobody reads it
    rc, status, outs = E.decompress_batch([bytes(bad)], [raw.size], [32768])
    assert status == [-7]


def test_chunks_assembled_inside_the_encode_launch_equal_the_two_kernel_path():
    """Round 3: chunks whose streams all belong to ONE encode launch are laid out and copied into place by the waves of that
    launch (encode_kernel.h: encode_account / encode_emit_own); the others -- memcpyed up front, or split planes plus an unsplit
    leftover block -- by cimg_layout_chunks / cimg_emit_blocks behind it.  Both give the oracle's bytes (every other test of this
    file runs the in-launch form where it applies); here the forms are compared on a batch that mixes all kinds of chunk, and the
    harness's own checks (every closer puts its stream count back to zero, every assembled chunk is marked ready, the queue head
    wraps around 2^32) run with it."""
    L = E.lib()
    L.emu_last_folded.restype = E.C.c_int
    a = synth.tiled_channel(np.float16, 1024, 200)                       # 400 KiB
    raw = a.view(np.uint8).ravel()
    cases = [
        ([131072] * 3 + [raw.size - 3 * 131072], 131072 + 32, 4, 4),     # three full chunks + a remainder chunk that is one (split) block
        ([100000] * 4, 100000 + 32, 0, 4),                               # every chunk has a leftover block: split planes + one unsplit stream
        ([131072, 20, 131072], 131072 + 32, 2, 3),                       # a 20-byte chunk in the middle: memcpyed up front
    ]
    for sizes, dest, want_folded, n in cases:
        src = np.concatenate([raw[:s] for s in sizes])
        got = {}
        for fold in (1, 0):
            L.emu_set_fold(fold)
            try:
                rc, cb, chunks = E.compress_batch(E.cparams(2), src, sizes, [dest] * len(sizes))
            finally:
                L.emu_set_fold(1)
            assert rc == 0
            assert L.emu_last_folded() == (want_folded if fold else 0), (sizes, fold, L.emu_last_folded())
            got[fold] = chunks
        assert got[0] == got[1]
        off = 0
        for c, s in zip(got[1], sizes):
            r, want = O.compress(O.cparams(2), src[off:off + s], destsize=dest)
            assert c == want
            off += s


def plane_kind_blocks(ts, rng):
    """One 32 KiB block per combination of plane kinds -- coded (compressible), stored (noise), run (one value) -- for every byte
    plane of a `ts`-byte type: 9 blocks for 2-byte, 81 for 4-byte elements.  Returns the pixels as bytes (blocks back to back)."""
    import itertools
    ne = 32768 // ts
    blocks = []
    for kinds in itertools.product("crk", repeat=ts):
        planes = []
        for k in kinds:
            if k == "c":
                p = (np.arange(ne) // int(rng.integers(3, 70)) % 200).astype(np.uint8)
                p[rng.integers(0, ne, 40)] ^= 0x55
            elif k == "r":
                p = rng.integers(0, 256, ne, dtype=np.uint8)
            else:
                p = np.full(ne, int(rng.integers(0, 256)), np.uint8)
            planes.append(p)
        blocks.append(np.stack(planes, axis=1).ravel())          # element i = (plane 0 byte, plane 1 byte, ...)
    return np.concatenate(blocks)


@pytest.mark.parametrize("compcode", [1, 0])
def test_every_combination_of_plane_kinds_decodes(compcode):
    """The lean decode kernel takes blocks by the KINDS of their planes (at most one coded, the others stored or run tokens) and
    has a register path for the commonest shape; this walks all 9 / 81 combinations for 2- and 4-byte elements, in an order that
    puts every shape behind every other kind of LDS content, for LZ4 and BloscLZ.  Bytes against the oracle, pixels back."""
    rng = np.random.default_rng(77 + compcode)
    for ts in (2, 4):
        raw = plane_kind_blocks(ts, rng)
        sizes = [raw.size]
        rc, cb, chunks = E.compress_batch(E.cparams(ts, compcode=compcode), raw, sizes, [raw.size + 32])
        assert rc == 0
        r, want = O.compress(O.cparams(ts, compcode=compcode), raw, destsize=raw.size + 32)
        assert chunks[0] == want
        for shape in (1, 2, 5):
            E.lib().emu_set_lean_shape(shape)
            try:
                rc, st, outs = E.decompress_batch(chunks, sizes, [32768])
            finally:
                E.lib().emu_set_lean_shape(-1)
            assert rc == 0 and not any(st)
            bad = np.nonzero(outs[0] != raw)[0]
            assert bad.size == 0, (ts, compcode, shape, "first wrong byte %d = block %d" % (bad[0], bad[0] // 32768))
