"""Chunk layer of the oracle (oracle/chunk.c): framing, filters, destsize rules, and every
decompressed-pixel known answer of the reference's OIIO-free tests (SURVEY.md sections 4 and 8c).

No reference test pins compressed bytes, so the framing assertions below restate the published
c-blosc2 chunk format; they are the "self-generated, unverified vs
c-blosc2" layer of the parity story ("Parity status").
"""
import os
import struct

import numpy as np
import pytest

import _oracle as O
from cimg import synth

DTYPES = [np.uint8, np.uint16, np.uint32, np.float32, np.float16]


def hdr(chunk):
    nbytes, blocksize, cbytes = struct.unpack_from("<iii", chunk, 4)
    return dict(version=chunk[0], versionlz=chunk[1], flags=chunk[2], typesize=chunk[3], nbytes=nbytes,
                blocksize=blocksize, cbytes=cbytes, filters=tuple(chunk[16:22]), compcode=chunk[22],
                blosc2_flags=chunk[31])


# ---- filters -------------------------------------------------------------------------------------
@pytest.mark.parametrize("ts", [1, 2, 4, 8, 3])
def test_shuffle_known_answer(ts):
    rng = np.random.default_rng(ts)
    for nbytes in (32768, 4096 + 5, ts * 11 + (ts - 1)):
        a = rng.integers(0, 256, nbytes, dtype=np.uint8)
        ne = nbytes // ts
        want = np.concatenate([a[:ne * ts].reshape(ne, ts).T.ravel(), a[ne * ts:]])
        got = O.shuffle(ts, a)
        assert np.array_equal(got, want)
        assert np.array_equal(O.unshuffle(ts, got), a)


@pytest.mark.parametrize("ts", [1, 2, 4])
def test_bitshuffle_known_answer(ts):
    rng = np.random.default_rng(10 + ts)
    for ne in (8192, 64, 100):          # 100 % 8 != 0: tail elements are copied verbatim
        a = rng.integers(0, 256, ne * ts, dtype=np.uint8)
        ne8 = ne - ne % 8
        bs = a[:ne8 * ts].reshape(ne8, ts).T                        # byte rows
        bits = np.unpackbits(bs[:, :, None], axis=2, bitorder="little")   # [ts, ne8, 8]
        rows = np.packbits(bits.transpose(0, 2, 1), axis=2, bitorder="little")  # [ts, 8, ne8/8]
        want = np.concatenate([rows.ravel(), a[ne8 * ts:]])
        got = O.bitshuffle(ts, a)
        assert np.array_equal(got, want)
        assert np.array_equal(O.bitunshuffle(ts, got), a)


# ---- framing ---------------------------------------------------------------------------------------
def test_header_and_bstarts_of_a_regular_chunk():
    a = synth.tiled_channel(np.float16, 4096, 16)        # 128 KiB = 4 blocks x 2 streams
    p = O.cparams(2)
    r, c = O.compress(p, a)
    h = hdr(c)
    assert r == len(c) == h["cbytes"]
    assert (h["version"], h["versionlz"], h["typesize"]) == (5, 1, 2)
    assert h["flags"] == 0x01 | 0x04 | (1 << 5)          # extended header marker + LZ4 format, split
    assert h["filters"] == (0, 0, 0, 0, 0, 1) and h["compcode"] == 1 and h["blosc2_flags"] == 0
    assert (h["nbytes"], h["blocksize"]) == (a.nbytes, 32768)
    bstarts = struct.unpack_from("<4i", c, 32)
    assert bstarts[0] == 32 + 16
    pos = bstarts[0]
    raw = a.view(np.uint8).ravel()
    for j in range(4):
        assert pos == bstarts[j]
        sh = O.shuffle(2, raw[j * 32768:(j + 1) * 32768])
        for s in range(2):
            (cs,) = struct.unpack_from("<i", c, pos)
            pos += 4
            stream = sh[s * 16384:(s + 1) * 16384]
            rr, out = O.lz4_compress(stream, cap=16384, accel=1)
            if rr in (0, 16384):
                assert cs == 16384 and c[pos:pos + cs] == stream.tobytes()
            else:
                assert cs == rr and c[pos:pos + cs] == out
            pos += cs
    assert pos == r
    assert O.cbuffer_sizes(c) == (a.nbytes, r, 32768)


def test_run_tokens_and_special_zero_chunk():
    p = O.cparams(1)
    r, c = O.compress(p, np.zeros(65536, np.uint8))
    assert r == 32 and hdr(c)["blosc2_flags"] == 1 << 4 and hdr(c)["cbytes"] == 32
    assert not O.decompress(c)[1].any()
    r, c = O.compress(p, np.full(65536, 255, np.uint8))
    assert r == 32 + 2 * 4 + 2 * 5
    assert struct.unpack_from("<i", c, 40)[0] == -255 and c[44] == 1
    assert (O.decompress(c)[1] == 255).all()
    # u16 constant 0x00FF: the high-byte stream is a zero run (no token), the low-byte stream a 0xFF run
    p2 = O.cparams(2)
    r, c = O.compress(p2, np.full(16384, 0x00FF, np.uint16))
    assert r == 32 + 4 + (4 + 1) + 4
    assert (O.decompress(c)[1].view(np.uint16) == 0x00FF).all()


def test_split_rule_and_leftover_block():
    p = O.cparams(4, blocksize=256)
    g = O.geometry(p, 1000)                       # 3 full blocks + 232-byte leftover
    assert (g.blocksize, g.nblocks, g.leftover, g.split) == (256, 4, 232, 1)
    assert g.nstreams_total == 3 * 4 + 1          # the leftover block is never split
    assert O.geometry(O.cparams(4, blocksize=64), 4096).split == 0       # 64/4 < 32 elements
    assert O.geometry(O.cparams(2, compcode=O.LZ4HC), 65536).split == 0
    assert O.geometry(O.cparams(2, filters=(0, 0, 0, 0, 0, O.BITSHUFFLE)), 65536).split == 0
    assert O.geometry(O.cparams(2, clevel=0), 65536).memcpyed == 1
    assert O.geometry(O.cparams(2), 31).memcpyed == 1
    a = np.arange(250, dtype=np.uint32)
    r, c = O.compress(p, a)
    assert np.array_equal(O.decompress(c)[1].view(np.uint32), a)


def test_incompressible_full_chunk_becomes_memcpyed_and_remainder_stays_framed():
    # destsize is always the *nominal* chunk size + 32 (schunk.h:73)
    rng = np.random.default_rng(5)
    p = O.cparams(2, blocksize=4096)
    full = rng.integers(0, 65536, 32768, dtype=np.uint16)       # 64 KiB "chunk"
    r, c = O.compress(p, full, destsize=65536 + 32)
    assert r == 65536 + 32 and hdr(c)["flags"] & 0x02
    assert c[32:] == full.tobytes()
    rem = full[:8192]                                            # 16 KiB remainder in the same buffer
    r, c = O.compress(p, rem, destsize=65536 + 32)
    assert r > rem.nbytes + 32 and not hdr(c)["flags"] & 0x02
    assert np.array_equal(O.decompress(c)[1].view(np.uint16), rem)
    # dest too small even for the memcpyed form -> 0
    assert O.compress(p, full, destsize=65536 + 31)[0] == 0


@pytest.mark.parametrize("ts", [1, 2, 4])
def test_two_phase_layout_equals_serial_under_every_destsize(ts):
    """The (size, need) records + serial walk (GPU structure) reproduce the inline budget clipping."""
    rng = np.random.default_rng(100 + ts)
    n = 6 * 1024
    base = rng.integers(0, 256, n, dtype=np.uint8)
    base[1024:3072] = np.resize(base[:64], 2048)             # some compressible blocks
    base[4096:4608] = 7
    p = O.cparams(ts, blocksize=1024)
    r_full, c_full = O.compress(p, base, destsize=n + 4096)
    for destsize in list(range(32, 200, 7)) + list(range(r_full - 1200, r_full + 40, 1)) + [n + 32, n + 31]:
        r1, c1 = O.compress(p, base, destsize=destsize)
        r2, c2 = O.compress(p, base, destsize=destsize, two_phase=True, nthreads=2)
        assert (r1, c1) == (r2, c2), destsize
        if r1 > 0:
            assert O.decompress(c1)[1].tobytes() == base.tobytes()


# ---- reference known answers (decompressed pixels) -----------------------------------------------------
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.float32])
def test_ref_schunk_iota_roundtrip(dtype):
    """test/src/test_schunk.cpp:39-75: iota(4096), block 64, chunk 256, lz4 level 9."""
    data = np.arange(4096).astype(dtype)
    it = data.dtype.itemsize
    p = O.cparams(it, clevel=9, blocksize=64)
    raw = data.view(np.uint8)
    nchunks = 4096 * it // 256
    out = []
    for k in range(nchunks):
        r, c = O.compress(p, raw[k * 256:(k + 1) * 256], destsize=256 + 32)
        assert r > 0 and O.cbuffer_sizes(c)[0] == 256            # chunk(0).size() == 256 / sizeof(T)
        out.append(O.decompress(c)[1])
    assert np.array_equal(np.concatenate(out).view(dtype), data)


def test_ref_channel_small_buffers():
    """test/src/test_channel.cpp:46-69: 50-byte iota (one tiny chunk) and 8192-byte iota, 2 chunks of 4096."""
    a = np.arange(50, dtype=np.uint8)
    r, c = O.compress(O.cparams(1), a, destsize=4194304 + 32)
    assert np.array_equal(O.decompress(c)[1], a)
    b = (np.arange(8192) & 255).astype(np.uint8)
    p = O.cparams(1, blocksize=128)
    for k in range(2):
        r, c = O.compress(p, b[k * 4096:(k + 1) * 4096], destsize=4096 + 32)
        assert np.array_equal(O.decompress(c)[1], b[k * 4096:(k + 1) * 4096])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("value", [0, 255, 199, 12, 90, 100, 25])
def test_ref_constant_channels(dtype, value):
    """test_image.cpp:552-859 / python tests: constant planes incl. chunk 768 of a 64x16 image."""
    a = np.full(64 * 16, value).astype(dtype)
    it = a.dtype.itemsize
    p = O.cparams(it, blocksize=256)
    raw = a.view(np.uint8)
    chunk = 768 // it // 64 * 64 * it or 64 * it
    got = []
    for off in range(0, raw.size, chunk):
        r, c = O.compress(p, raw[off:off + chunk], destsize=chunk + 32)
        assert r > 0
        got.append(O.decompress(c)[1])
    assert np.array_equal(np.concatenate(got).view(a.dtype), a)


@pytest.mark.parametrize("dtype", DTYPES)
def test_synthetic_families_roundtrip(dtype):
    it = np.dtype(dtype).itemsize
    p = O.cparams(it)
    for arr in (synth.tiled_channel(dtype, 1024, 70), synth.zero_channel(dtype, 1024, 70),
                synth.random_channel(dtype, 1024, 70), synth.natural_channel(dtype, 1024, 70)):
        r, c = O.compress(p, arr, destsize=4194304 + 32)
        assert r > 0
        assert O.decompress(c)[1].tobytes() == arr.tobytes()


def test_decoder_flags_corruption():
    a = synth.natural_channel(np.uint16, 1024, 32)
    r, c = O.compress(O.cparams(2), a)
    bad = bytearray(c)
    bad[0] = 9
    with pytest.raises(ValueError):
        O.cbuffer_sizes(bytes(bad))
    bad = bytearray(c)
    struct.pack_into("<i", bad, 32, len(c) + 100)           # bstart outside the chunk
    assert O.decompress(bytes(bad), a.nbytes)[0] < 0


def test_lz4hc_chunks_decode(golden_dir):
    """lz4hc is codec format 1 = plain LZ4 blocks: chunks coded by liblz4's LZ4_compress_HC (tests/golden/
    make_lz4hc_golden.py) decode through the ordinary path (enums.h:18-24 lists lz4hc as a valid codec)."""
    kat = np.load(os.path.join(golden_dir, "lz4hc_kat.npz"))
    for name in kat["cases"]:
        chunk, src = kat["chunk|" + str(name)], kat["in|" + str(name)]
        assert chunk[22] == O.LZ4HC and (chunk[2] >> 5) == 1 and chunk[2] & 0x10
        n, pix = O.decompress(chunk)
        assert n == src.size and pix.tobytes() == src.tobytes(), name
