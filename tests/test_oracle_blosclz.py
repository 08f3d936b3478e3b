"""oracle/blosclz.c against the BloscLZ 2.3.0 payloads of c-blosc 1.21.0 (tests/golden/blosclz_kat.npz).

This is the pin of the blosclz restatement: compressed bytes of every vector (incl. every "gave up / does not
fit -> stored raw" decision) equal what the system library emitted.  It pins to c-blosc1's blosclz; c-blosc2
vendors a later release of the same codec (oracle/blosclz.c header).
"""
import os

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "blosclz_kat.npz"))


def test_library_versions(kat):
    assert str(kat["blosc_version"]) == "1.21.0" and str(kat["blosclz_version"]) == "2.3.0"


def test_compressed_bytes_match_blosclz_230(kat):
    coded = raw = 0
    for name in kat["cases"]:
        fam, n, clevel = str(name).rsplit("|", 2)
        n, clevel = int(n), int(clevel)
        src = kat[f"in|{fam}|{n}"]
        want = kat["out|" + str(name)].tobytes()
        r, got = O.blosclz_compress(src, clevel=clevel, cap=n)
        if want:
            assert r == len(want) and got == want, name
            d, pix = O.blosclz_decompress(want, n)
            assert d == n and pix == src.tobytes(), name
            coded += 1
        else:
            assert r == 0 or r == n, name          # c-blosc stores the stream raw in both cases
            raw += 1
    assert coded >= 100 and raw >= 50


def test_need_is_the_smallest_budget(kat):
    """need = max(66, size + 1): the call succeeds with cap = need and fails with cap = need - 1, same bytes."""
    seen = 0
    for name in kat["cases"]:
        fam, n, clevel = str(name).rsplit("|", 2)
        want = kat["out|" + str(name)].tobytes()
        if not want or int(n) > 16384:
            continue
        src = kat[f"in|{fam}|{n}"]
        r, got, need = O.blosclz_compress(src, clevel=int(clevel), cap=int(n), want_need=True)
        assert need == max(66, r + 1)
        r2, got2 = O.blosclz_compress(src, clevel=int(clevel), cap=need)
        r3, _ = O.blosclz_compress(src, clevel=int(clevel), cap=need - 1)
        assert (r2, got2) == (r, got) and r3 == 0, name
        seen += 1
    assert seen >= 40


def test_small_and_degenerate_streams():
    assert O.blosclz_compress(np.zeros(15, np.uint8), cap=200)[0] == 0        # length < 16
    assert O.blosclz_compress(np.zeros(100, np.uint8), cap=65)[0] == 0        # maxout < 66
    r, c = O.blosclz_compress(np.zeros(100, np.uint8), cap=66)
    assert r > 0 and O.blosclz_decompress(c, 100) == (100, bytes(100))
    assert O.blosclz_decompress(b"", 10)[0] == 0
    # a match needs at least one byte after its distance byte (the next control byte): truncated -> 0
    lit = bytes([0x03 | 0x20, 1, 2, 3, 4])
    assert O.blosclz_decompress(lit + bytes([0x20, 0x00]), 16)[0] == 0
    assert O.blosclz_decompress(lit + bytes([0x20, 0x00, 0x00, 9]), 16) == (8, bytes([1, 2, 3, 4, 4, 4, 4, 9]))


def test_chunk_roundtrip_blosclz_split_and_unsplit():
    """blosclz through the chunk layer: AUTO_SPLIT splits shuffled blocks exactly as for lz4."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compressed-image_amd"))
    from cimg import synth
    for dtype, filt in ((np.uint16, O.SHUFFLE), (np.uint8, O.SHUFFLE), (np.uint16, O.BITSHUFFLE), (np.float32, O.SHUFFLE)):
        a = synth.tiled_channel(dtype, 1024, 70).view(np.uint8).ravel()
        ts = np.dtype(dtype).itemsize
        p = O.cparams(ts, clevel=9, compcode=O.BLOSCLZ, filters=(0, 0, 0, 0, 0, filt))
        r, c = O.compress(p, a, destsize=a.size + 32)
        r2, c2 = O.compress(p, a, destsize=a.size + 32, two_phase=True, nthreads=2)
        assert r > 0 and (r, c) == (r2, c2)
        assert (c[2] >> 5) == 0                                       # codec format 0 in the header flags
        assert bool(c[2] & 0x10) == (filt == O.BITSHUFFLE and ts > 1 or False) or ts == 1 or filt == O.SHUFFLE
        d, pix = O.decompress(c)
        assert d == a.size and pix.tobytes() == a.tobytes()
