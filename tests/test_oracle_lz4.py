"""Pin the oracle's LZ4 block layer (oracle/lz4_block.c) against liblz4 1.9.3 known answers.

The reference reaches LZ4 only through c-blosc2's per-stream LZ4_compress_fast / LZ4_decompress_safe
(SURVEY.md section 8a N4); tests/golden/lz4_kat.npz was produced by tests/golden/make_lz4_golden.py.
"""
import os

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "lz4_kat.npz"))


def test_encoder_matches_liblz4_bytes(kat):
    n_zero = 0
    for key in kat["cases"]:
        key = str(key)
        name, a, c = key.split("|")
        src = kat["in|" + name]
        ret = int(kat["ret|" + key])
        r, out = O.lz4_compress(src, cap=int(c[1:]), accel=int(a[1:]))
        assert r == ret, key
        assert out == kat["out|" + key].tobytes(), key
        n_zero += ret == 0
    assert len(kat["cases"]) >= 900 and n_zero > 100      # the "does not fit -> 0" rule is exercised


def test_min_capacity_matches_liblz4(kat):
    keys = [k for k in kat.files if k.startswith("need|")]
    assert len(keys) >= 90
    for key in keys:
        _, name, a = key.split("|")
        src = kat["in|" + name]
        n = src.size
        r, out, need = O.lz4_compress(src, cap=n + n // 255 + 16, accel=int(a[1:]), want_need=True)
        assert need == int(kat[key]), key
        assert O.lz4_compress(src, cap=need, accel=int(a[1:]))[1] == out
        assert O.lz4_compress(src, cap=need - 1, accel=int(a[1:]))[0] == 0


def test_decoder_inverts_every_vector(kat):
    for key in kat["cases"]:
        key = str(key)
        ret = int(kat["ret|" + key])
        if ret <= 0:
            continue
        src = kat["in|" + key.split("|")[0]]
        r, dec = O.lz4_decompress(kat["out|" + key], src.size)
        assert r == src.size and dec == src.tobytes(), key


def test_decoder_rejects_damage():
    src = np.resize(np.arange(50, dtype=np.uint8), 4000)
    r, comp = O.lz4_compress(src)
    assert 0 < r < 4000
    assert O.lz4_decompress(comp[:-3], 4000)[0] < 0           # truncated
    assert O.lz4_decompress(comp, 3999)[0] < 0                # output too small
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0xFF
    rr, dec = O.lz4_decompress(bytes(bad), 4000)
    assert rr < 0 or dec != src.tobytes()


def test_tiny_inputs_are_literal_only():
    for n in range(1, 13):
        src = np.full(n, 7, np.uint8)
        r, out = O.lz4_compress(src, cap=n + 16)
        assert r == n + 1 and out[0] == n << 4 and out[1:] == src.tobytes()
