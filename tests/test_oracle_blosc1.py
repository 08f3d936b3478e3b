"""Structural cross-check of the oracle against c-blosc 1.21 output (tests/golden/blosc1_kat.npz).

c-blosc 1 is not the reference's codec; what is compared is everything beneath the frame that the
Blosc lineage shares (see tests/golden/make_blosc1_golden.py): shuffle layout, split rule, per-stream
LZ4 call, raw fallback, bstarts.  The oracle's own primitives rebuild every stream payload.
"""
import os
import struct

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "blosc1_kat.npz"))


def test_stream_payloads_match_cblosc1(kat):
    checked = raw = 0
    for name in kat["cases"]:
        name = str(name)
        src = kat["in|" + name]
        frame = kat["out|" + name].tobytes()
        ts, clevel, _ = (int(v) for v in kat["par|" + name])
        version, versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
        nbytes, blocksize, cbytes = struct.unpack_from("<iii", frame, 4)
        assert version == 2 and typesize == ts and nbytes == src.size and cbytes == len(frame)
        assert not flags & 0x2                     # not memcpyed (dest had slack)
        dont_split = bool(flags & 0x10)
        shuffled = bool(flags & 0x1)
        nblocks = -(-nbytes // blocksize)
        bstarts = struct.unpack_from(f"<{nblocks}i", frame, 16)
        for j in range(nblocks):
            blk = src[j * blocksize:(j + 1) * blocksize]
            leftover = blk.size < blocksize
            f = O.shuffle(ts, blk) if shuffled else blk
            nstreams = 1 if (dont_split or leftover) else ts
            ne = blk.size // nstreams
            pos = bstarts[j]
            for s in range(nstreams):
                (cs,) = struct.unpack_from("<i", frame, pos)
                pos += 4
                stream = f[s * ne:(s + 1) * ne]
                r, out = O.lz4_compress(stream, cap=ne, accel=10 - clevel)
                if r == 0 or r == ne:
                    assert cs == ne and frame[pos:pos + cs] == stream.tobytes(), (name, j, s)
                    raw += 1
                else:
                    assert cs == r and frame[pos:pos + cs] == out, (name, j, s)
                pos += cs
                checked += 1
            if j + 1 < nblocks:
                assert pos == bstarts[j + 1]
            else:
                assert pos == cbytes
    assert checked >= 20 and raw >= 3
