"""Interleaved pixels -> planes (reference: image_algo::deinterleave, compressed/image_algo.h:84-111, the step between reading
scanlines and compressing them, image.h:1880).  The kernel (csrc/deinterleave_kernel.h) on the host lane emulator and on the
GPU against numpy: every element size, channel counts 1..9, pixel counts around the tile and the 16-byte unit boundaries."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "compressed-image_amd")]

CASES = [(nch, ts, npix) for nch in (1, 2, 3, 4, 5, 9) for ts in (1, 2, 4, 8)
         for npix in (0, 1, 7, 15, 16, 17, 255, 1000, 4096, 4099)]


def _expected(raw, nch, ts):
    npix = raw.size // (nch * ts)
    return raw.reshape(npix, nch, ts).transpose(1, 0, 2).reshape(nch, npix * ts)


def test_emulated_kernel_against_numpy():
    import _emu
    rng = np.random.default_rng(5)
    for nch, ts, npix in CASES:
        raw = rng.integers(0, 256, npix * nch * ts, dtype=np.uint8)
        got = _emu.deinterleave(raw, nch, ts)
        assert got.shape == (nch, npix * ts) and np.array_equal(got, _expected(raw, nch, ts)), (nch, ts, npix)
    big = rng.integers(0, 256, 50001 * 4 * 2, dtype=np.uint8)              # many tiles, a ragged last one
    assert np.array_equal(_emu.deinterleave(big, 4, 2), _expected(big, 4, 2))
    with pytest.raises(ValueError):
        _emu.deinterleave(np.zeros(3 * 2100 * 8, np.uint8), 2100, 8)        # a pixel group wider than a tile


@pytest.mark.gpu
def test_gpu_kernel_against_numpy_and_the_fused_producer_path():
    from cimg import hip
    import _oracle as O
    eng = hip.Engine(0)
    try:
        rng = np.random.default_rng(6)
        for nch, ts, npix in CASES + [(4, 2, 1 << 20), (3, 4, 333333)]:
            if npix == 0:
                continue
            raw = rng.integers(0, 256, npix * nch * ts, dtype=np.uint8)
            stride = (npix * ts + 15) & ~15
            d_in, d_out = eng.alloc(raw.size), eng.alloc(stride * nch)
            d_in.upload(raw)
            eng.deinterleave_device(d_in.ptr, nch, ts, npix, d_out.ptr, stride)
            eng.synchronize()
            out = d_out.download()
            got = np.stack([out[c * stride:c * stride + npix * ts] for c in range(nch)])
            assert np.array_equal(got, _expected(raw, nch, ts)), (nch, ts, npix)
            d_in.free(); d_out.free()
        with pytest.raises(hip.CodecError):
            eng.deinterleave_device(1 << 20, 2, 3, 100, 1 << 21, 1600)      # element size 3
    finally:
        eng.close()
