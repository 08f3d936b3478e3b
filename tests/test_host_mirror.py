"""The reference's OIIO-free C++ test cases (tests/cpp/host_mirror_test.cpp) against the host mirror,
linked to the emulator-backed mock of the C ABI (CPU) or to libcimg_hip.so (GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")


def _build_and_run(libdir, lib, exe):
    out = os.path.join(ROOT, "tests", "cpp", exe)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "compressed-image_amd", "include"), SRC, "-o", out,
                           "-L", libdir, "-l" + lib, "-Wl,-rpath," + libdir, "-pthread"])
    dump = out + ".iterator_chunks.bin"
    res = subprocess.run([out], capture_output=True, text=True, timeout=600, env=dict(os.environ, CIMG_TEST_DUMP=dump))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "0 failures" in res.stdout
    _iterator_chunks_against_oracle(dump)


def _iterator_chunks_against_oracle(dump):
    """The chunks a double-buffered iterator pass left in a channel (written by the C++ test) are, byte for byte, what
    the oracle makes of the same pixels (the modify path compared with the oracle, not only with
    pixel known-answers)."""
    import numpy as np
    import _oracle
    raw = open(dump, "rb").read()
    os.remove(dump)
    typesize, blocksize, chunk_bytes, nchunks, w, h = np.frombuffer(raw[:24], np.uint32).tolist()
    pixels = np.frombuffer(raw[24:24 + w * h * typesize], np.uint8)
    assert nchunks * chunk_bytes == pixels.size
    pos = 24 + pixels.size
    p = _oracle.cparams(typesize, clevel=9, blocksize=blocksize, compcode=_oracle.LZ4)
    for k in range(nchunks):
        n = int(np.frombuffer(raw[pos:pos + 4], np.uint32)[0])
        got = raw[pos + 4:pos + 4 + n]
        pos += 4 + n
        rc, want = _oracle.compress(p, pixels[k * chunk_bytes:(k + 1) * chunk_bytes])
        assert rc == n and want == got, "chunk %d of the iterator pass differs from the oracle" % k
    assert pos == len(raw)


def test_host_mirror_on_emulator():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    _build_and_run(os.path.join(ROOT, "tests", "emu"), "cimg_hip_mock", "host_mirror_test_mock")


@pytest.mark.gpu
def test_host_mirror_on_gpu():
    _build_and_run(os.path.join(ROOT, "compressed-image_amd"), "cimg_hip", "host_mirror_test_gpu")
