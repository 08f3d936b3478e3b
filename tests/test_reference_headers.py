"""The reference's OWN host headers over this repository's C ABI.

Build-container only: /root/reference does not travel to the GPU box, and nothing of it is copied -- its headers are named on
the include path where they lie.  tests/cpp/ref_headers_test.cpp includes the reference's blosc2/wrapper.h, schunk.h,
lazyschunk.h (+ schunk_mixin.h through them) and, in a second build, blosc2/typedefs.h + iterators/iterator.h, compiles them
against include/blosc2.h -- this repository's re-declaration of the eleven c-blosc2 symbols the reference binds -- and links
tests/emu/libcimg_hip_mock.so (the C ABI served by the host lane emulator).  It then replays the checks of the reference's
test/src/test_schunk.cpp:19-75 and the channel's chunk loop on the iterator.  <format>, which this image's libstdc++ 11 lacks, is
served by the {fmt} headers the image ships with torch (tests/cpp/compat/format).

What does NOT compile here, and why: compressed/channel.h stops at `#include "nlohmann/json.hpp"` (channel.h:12 -- a library the
image does not have; not stubbed), and with it image.h.  The seam itself -- every call the reference makes into c-blosc2 -- sits
below that line and is exercised here.  This is a test of the boundary, not an oracle: the bytes behind it are the emulator's.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/compressed_image/include"


def _fmt_include():
    try:
        import torch
    except ImportError:
        return None
    inc = os.path.join(os.path.dirname(torch.__file__), "include")
    return inc if os.path.exists(os.path.join(inc, "fmt", "format.h")) else None


@pytest.mark.parametrize("with_iterator", [False, True], ids=["schunk_layer", "iterator_layer"])
def test_reference_headers_run_over_the_c_abi(tmp_path, with_iterator):
    if not os.path.exists(os.path.join(REF_INC, "compressed", "blosc2", "wrapper.h")):
        pytest.skip("/root/reference is not on this machine (it never travels to the GPU box)")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    fmt_inc = _fmt_include()
    if fmt_inc is None:
        pytest.skip("no {fmt} headers in this image to serve <format>")
    mock = os.path.join(ROOT, "tests", "emu", "libcimg_hip_mock.so")
    if not os.path.exists(mock):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    exe = str(tmp_path / "ref_headers_test")
    cmd = ["g++", "-std=c++20", "-O1", "-I", REF_INC, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp", "compat"),
           "-I", fmt_inc, os.path.join(ROOT, "tests", "cpp", "ref_headers_test.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "tests", "emu"), "-lcimg_hip_mock", "-Wl,-rpath," + os.path.join(ROOT, "tests", "emu")]
    if with_iterator:
        cmd.insert(3, "-DREF_WITH_ITERATOR")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "the reference's headers no longer compile against include/blosc2.h:\n" + r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
