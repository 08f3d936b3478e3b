"""The reference's OIIO-free C++ test cases (tests/cpp/host_mirror_test.cpp) against the host mirror,
linked to the emulator-backed mock of the C ABI (CPU) or to libcimg_hip.so (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")


def _build_and_run(libdir, lib, exe):
    out = os.path.join(ROOT, "tests", "cpp", exe)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "compressed-image_amd", "include"), SRC, "-o", out,
                           "-L", libdir, "-l" + lib, "-Wl,-rpath," + libdir, "-pthread"])
    res = subprocess.run([out], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "0 failures" in res.stdout


def test_host_mirror_on_emulator():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    _build_and_run(os.path.join(ROOT, "tests", "emu"), "cimg_hip_mock", "host_mirror_test_mock")


@pytest.mark.gpu
def test_host_mirror_on_gpu():
    _build_and_run(os.path.join(ROOT, "compressed-image_amd"), "cimg_hip", "host_mirror_test_gpu")
