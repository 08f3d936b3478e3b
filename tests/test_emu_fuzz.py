"""Memory safety and termination of the kernel logic on damaged input, under AddressSanitizer + UBSan.

GPU sanitizers are not available on the pool, so the emulated kernels (the same csrc/*.h sources, 64-lane host
emulator) are built with -fsanitize=address,undefined and the workgroup's LDS is modelled as a host buffer of
EXACTLY the launch size (tests/emu/fuzz_main.cpp).  Thousands of damaged chunks and LZ4 streams are decoded; any
access outside what a real launch owns is a sanitizer report, any non-terminating loop a timeout."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_damaged_input_never_leaves_the_allocation_or_hangs():
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "fuzz_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(emu, "fuzz_asan"), "4000"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fuzz ok" in r.stdout
