# Mutation fuzz of the zstd frame decoder (csrc/zstd_decode.h) and of the zstd chunk path of the emulated kernels under
# AddressSanitizer + UBSan, on the host (GPU sanitizers are not available on this pool).  Builds tests/emu/libcimg_emu_asan.so and
# re-executes itself with libasan preloaded.  Usage: python tools/zstd_fuzz_asan.py [mutations per frame]
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "emu", "libcimg_emu_asan.so")
if os.environ.get("CIMG_FUZZ_CHILD") != "1":
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-strict-aliasing", "-w", "-I", os.path.join(ROOT, "compressed-image_amd", "csrc"), os.path.join(ROOT, "tests", "emu", "emu.cpp"), "-o", LIB])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, CIMG_FUZZ_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
import ctypes as C
import numpy as np
L = C.CDLL(LIB)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kat = np.load(os.path.join(ROOT, "tests", "golden", "zstd_kat.npz"))
rng = np.random.default_rng(2026)
p = lambda a: a.ctypes.data_as(C.c_void_p)
n_ok = n_err = 0
for key in kat["frames"]:
    key = str(key)
    src = kat["in|" + key.split("|")[0]]
    fr = kat["frame|" + key]
    if fr.size < 8: continue
    out = np.zeros(src.size + 16, np.uint8)
    for it in range(per):
        bad = fr.copy()
        kind = it % 4
        if kind == 0:   bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: bad[int(rng.integers(4, bad.size)):] = rng.integers(0, 256, 1, dtype=np.uint8)[0]
        elif kind == 2: bad = bad[:int(rng.integers(1, bad.size))].copy()
        else:
            for _ in range(4): bad[int(rng.integers(4, bad.size))] = int(rng.integers(0, 256))
        cap = src.size if it % 3 else int(rng.integers(0, src.size + 1))
        r = L.emu_zstd_decode(p(bad), int(bad.size), p(out), int(cap))
        assert r <= cap
        assert not (-999 <= r <= -990), ("the decoder's forms disagree", key, it, r)      # in place / staged / serial / walked and replayed
        if r < 0: n_err += 1
        else: n_ok += 1
# chunks through the emulated kernels
for name in kat["chunks"]:
    name = str(name)
    chunk = kat["chunk|" + name]; src = kat["cin|" + name]
    bs = int(np.frombuffer(chunk[8:12].tobytes(), "<i4")[0])
    for it in range(per // 4):
        bad = chunk.copy()
        for _ in range(1 + it % 3): bad[int(rng.integers(32, bad.size))] ^= 1 << int(rng.integers(0, 8))
        comp = np.zeros(bad.size + 64, np.uint8); comp[:bad.size] = bad
        raw = np.zeros(src.size + 64, np.uint8)
        off = np.zeros(1, np.int64); nb = np.array([src.size], np.int32); bsz = np.array([bs], np.int32); st = np.zeros(1, np.int32)
        L.emu_set_zstd_plan((-1, 0, 256)[it % 3])       # walk + replay launches / fused kernels / plans that overflow
        L.emu_decompress_batch(1, p(comp), p(off), p(nb), p(bsz), p(raw), p(off), p(st))
print("zstd fuzz under ASAN/UBSan: %d mutated frames decoded, %d rejected, chunks x %d -- no finding" % (n_ok, n_err, per // 4))
