"""The C-ABI library loads and exports every symbol include/*.h declares (no compute: no GPU here)."""
import os
import re

from cimg import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return {n for n in names if n.startswith(("cimg_", "blosc2_")) or n in ("register_filters", "print_error")}


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    declared = _declared("cimg_hip.h") | _declared("blosc2.h")
    assert len(declared) >= 33, declared
    assert declared == set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_header_only_entry_points_work_without_a_gpu():
    lib = hip.load()
    assert lib.print_error(-11) == b"Invalid value in header"
    assert lib.cimg_kernel_name(hip.K_DECODE) == b"cimg_decode_blocks"
    p = hip.cparams(2)
    assert (p.typesize, p.clevel, p.blocksize, p.compcode, p.splitmode, tuple(p.filters)) == (2, 9, 32768, 1, 3, (0, 0, 0, 0, 0, 1))
    import struct
    hdr = bytes([5, 1, 0x25, 2]) + struct.pack("<iii", 4096, 1024, 999) + bytes(16)
    assert hip.cbuffer_sizes(hdr) == (4096, 999, 1024)
