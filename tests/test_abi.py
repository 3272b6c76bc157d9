"""CPU test: the C-ABI library loads here (no GPU) and exports every symbol include/vpp_amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vpp_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vpp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from vpp_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_layout_helper_matches_python():
    from vpp_amd import capi, image as vi
    lib = capi.lib()
    for (nr, nc, es, b, al) in [(1080, 1920, 4, 0, 16), (2160, 3840, 3, 2, 16), (2160, 3840, 3, 2, 32), (271, 481, 8, 3, 32), (5, 7, 1, 18, 256)]:
        pitch, size, first = ctypes.c_int32(), ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.vpp_image_layout(nr, nc, es, b, al, ctypes.byref(pitch), ctypes.byref(size), ctypes.byref(first)) == 0
        assert (pitch.value, size.value, first.value) == vi.layout(nr, nc, es, b, al)
