"""Loader for the C-ABI shared library (include/vpp_amd.h).  There is NO CPU fallback: if the HIP library is
missing this raises, and every product entry point goes through it."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvpp_amd.so")
_lib = None

OK, ERR_INVALID_ARG, ERR_BORDER_TOO_SMALL, ERR_HIP, ERR_UNSUPPORTED, ERR_CAPACITY = range(6)


class VppError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"vpp_amd status {status}: {msg}")
        self.status = status


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vpp_last_error.restype = ctypes.c_char_p
        _lib.vpp_version.restype = ctypes.c_char_p
    return _lib


def check(status):
    if status != OK:
        raise VppError(status, lib().vpp_last_error().decode())
    return status


def stream_ptr():
    """Raw hipStream_t of torch's current stream (so torch allocations and our kernels are ordered)."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
