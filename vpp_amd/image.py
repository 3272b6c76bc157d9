"""Pitched image buffers with the exact layout of vpp's ``imageNd::allocate``
(reference: vpp/core/imageNd.hpp:151-196) on the host (numpy) and on the device (a torch uint8 tensor used
purely as an HBM allocation).  ``desc`` is the C-ABI ``vpp_image_desc`` (include/vpp_amd.h)."""
import ctypes
import numpy as np

U8, I8, U16, I16, I32, U32, F32 = range(7)
_NP = {U8: np.uint8, I8: np.int8, U16: np.uint16, I16: np.int16, I32: np.int32, U32: np.uint32, F32: np.float32}
DEFAULT_ALIGN = 32  # VPP_DEFAULT_IMAGE_ALIGNMENT with __AVX2__ (imageNd.hpp:10-18)


class ImageDesc(ctypes.Structure):
    _fields_ = [("first_pixel", ctypes.c_void_p), ("nrows", ctypes.c_int32), ("ncols", ctypes.c_int32),
                ("pitch", ctypes.c_int32), ("border", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("channels", ctypes.c_int32)]


def dtype_size(dtype):
    return np.dtype(_NP[dtype]).itemsize


def layout(nrows, ncols, elem_bytes, border=0, align=DEFAULT_ALIGN):
    """(pitch, alloc_bytes, first_pixel_offset) — imageNd.hpp:151-196."""
    border_size = border * elem_bytes
    border_padding = 0
    if border_size % align:
        border_padding = align - border_size % align
        border_size += border_padding
    pitch = ncols * elem_bytes + 2 * border_size
    if pitch % align:
        pitch += align - pitch % align
    size = (nrows + 2 * border) * pitch
    first = border_padding + border * pitch + border * elem_bytes
    return pitch, size, first


class _ImageBase:
    def _setup(self, nrows, ncols, dtype, channels, border, align):
        self.nrows, self.ncols, self.dtype, self.channels, self.border, self.align = nrows, ncols, dtype, channels, border, align
        self.elem_bytes = dtype_size(dtype) * channels
        self.pitch, self.alloc_bytes, self.first_offset = layout(nrows, ncols, self.elem_bytes, border, align)

    def _desc(self, base_ptr):
        return ImageDesc(base_ptr + self.first_offset, self.nrows, self.ncols, self.pitch, self.border, self.dtype, self.channels)

    def _view(self, raw, with_border):
        """ndarray view (rows, cols, channels) over `raw` (1-D uint8 ndarray of alloc_bytes)."""
        b = self.border if with_border else 0
        off = self.first_offset - b * self.pitch - b * self.elem_bytes
        comp = dtype_size(self.dtype)
        return np.ndarray(shape=(self.nrows + 2 * b, self.ncols + 2 * b, self.channels), dtype=_NP[self.dtype],
                          buffer=raw, offset=off, strides=(self.pitch, self.elem_bytes, comp))


class HostImage(_ImageBase):
    """image2d<vector<dtype,channels>> in host memory, zero-initialised."""

    def __init__(self, nrows, ncols, dtype=U8, channels=1, border=0, align=DEFAULT_ALIGN):
        self._setup(nrows, ncols, dtype, channels, border, align)
        # + 64: vpp's SIMD loops (fast.hpp:304-312) read whole 32-byte chunks past the last row's end
        store = np.zeros(self.alloc_bytes + align + 64, dtype=np.uint8)
        shift = (-store.ctypes.data) % align
        self.raw = store[shift:shift + self.alloc_bytes]
        self._store = store

    @property
    def desc(self):
        return self._desc(self.raw.ctypes.data)

    def view(self, with_border=False):
        return self._view(self.raw, with_border)

    def like(self, **kw):
        a = dict(nrows=self.nrows, ncols=self.ncols, dtype=self.dtype, channels=self.channels, border=self.border, align=self.align)
        a.update(kw)
        return HostImage(**a)


class DeviceImage(_ImageBase):
    """The same layout in HBM.  torch is only the allocator here."""

    def __init__(self, nrows, ncols, dtype=U8, channels=1, border=0, align=DEFAULT_ALIGN, device="cuda:0"):
        import torch
        self._setup(nrows, ncols, dtype, channels, border, align)
        # the image starts on a 256 B boundary at least 256 bytes into the tensor: the streaming kernels may begin a 16-byte chunk up to
        # 16 bytes before the first addressable byte (never used; see include/vpp_amd.h), which must not leave the allocation
        self.store = torch.zeros(self.alloc_bytes + 512, dtype=torch.uint8, device=device)
        self.shift = 256 + (-self.store.data_ptr()) % 256
        self.device = device

    @property
    def base_ptr(self):
        return self.store.data_ptr() + self.shift

    @property
    def desc(self):
        return self._desc(self.base_ptr)

    def upload(self, host):
        import torch
        assert host.alloc_bytes == self.alloc_bytes and host.pitch == self.pitch
        self.store[self.shift:self.shift + self.alloc_bytes].copy_(torch.from_numpy(host.raw))
        return self

    def download(self, host=None):
        host = host or HostImage(self.nrows, self.ncols, self.dtype, self.channels, self.border, self.align)
        host.raw[:] = self.store[self.shift:self.shift + self.alloc_bytes].cpu().numpy()
        return host

    @classmethod
    def from_host(cls, host, device="cuda:0"):
        return cls(host.nrows, host.ncols, host.dtype, host.channels, host.border, host.align, device).upload(host)


def desc_array(images):
    """C array of descriptors (pyramid levels)."""
    arr = (ImageDesc * len(images))()
    for i, im in enumerate(images):
        arr[i] = im.desc
    return arr
