"""Seeded synthetic inputs (the BASELINE.md / SURVEY 8d configurations) shared by bench.py, the tools and the tests."""
import ctypes

import numpy as np

from vpp_amd import image as vi
from vpp_amd.image import HostImage, DeviceImage, U8, I32, F32  # noqa: F401

P = ctypes.byref


def rand_image(nrows, ncols, dtype=vi.U8, channels=1, border=0, seed=0, lo=None, hi=None, align=vi.DEFAULT_ALIGN, fill_border=False):
    rng = np.random.default_rng(seed)
    im = HostImage(nrows, ncols, dtype, channels, border, align)
    v = im.view(with_border=fill_border)
    if dtype == vi.F32:
        v[...] = rng.uniform(-100 if lo is None else lo, 100 if hi is None else hi, size=v.shape).astype(np.float32)
    else:
        info = np.iinfo(v.dtype)
        a = info.min if lo is None else lo
        b = info.max if hi is None else hi
        v[...] = rng.integers(a, b, size=v.shape, endpoint=True).astype(v.dtype)
    return im


def texture(nrows, ncols, seed=5, sigma=2.0):
    """Blurred-noise texture in 0..255 (BASELINE config 4)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, size=(nrows + 16, ncols + 16))
    k = np.exp(-0.5 * (np.arange(-6, 7) / sigma) ** 2)
    k /= k.sum()
    x = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, x)
    x = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, x)
    x = x[8:-8, 8:-8]
    x = (x - x.min()) / (x.max() - x.min())
    return x * 255.0


def translate(img, dr, dc):
    """Bilinear resample: out(r,c) = img(r - dr, c - dc) (content moves by (+dr,+dc)), replicate edges."""
    nr, nc = img.shape
    rr = np.clip(np.arange(nr)[:, None] - dr, 0, nr - 1.001)
    cc = np.clip(np.arange(nc)[None, :] - dc, 0, nc - 1.001)
    r0 = np.floor(rr).astype(int)
    c0 = np.floor(cc).astype(int)
    a = rr - r0
    b = cc - c0
    return (1 - a) * (1 - b) * img[r0, c0] + a * (1 - b) * img[r0 + 1, c0] + (1 - a) * b * img[r0, c0 + 1] + a * b * img[r0 + 1, c0 + 1]


def rects_image(nrows, ncols, seed=4, n=None):
    """Piecewise-constant random rectangles + +-4 noise (BASELINE config 3, FAST9)."""
    rng = np.random.default_rng(seed)
    img = np.full((nrows, ncols), 128, dtype=np.int32)
    n = n or (nrows * ncols) // 600
    for _ in range(n):
        h, w = rng.integers(4, 65, size=2)
        r, c = rng.integers(0, nrows), rng.integers(0, ncols)
        img[r:r + h, c:c + w] = rng.integers(0, 256)
    img += rng.integers(-4, 5, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def fast9_bench_frame(nrows=2160, ncols=3840, seed=4):
    """The FAST-9 bench / profile frame: rectangles + noise at a corner density of ~2.2 % at th = 20 (C3 asks for 1-3 %;
    rects_image's default rectangle count gives 5.5 %, which rounds 1-4 benched)."""
    return rects_image(nrows, ncols, seed=seed, n=(nrows * ncols) // 2400)


def u8_image(arr, border=0, align=vi.DEFAULT_ALIGN):
    im = HostImage(arr.shape[0], arr.shape[1], vi.U8, 1, border, align)
    im.view()[..., 0] = arr
    return im


def flow_scene(nr, nc, seed=6, spacing=5):
    """C4 texture with 4 piecewise translations (quadrants), |flow| <= 6 px (BASELINE config 5, scaled)."""
    tex = texture(nr, nc, seed=seed, sigma=1.5)
    f1 = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    f2 = f1.copy().astype(np.float64)
    shifts = [(2.0, -3.0), (-4.0, 1.0), (5.0, 4.0), (0.0, -6.0)]
    h, w = nr // 2, nc // 2
    for (dr, dc), (r0, c0) in zip(shifts, [(0, 0), (0, w), (h, 0), (h, w)]):
        f2[r0:r0 + h, c0:c0 + w] = translate(tex, dr, dc)[r0:r0 + h, c0:c0 + w]
    f2 = np.clip(np.rint(f2), 0, 255).astype(np.uint8)
    rr, cc = np.meshgrid(np.arange(spacing, nr - spacing, spacing), np.arange(spacing, nc - spacing, spacing), indexing="ij")
    kps = np.stack([rr.ravel(), cc.ravel()], 1).astype(np.int32)
    return f1, f2, kps
