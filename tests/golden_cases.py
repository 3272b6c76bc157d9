"""Seeded inputs of the golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference)."""
import zlib

import numpy as np

import pyr
from test_oracle_algos import pyrlk_cc_fixture, lk_scene
from util import rand_image, HostImage, rects_image, u8_image, texture, translate
from vpp_amd import image as vi


def crc(*arrs):
    c = 0
    for a in arrs:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


def box_case():
    src = rand_image(48, 80, vi.U8, 3, border=2, seed=3, fill_border=True)
    return src, src.like(border=0)


def add_case():
    b = rand_image(32, 48, vi.I32, seed=1, lo=0, hi=2**30 - 1)
    c = rand_image(32, 48, vi.I32, seed=2, lo=0, hi=2**30 - 1)
    return b, c, b.like()


def pyramid_case():
    img = rand_image(41, 56, vi.U8, 1, seed=7)
    return img, [HostImage(nr, nc, vi.U8, 1, 3) for nr, nc in pyr.level_dims(41, 56, 3)]


def fast_case():
    im = u8_image(rects_image(96, 128, seed=4), border=3)
    v = im.view(with_border=True)[..., 0]
    v[...] = np.pad(im.view()[..., 0], 3, mode="symmetric")
    return im


def pyrlk_case():
    f1, f2, kps = lk_scene(160, 200, 150)
    return u8_image(f1), u8_image(f2), kps


def lk_golden_case():
    f1, f2 = pyrlk_cc_fixture()
    return u8_image(f1), u8_image(f2), np.array([[50, 50], [48, 51], [52, 49], [50.5, 49.25]], np.float32)


def sdof_case():
    from test_gpu_sdof import flow_scene
    f1, f2, kps = flow_scene(100, 140)
    return u8_image(f1, border=3), u8_image(f2, border=3), kps, (9, 3, 0, 2, 5)


def ingest_case():
    """rgb frame (no border) -> gray with border 3, the examples/video_extruder.cc:46-48 chain; plus the plain 4-channel call."""
    rgb = rand_image(45, 70, vi.U8, 3, border=0, seed=21)
    rgba = rand_image(33, 52, vi.U8, 4, border=2, seed=22, fill_border=True)
    return rgb, HostImage(45, 70, vi.U8, 1, 3), rgba, HostImage(33, 52, vi.U8, 1, 2)


def fast_dense_case():
    """FAST_internals::fast_detector9(A, B, th) on the FAST fixture; thresholds incl. 0 and a negative one (plain int compares)."""
    return fast_case(), (20, 7, 0, -3)


def video_extruder_case():
    """A 7-frame sequence (texture + rectangles drifting by (0.9, -1.3) px per frame), gray frames with a mirror border of 3, and video_extruder_update's options
    (th, spacing, period, max trajectory length, scales, winsize, sweeps): re-detection every 3rd frame."""
    nr, nc, T = 96, 136, 7
    base = texture(nr + 40, nc + 40, seed=9, sigma=1.5)
    rect = rects_image(nr + 40, nc + 40, seed=4).astype(np.float64)
    frames = []
    for t in range(T):
        f = 0.6 * translate(base, 0.9 * t, -1.3 * t) + 0.4 * translate(rect, 0.9 * t, -1.3 * t)
        im = u8_image(np.clip(np.rint(f[20:20 + nr, 20:20 + nc]), 0, 255).astype(np.uint8), border=3)
        im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
        frames.append(im)
    return frames, (10, 10, 3, 15, 3, 9, 2)
