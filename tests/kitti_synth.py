"""KITTI-format files for the evaluation-harness tests, written with an independent PNG encoder (zlib + numpy): every scanline
filter type of the PNG specification is exercised so that the C++ decoder's unfiltering is checked, not just its inflate."""
import os
import struct
import zlib

import numpy as np


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, arr, filters=(0,)):
    """arr: (h, w) or (h, w, ch) uint8 / uint16; filters: scanline filter types cycled over the rows."""
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, ch = a.shape
    depth = 16 if a.dtype == np.uint16 else 8
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = a.astype(">u2").view(np.uint8).reshape(h, -1) if depth == 16 else a.reshape(h, -1)
    bpp = ch * depth // 8
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for r in range(h):
        cur = rows[r].astype(np.int32)
        ft = filters[r % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0: out = cur
        elif ft == 1: out = cur - left
        elif ft == 2: out = cur - prev
        elif ft == 3: out = cur - ((left + prev) >> 1)
        else: out = cur - np.array([_paeth(int(x), int(y), int(z)) for x, y, z in zip(left, prev, upleft)], np.int32)
        raw.append(ft); raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    z = zlib.compress(bytes(raw), 6)
    half = len(z) // 2  # two IDAT chunks: the decoder must concatenate them
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))


def read_png(path):
    """Decoder for what evaluation/utils/png.hh writes (filter type 0 only)."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, z, hdr = 8, b"", None
    while p < len(d):
        n, t = struct.unpack(">I", d[p:p + 4])[0], d[p + 4:p + 8]
        body = d[p + 8:p + 8 + n]
        assert struct.unpack(">I", d[p + 8 + n:p + 12 + n])[0] == zlib.crc32(t + body)
        if t == b"IHDR": hdr = struct.unpack(">IIBBBBB", body)
        if t == b"IDAT": z += body
        p += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {0: 1, 2: 3}[ctype]
    raw = np.frombuffer(zlib.decompress(z), np.uint8).reshape(h, 1 + w * ch * depth // 8)
    assert (raw[:, 0] == 0).all()
    body = raw[:, 1:]
    return (body.reshape(h, w, ch, 2).astype(np.uint16) @ np.array([256, 1], np.uint16)).astype(np.uint16) if depth == 16 else body.reshape(h, w, ch)


def encode_flow(flow_rc, valid):
    """(h, w, 2) float (row, col) flow + (h, w) bool -> KITTI uint16 RGB (R = col flow, G = row flow, B = valid)."""
    q = np.clip(flow_rc * 64.0 + 32768.0, 0, 65535).astype(np.uint16)
    out = np.zeros(flow_rc.shape[:2] + (3,), np.uint16)
    out[..., 0] = np.where(valid, q[..., 1], 0); out[..., 1] = np.where(valid, q[..., 0], 0); out[..., 2] = valid
    return out


def make_tree(root, frames, flows, valids):
    """training/image_0/NNNNNN_10.png, _11.png (8-bit gray, as in KITTI) and training/flow_noc/NNNNNN_10.png."""
    os.makedirs(os.path.join(root, "training", "image_0"), exist_ok=True)
    os.makedirs(os.path.join(root, "training", "flow_noc"), exist_ok=True)
    for i, ((f1, f2), fl, va) in enumerate(zip(frames, flows, valids)):
        write_png(os.path.join(root, "training", "image_0", "%06d_10.png" % i), f1, filters=(1, 2, 0))
        write_png(os.path.join(root, "training", "image_0", "%06d_11.png" % i), f2, filters=(2,))
        write_png(os.path.join(root, "training", "flow_noc", "%06d_10.png" % i), encode_flow(fl, va), filters=(0, 1))
