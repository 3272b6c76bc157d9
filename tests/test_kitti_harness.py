"""the KITTI evaluation harness (evaluation/utils/kitti.hh, evaluation/semi_dense_optical_flow/KITTI.cc).
CPU: file formats and error statistics against an independent encoder / numpy.  GPU: the harness end to end on a synthetic
KITTI tree whose ground-truth flow is known."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import kitti_synth as ks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, "vpp_amd", "include"), "-I" + os.path.join(ROOT, "include")]


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a.astype("<u2")).tobytes())


def test_png_codec_flow_format_and_error_statistics(tmp_path):
    rng = np.random.default_rng(5)
    d = str(tmp_path)
    imgs = {"gray8": rng.integers(0, 256, (19, 23), dtype=np.uint8), "rgb8": rng.integers(0, 256, (17, 31, 3), dtype=np.uint8),
            "rgb16": rng.integers(0, 65536, (13, 21, 3), dtype=np.uint16), "gray16": rng.integers(0, 65536, (9, 11), dtype=np.uint16),
            "rgba8": rng.integers(0, 256, (8, 9, 4), dtype=np.uint8)}
    for name, a in imgs.items():
        ks.write_png(os.path.join(d, name + ".png"), a, filters=(0, 1, 2, 3, 4))
    h, w = 12, 16
    flow = rng.uniform(-40, 40, (h, w, 2)).astype(np.float32); flow = np.round(flow * 64) / 64  # representable in the format
    flow[0, 0] = (-600, 600)                                                                    # clamps at the 16-bit range
    valid = rng.random((h, w)) < 0.7; valid[3, 5] = True
    ks.write_png(os.path.join(d, "flow_in.png"), ks.encode_flow(flow, valid), filters=(4, 0))
    est = flow + rng.choice([0, 0.5, 2, 4, 7, 20], size=(h, w, 1)).astype(np.float32) * np.array([0.6, 0.8], np.float32)
    est_valid = rng.random((h, w)) < 0.8; est_valid[2, 2] = valid[2, 2] = True
    ks.write_png(os.path.join(d, "est.png"), ks.encode_flow(est, est_valid)); ks.write_png(os.path.join(d, "ref.png"), ks.encode_flow(flow, valid))
    exe = os.path.join(d, "kitti_io_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w"] + INC + [os.path.join(ROOT, "tests", "cpp", "kitti_io_test.cc"), "-o", exe, "-lz"])
    lines = dict(l.split(" ", 1) for l in subprocess.check_output([exe, d], text=True).strip().splitlines())
    for name, a in imgs.items():
        hh, ww = a.shape[:2]; ch = 1 if a.ndim == 2 else a.shape[2]
        assert lines[name] == "%d %d %d %d %d" % (ww, hh, ch, 16 if a.dtype == np.uint16 else 8, _crc(a)), name
    assert lines["missing"] == "0" and lines["gray_as_rgb"] == "1"
    fs = [float(x) for x in lines["flow_sample"].split()]
    assert fs == [pytest.approx(float(flow[3, 5, 0])), pytest.approx(float(flow[3, 5, 1])), 1.0]
    back = ks.read_png(os.path.join(d, "flow_out.png"))                     # what the C++ writer produced, read by the independent decoder
    np.testing.assert_array_equal(back, ks.encode_flow(np.clip(flow, -512, 65535 / 64 - 512), valid))
    # statistics (kitti.hh:75-135): vectors present in both, end-point error, shares above 1 / 3 / 5 / 10 px
    dec = lambda q: (q.astype(np.float32) - 32768) / 64
    e, r = ks.encode_flow(est, est_valid), ks.encode_flow(flow, valid)
    both = (e[..., 2] > 0) & (r[..., 2] > 0)
    err = np.sqrt((dec(e[..., 1]) - dec(r[..., 1])) ** 2 + (dec(e[..., 0]) - dec(r[..., 0])) ** 2)[both]
    st = lines["stats"].split()
    want = [100 * (err > t).mean() for t in (1, 3, 5, 10)] + [err.mean(), 100 * est_valid.mean()]
    assert [float(x) for x in st[:6]] == pytest.approx(want, rel=1e-4)
    assert int(st[6]) == both.sum() and int(st[7]) == min(int(err.reshape(-1)[np.flatnonzero(both.reshape(-1)).tolist().index(2 * w + 2)] * 20), 255)


@pytest.mark.gpu
def test_kitti_harness_end_to_end_on_a_synthetic_tree(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import texture
    exe = os.path.join(ROOT, "evaluation", "semi_dense_optical_flow", "KITTI")
    if not os.path.exists(exe):  # normally built by __graft_entry__.build(); build it here rather than fail on a fresh checkout
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-DVPP_AMD_DEVICE"] + INC + [os.path.join(ROOT, "evaluation", "semi_dense_optical_flow", "KITTI.cc"), "-o", exe,
                               "-L" + os.path.join(ROOT, "vpp_amd", "csrc"), "-lvpp_amd", "-lz", "-Wl,-rpath,$ORIGIN/../../vpp_amd/csrc", "-Wl,--allow-shlib-undefined"])
    h, w, moves = 188, 620, [(2, -3), (0, 5), (-4, 1)]   # KITTI frames are 375 x 1242; half of that keeps the CPU side quick
    frames, flows, valids = [], [], []
    for k, (dr, dc) in enumerate(moves):
        big = texture(h + 32, w + 32, seed=40 + k)
        f1 = np.ascontiguousarray(big[16:16 + h, 16:16 + w]); f2 = np.ascontiguousarray(big[16 - dr:16 - dr + h, 16 - dc:16 - dc + w])  # content moves by (dr, dc)
        frames.append((f1, f2)); flows.append(np.broadcast_to(np.array([dr, dc], np.float32), (h, w, 2)).copy()); valids.append(np.ones((h, w), bool))
    root = str(tmp_path / "kitti"); ks.make_tree(root, frames, flows, valids)
    cfg, res = str(tmp_path / "cfg.txt"), str(tmp_path / "res.txt")
    open(cfg, "w").write("# harness test\nnscales: 3\nwinsize = 9\npropagation 2\ndetector_th 5\nblock_size: 8\n")
    subprocess.check_call([exe, root, str(len(moves)), cfg, res])
    out = dict((k.strip(), float(v)) for k, v in (l.split(":") for l in open(res) if ":" in l and not l.startswith("#")))
    assert out["nkeypoints"] > 200 and out["runtime"] > 0
    assert out["errors"] < 5.0 and out["mean_endpoint_error"] < 1.0, out     # pure translations: nearly every vector within 3 px
    assert out["nscales"] == 3 and out["detector_th"] == 5 and out["block_size"] == 8 and out["patchsize"] == 5
    assert subprocess.call([exe, root, "99", cfg, res], stderr=subprocess.DEVNULL) == 3    # a missing pair is reported, not ignored
