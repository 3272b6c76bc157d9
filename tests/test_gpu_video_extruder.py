"""video_extruder_update end to end against the REAL reference on BASELINE configs[4]: 2160x3840, 10 frames,
the defaults of video_extruder.hpp:35-41.  The checker is oracle/_ref/libvpp_ref_ve.so — matt-42/vpp's own video_extruder headers
compiled by oracle/ref/Makefile where /root/reference exists; the library travels to the GPU box with the repo.  There is no
silent skip: a checkout that has the reference but not the library fails, and so does a box that has neither."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libvpp_ref_ve.so")
OUT = os.path.join(ROOT, "tests", "cpp", "_build")


def _build():
    import __graft_entry__ as g
    g.build()   # also builds oracle/_ref where /root/reference exists
    assert os.path.exists(REFLIB), ("oracle/_ref/libvpp_ref_ve.so is missing: build it where /root/reference exists (make -C oracle ref); "
                                    "the video_extruder parity check has no other checker and does not skip")
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "video_extruder_parity")
    refdir = os.path.dirname(REFLIB)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-DVPP_AMD_DEVICE", "-I" + os.path.join(ROOT, "vpp_amd", "include"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "video_extruder_parity.cc"), "-o", exe, "-L" + os.path.join(ROOT, "vpp_amd", "csrc"), "-lvpp_amd",
                           "-L" + refdir, "-lvpp_ref_ve", "-Wl,-rpath," + os.path.join(ROOT, "vpp_amd", "csrc"), "-Wl,-rpath," + refdir, "-Wl,--allow-shlib-undefined"])
    return exe


def test_video_extruder_parity_program_builds():
    _build()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(240, 320, 9), (2160, 3840, 10)])
def test_video_extruder_matches_the_reference(shape):
    exe = _build()
    out = subprocess.run([exe] + [str(x) for x in shape], capture_output=True, text=True, timeout=1200)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "checker: the reference's own headers" in out.stdout and "video_extruder_parity ok" in out.stdout
