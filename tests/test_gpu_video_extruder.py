"""video_extruder_update end to end against the REAL reference on BASELINE configs[4]: 2160x3840, 10 frames,
the defaults of video_extruder.hpp:35-41.  The checker is oracle/_ref/libvpp_ref_ve.so — matt-42/vpp's own video_extruder headers
compiled by oracle/ref/Makefile where /root/reference exists; the library travels to the GPU box with the repo.  There is no
silent skip: a checkout that has the reference but not the library fails, and so does a box that has neither."""
import os
import subprocess

import pytest

import cppbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    if not os.path.exists(cppbuild.REFLIB_VE) and os.path.isdir("/root/reference/vpp"):
        import __graft_entry__ as g
        g.build()   # builds oracle/_ref where /root/reference exists
    return cppbuild.video_extruder_parity()   # prebuilt by the CPU test / build(); recompiled only if stale and g++ exists


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
