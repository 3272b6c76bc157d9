"""The reference's OWN unit tests on the MI355X: tests/*.cc of matt-42/vpp, unmodified, compiled by hipcc as single-source programs over
the drop-in headers (their pixel_wise / block_wise lambdas are gfx950 kernels, the algorithm headers dispatch into the C ABI) — built by
__graft_entry__.build() where the reference tree exists (oracle/ref/Makefile: unit_tests_gpu, outputs under oracle/_ref/), run here.
Asserts are on; exit code 0 means the reference's own expectations hold on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "oracle", "_ref", "unit_tests_gpu")
NAMES = ["imageNd", "image2d", "image3d", "imageNd_iterator", "boxNd_iterator", "box_nbh2d", "pixel_wise", "block_wise", "border", "fill", "sum",
         "colorspace_conversions", "pyramid", "tuple_utils", "window", "sandbox", "lbp", "cast"]


@pytest.mark.parametrize("name", NAMES)
def test_reference_unit_test_passes_on_the_gpu(name):
    exe = os.path.join(DIR, name)
    if not os.path.exists(exe):
        pytest.skip("not built: the reference tree was absent where build() ran")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
