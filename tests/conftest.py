import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (parity build of the restatement)."""
    from oracle import binding
    return binding.load(omp=False)


@pytest.fixture(scope="session")
def ref():
    """The reference's own headers compiled against shims (oracle/_ref); skip where it was not built."""
    from oracle import binding
    lib = binding.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libvpp_ref.so not built (no /root/reference on this box)")
    return lib


@pytest.fixture(scope="session")
def lib():
    """The product: C-ABI HIP library on cuda:0.  No fallback: a missing library is an error, not a skip."""
    import torch
    from vpp_amd import capi
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    L = capi.lib()
    capi.check(L.vpp_init(0))
    return L
