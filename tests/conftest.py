import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("NS3D_COOP_CHECK", "1")     # k_pt_persist: a bounded wait that expires is an error, not a silent pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as K
    K.lib()
    return K


@pytest.fixture(scope="session")
def hip():
    """navierstokes3d_amd.kernels on cuda:0 — fails loudly if the HIP extension or the GPU is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from navierstokes3d_amd import build
    build.build()
    from navierstokes3d_amd import kernels
    return kernels
