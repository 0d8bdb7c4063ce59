import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library on a GPU box; GPU tests never fall back to anything else."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from dist_amd import lib
    return lib.load()
