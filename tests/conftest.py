import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TOOLS = os.path.join(ROOT, "tools")        # tools/dropin_caller.py: the reference's restated loop body the dropin tests drive
if TOOLS not in sys.path:
    sys.path.insert(1, TOOLS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (hipcc cross-compiles gfx950 without a GPU)."""
    from naruto_amd import _lib
    _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
