import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def built():
    """Compile the native libraries once (hipcc cross-compiles gfx950 without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    return True


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
