import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from tissue_analysis_amd import _capi
        return _capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx():
    """One C-ABI context on cuda:0 for the whole GPU session (fails loudly if there is no GPU)."""
    from tissue_analysis_amd import _capi
    ctx = _capi.Context(0)
    yield ctx
    ctx.close()
