import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """CPU oracle (test infrastructure): builds liboracle.so on first use."""
    import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP product library; GPU tests fail loudly if it is missing."""
    from cuda_audio_amd import _lib

    return _lib.load()
