import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    """oracle/port.c — the C restatement (checker only)."""
    from oracle import binding
    return binding.port()


@pytest.fixture(scope="session")
def ref():
    """oracle/_ref — the reference kernel compiled for x86-64; absent where it was never built."""
    from oracle import binding
    r = binding.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return r


@pytest.fixture(scope="session")
def gpu_instance():
    from chunkyclplugin_amd.renderer import RendererInstance
    return RendererInstance.get(0)
