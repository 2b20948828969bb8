import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def built():
    """Make sure the CPU-side artefacts exist (oracle + synth generator + the HIP library file)."""
    import __graft_entry__ as g
    g.build_cpu_side()
    return True


@pytest.fixture(scope='session')
def gpu_ctx():
    """One device context for the whole GPU session.  Fails loudly when the library or the GPU is missing."""
    from pav_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
