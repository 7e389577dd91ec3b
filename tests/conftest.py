import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def avr_lib():
    """The product C-ABI library (built by hipcc; loadable without a GPU)."""
    from amrvolumerenderer_amd import _capi, build
    build.build()
    return _capi.lib()


@pytest.fixture(scope="session")
def ctx(avr_lib):
    """A rendering context on cuda:0 -- GPU tests only.  Fails loudly without a device."""
    from amrvolumerenderer_amd import runtime
    return runtime.Context(0)
