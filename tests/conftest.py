import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import mcpt_loader
    return mcpt_loader.load()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip(pkg):
    """The HIP backend.  Built in-tree with hipcc when missing or stale (hipcc cross-compiles gfx950 without a GPU);
    loading must not fall back to anything."""
    pkg.build.build()
    pkg.hip_backend.lib()
    return pkg.hip_backend
