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


@pytest.fixture(scope="session")
def hip_check(pkg, hip):
    """Path of the checking build of the same ABI (libmcpt_hip_check.so: -DMCPT_CHECK_DIRECT_SKIP -DMCPT_TEST_HOOKS)."""
    path = pkg.build.build_check()
    hip.lib(path)
    return path


# MCPT_BVH x MCPT_QUANT_NODES (None = automatic); lbvh / ploc = the trees built on the GPU
TREES = [("sah", None), ("sah", "0"), ("reference", "0"), ("reference", "1"), ("lbvh", None), ("lbvh", "0"), ("ploc", None)]


@pytest.fixture
def tree_env(monkeypatch):
    """Selects the traversal tree of scenes created afterwards: (MCPT_BVH, MCPT_QUANT_NODES)."""
    def set_tree(bvh, quant):
        monkeypatch.setenv("MCPT_BVH", bvh)
        if quant is None:
            monkeypatch.delenv("MCPT_QUANT_NODES", raising=False)
        else:
            monkeypatch.setenv("MCPT_QUANT_NODES", quant)
    return set_tree
