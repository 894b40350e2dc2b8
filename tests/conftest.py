import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polar-codes_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    gdir = os.path.join(ROOT, "tests", "golden")

    class G:
        def __getattr__(self, name):
            z = np.load(os.path.join(gdir, name + ".npz"), allow_pickle=False)
            setattr(self, name, z)
            return z

    return G()


@pytest.fixture(autouse=True)
def _kernel_switches_enabled(monkeypatch):
    """Tests select kernels through the library's experiment switches (POLAR_SC_KERNEL=lds ...), which it reads from the
    environment only when POLAR_EXPERIMENT=1 is set."""
    monkeypatch.setenv("POLAR_EXPERIMENT", "1")
