import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _reference_sampling_mode():
    """The package default is set_sampling("device") (the fast path).  The suite's seeded comparisons with the
    reference need its host arithmetic on numpy's global stream, so the session runs in "numpy" mode and the
    device-mode tests switch explicitly."""
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")
    yield
