import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Under `pytest -x` a failure hides everything collected after it, so the files run in the order of the evidence
# they carry: oracle / fixture parity first, the statistical comparisons next, the subprocess runs of bench.py
# (which measure the box as much as the code) last.  Files not named keep their alphabetical place in front.
_ORDER = [
    "test_gpu_kernels", "test_gpu_golden", "test_calc_probs_golden", "test_gpu_host_abi", "test_gpu_fused",
    "test_gpu_production_pin", "test_gpu_bounded", "test_gpu_batch", "test_gpu_star_chain", "test_toi465",
    "test_toi1228", "test_target_ops", "test_sharding", "test_gpu_rccl_world1",
    "test_gpu_equivalence", "test_gpu_notebook_anchors",
    "test_bench_contract",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i + 1 for i, name in enumerate(_ORDER)}

    def key(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(stem, 0)

    items.sort(key=key)          # stable: the order inside a file is untouched


@pytest.fixture(scope="session", autouse=True)
def _reference_sampling_mode():
    """The package default is set_sampling("device") (the fast path).  The suite's seeded comparisons with the
    reference need its host arithmetic on numpy's global stream, so the session runs in "numpy" mode and the
    device-mode tests switch explicitly."""
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")
    yield
