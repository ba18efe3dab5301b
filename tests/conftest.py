import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "testing_library: the test drives switches of include/trx_debug.h, which only "
                                       "libtrx_testing.so exports (set automatically: see pytest_collection_modifyitems)")


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
    # A test that names a switch or probe of include/trx_debug.h (trx_set_*, trx_debug_*) -- itself or through a helper of
    # its module -- runs on the TESTING library; every other test runs on the production library, libtrx.so, which
    # exports none of them (tests/test_abi.py).  The parity tests proper are of the second kind.
    import inspect
    import re
    pat = re.compile(r"trx_set_|trx_debug_")
    helpers = {}
    for item in items:
        fn = getattr(item, "function", None)
        if fn is None:
            continue
        mod = item.module
        if mod not in helpers:
            helpers[mod] = [n for n, f in vars(mod).items()
                            if inspect.isfunction(f) and f.__module__ == mod.__name__ and not n.startswith("test_")
                            and pat.search(inspect.getsource(f))]
        src = inspect.getsource(fn)
        if pat.search(src) or any(re.search(r"\b%s\(" % h, src) for h in helpers[mod]):
            item.add_marker(pytest.mark.testing_library)


@pytest.fixture(autouse=True)
def _library_of_the_test(request):
    if request.node.get_closest_marker("testing_library") is None:
        yield
        return
    from triceratops_amd import _lib
    _lib.use_testing_library(True)
    try:
        yield
    finally:
        _lib.use_testing_library(False)


@pytest.fixture(scope="session", autouse=True)
def _reference_sampling_mode():
    """The package default is set_sampling("device") (the fast path).  The suite's seeded comparisons with the
    reference need its host arithmetic on numpy's global stream, so the session runs in "numpy" mode and the
    device-mode tests switch explicitly."""
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")
    yield
