import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b):
    """max|a-b| / max(|b|, tiny): error relative to the scale of the reference tensor."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


# Collection order under `pytest -x`: the quantizer's own parity suites come first, the optional
# extras (boundary convolutions, filtered search, matrix-core probe) after them, so that a failure
# in an extra cannot keep the forward / EMA / statistics rows from running.
_ORDER = ["test_oracle_golden", "test_cabi", "test_gpu_parity", "test_gpu_short_sweep", "test_gpu_module", "test_gpu_stats",
          "test_gpu_configs", "test_gpu_fullsize", "test_gpu_census", "test_gpu_boundary", "test_gpu_filtered",
          "test_gpu_filter_keys", "test_gpu_vmcnt_order"]
_LAST = ["test_zz_no_alloc"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _LAST:
            return len(_ORDER) + 1 + _LAST.index(name)
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=rank)          # stable: keeps the order inside a file
