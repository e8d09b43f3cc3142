import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests`` on a machine without a gfx950 skips the GPU tests instead of failing them one by one
    (the product has no CPU fallback: without the device every entry point raises BackendUnavailable)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        from pymbar_amd import _lib

        have = _lib.device_count() >= 1
        why = "no HIP device visible"
    except Exception as exc:  # library not built / not loadable
        have = False
        why = f"libmbar_hip.so unavailable ({exc})"
    if not have:
        skip = pytest.mark.skip(reason=f"needs an MI355X: {why}")
        for it in gpu_items:
            it.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _fresh_resident_cache():
    """The drop-in solver module keeps device copies of recently used host matrices (pymbar_amd.mbar_solvers._ResidentCache):
    no test inherits another one's."""
    yield
    from pymbar_amd import mbar_solvers

    mbar_solvers.drop_resident_cache()
