"""pytest plugin used by tests/test_reference_suite.py: runs the REFERENCE'S OWN test files (read in place under
/root/reference) with ``pymbar.MBAR`` and ``pymbar.mbar_solvers`` replaced by this repository's host-side mirror.
The device is replaced by the CPU stand-in (tests/cpu_standin.py, numpy oracle behind the DeviceMatrix interface), so
what is exercised is the drop-in boundary itself: names, argument meaning, return types, error behaviour and the
numerical results the reference's tests assert."""
import sys


def pytest_configure(config):
    import pymbar  # the reference (PYTHONPATH=/root/reference)
    import pymbar.mbar

    import pymbar_amd
    import pymbar_amd.device
    import pymbar_amd.mbar_solvers
    from tests.cpu_standin import OracleMatrix

    pymbar_amd.device.DeviceMatrix = OracleMatrix
    pymbar.MBAR = pymbar_amd.MBAR
    pymbar.mbar.MBAR = pymbar_amd.MBAR
    pymbar.mbar_solvers = pymbar_amd.mbar_solvers
    pymbar.mbar.mbar_solvers = pymbar_amd.mbar_solvers
    sys.modules["pymbar.mbar_solvers"] = pymbar_amd.mbar_solvers
    config.addinivalue_line("markers", "flaky: (reference marker)")


def pytest_sessionstart(session):
    import numpy as np

    np.random.seed(20240917)  # the reference's fixtures draw from numpy's global state: make the run reproducible


def pytest_runtest_setup(item):
    import zlib

    import numpy as np

    np.random.seed(zlib.crc32(item.nodeid.encode()) & 0x7FFFFFFF)
