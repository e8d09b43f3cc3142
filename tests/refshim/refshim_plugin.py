"""pytest plugin used by tests/test_reference_suite.py and tools/reference_on_gpu_box.sh: runs the REFERENCE'S OWN test
files (read in place from the reference tree on PYTHONPATH) with ``pymbar.MBAR`` and ``pymbar.mbar_solvers`` replaced by this
repository's host-side mirror.

``MBAR_REFSHIM_DEVICE=standin`` (default; the build container has no GPU): the device is the CPU stand-in
(tests/cpu_standin.py, numpy oracle behind the DeviceMatrix interface) -- what is exercised is the drop-in boundary itself:
names, argument meaning, return types, error behaviour and the numerical results the reference's tests assert.
``MBAR_REFSHIM_DEVICE=hip`` (an MI355X box with a staged copy of the reference tree): the real ``DeviceMatrix`` /
``libmbar_hip.so``; the session ends by printing the process's mapping of the library (``/proc/self/maps``)."""
import os
import sys

ON_HIP = os.environ.get("MBAR_REFSHIM_DEVICE", "standin") == "hip"


def pytest_configure(config):
    import pymbar  # the reference (PYTHONPATH=/root/reference)
    import pymbar.mbar

    import pymbar_amd
    import pymbar_amd.device
    import pymbar_amd.mbar_solvers
    if ON_HIP:
        from pymbar_amd import _lib

        _lib.require_device()  # raises BackendUnavailable without libmbar_hip.so and a gfx950 device
    else:
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


def pytest_terminal_summary(terminalreporter):
    if not ON_HIP:
        return
    import pymbar
    import pymbar_amd.device

    with open("/proc/self/maps") as fh:
        mapped = sorted({line.split()[-1] for line in fh if "libmbar_hip" in line or "/oracle/" in line})
    terminalreporter.write_line("refshim: reference tree %s; device %s; mapped native code %s; oracle imported: %s" % (
        os.path.dirname(pymbar.__file__), pymbar_amd.device.device_info()["name"], mapped,
        any(m == "oracle" or m.startswith("oracle.") for m in sys.modules)))
