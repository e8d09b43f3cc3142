"""pymbar_amd -- the MBAR solver hot path of choderalab/pymbar on AMD MI355X (gfx950).

``pymbar_amd.mbar_solvers`` exports the names of ``pymbar.mbar_solvers``; ``pymbar_amd.MBAR`` mirrors
the solver-facing part of ``pymbar.MBAR``.  All K x N sweeps run in ``csrc/libmbar_hip.so``
(hand-written HIP, C ABI in ``include/mbar_hip.h``), reached through ctypes.  Importing the package
does not load the library or touch a GPU; the first computation does, and fails loudly if it cannot.
"""
from . import mbar_solvers, testsystems, utils
from ._lib import BackendUnavailable, MbarHipError, trim_device_cache
from .mbar import MBAR
from .utils import ParameterError

__all__ = ["MBAR", "mbar_solvers", "testsystems", "utils", "ParameterError", "BackendUnavailable", "MbarHipError",
           "trim_device_cache"]
__version__ = "0.1.0"
