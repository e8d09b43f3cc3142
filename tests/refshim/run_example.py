"""Run one of the REFERENCE'S example scripts (in place under /root/reference/examples) with ``pymbar.MBAR`` and
``pymbar.mbar_solvers`` replaced by this repository's drop-in -- on the CPU stand-in device, or with
``MBAR_REFSHIM_DEVICE=hip`` on the real one (see refshim_plugin.py); ``MBAR_REFERENCE_TREE`` names the reference tree.
Usage: python run_example.py <reference tree>/examples/<dir>/<script>.py"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, os.environ.get("MBAR_REFERENCE_TREE", "/root/reference"), ROOT]

import numpy as np  # noqa: E402

np.random.seed(0)
import pymbar  # noqa: E402
import pymbar.mbar  # noqa: E402

import pymbar_amd  # noqa: E402
import pymbar_amd.device  # noqa: E402
import pymbar_amd.mbar_solvers  # noqa: E402
if os.environ.get("MBAR_REFSHIM_DEVICE", "standin") != "hip":
    from tests.cpu_standin import OracleMatrix  # noqa: E402

    pymbar_amd.device.DeviceMatrix = OracleMatrix
pymbar.MBAR = pymbar_amd.MBAR
pymbar.mbar.MBAR = pymbar_amd.MBAR
pymbar.mbar_solvers = pymbar_amd.mbar_solvers
sys.modules["pymbar.mbar_solvers"] = pymbar_amd.mbar_solvers
os.chdir(os.environ.get("TMPDIR", "/tmp"))  # the examples write plots / tables into the working directory
runpy.run_path(sys.argv[1], run_name="__main__")
if os.environ.get("MBAR_REFSHIM_DEVICE", "standin") == "hip":
    with open("/proc/self/maps") as fh:
        print("run_example: mapped native code", sorted({ln.split()[-1] for ln in fh if "libmbar_hip" in ln}), file=sys.stderr)
