"""SURVEY.md 8(b) / INTEGRATION.md section 2, tested literally: the reference's own, UNCHANGED ``pymbar.MBAR`` with only
``pymbar.mbar.mbar_solvers`` re-bound to ``pymbar_amd.mbar_solvers`` must reproduce the reference (default and "robust"
protocols, every uncertainty method, expectations, perturbed free energies, overlap, bootstraps) to 1e-12 (1e-10 for
covariance-derived quantities).  Build container only: the reference tree is not on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pymbar")), reason="reference tree not mounted")
def test_unchanged_reference_MBAR_on_the_drop_in_solver_module():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "refshim"), REF, ROOT])
    out = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "tests", "refshim", "boundary_check.py")],
                         env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    verdict = json.loads(out.stdout.strip().splitlines()[-1])
    assert verdict["ok"] and verdict["checks"] >= 60, verdict
    assert verdict["worst_deviation"] < 1e-9, verdict
    # the drop-in keeps the matrix resident between the unchanged MBAR's calls into the solver module: one upload per object
    # (+ one per bootstrap replicate, each a freshly gathered matrix)
    assert verdict["uploads_per_construction"] == 1 and verdict["uploads_per_construction_with_3_bootstraps"] == 4, verdict
