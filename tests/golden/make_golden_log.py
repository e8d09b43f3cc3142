"""Log lines of the UNMODIFIED reference's ``adaptive(..., options={"verbose": True})`` as a fixture.

    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_log.py

Inputs: the matrices already stored in ``config1_ho_K5_N5000.npz`` / ``ho_unsampled_K4_N2300.npz``; outputs: every
``pymbar`` logger record (level, message) the reference emits for ``min_sc_iter`` 0, 2 and 5, for a run that stops at
``maxiter`` without converging, and the ``f_k`` it returns.  pymbar/mbar_solvers.py:598-660 is what writes them.
"""
import json
import logging
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

import pymbar  # noqa: E402  (the reference, from PYTHONPATH=/root/reference)
from pymbar import mbar_solvers as ref  # noqa: E402

assert os.path.realpath(pymbar.__file__).startswith("/root/reference"), pymbar.__file__


class Grab(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.DEBUG)
        self.records = []

    def emit(self, record):
        self.records.append([record.levelname, record.getMessage()])


def run(u_kn, N_k, **options):
    grab = Grab()
    lg = logging.getLogger("pymbar.mbar_solvers")
    old = lg.level
    lg.addHandler(grab)
    lg.setLevel(logging.DEBUG)
    try:
        tol = options.pop("tol", 1e-12)
        res = ref.adaptive(u_kn, 1.0 * N_k, np.zeros(len(N_k)), tol=tol, options=dict(verbose=True, **options))
    finally:
        lg.removeHandler(grab)
        lg.setLevel(old)
    return dict(records=grab.records, x=[float(v) for v in res["x"]], success=bool(res["success"]), options=options, tol=tol)


out = {}
with np.load(os.path.join(HERE, "config1_ho_K5_N5000.npz")) as g:
    u_kn, N_k = g["u_kn"], g["N_k"]
for name, opts in (("min_sc_iter_0", dict(min_sc_iter=0)), ("min_sc_iter_2", dict(min_sc_iter=2)), ("min_sc_iter_5", dict(min_sc_iter=5)),
                   ("maxiter_3_not_converged", dict(min_sc_iter=2, maxiter=3))):
    out["config1/" + name] = run(u_kn, N_k, **opts)
with open(os.path.join(HERE, "adaptive_verbose_log.json"), "w") as fh:
    json.dump(out, fh, indent=1)
for k, v in out.items():
    print(k, len(v["records"]), "records;", v["records"][-1])
