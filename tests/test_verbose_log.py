"""``adaptive(..., options={"verbose": True})`` writes the reference's log lines (pymbar/mbar_solvers.py:598-660): the
per-iteration gradient norms, the two wordings of a self-consistent choice ("because min_sci_iter=..." / "for lower
gradient"), the Newton-Raphson line, the summary and the not-converged warnings.  Fixture: the records of the UNMODIFIED
reference on config 1 (tests/golden/make_golden_log.py -> adaptive_verbose_log.json).

Compared: level and wording of every record exactly (numbers masked), iteration indices and choices exactly -- except the
choice of the LAST iteration of a converged run, which compares two gradient norms at round-off level (both candidates are
the fixed point) -- and the printed numbers to their 5 printed digits where they are above round-off."""
import json
import logging
import os
import re

import numpy as np
import pytest

import pymbar_amd.device
from pymbar_amd import mbar_solvers as ms
from tests.conftest import GOLDEN, load_golden
from tests.cpu_standin import OracleMatrix

NUM = re.compile(r"(?<![\w.])[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?(?![\w.])")
CASES = ("min_sc_iter_0", "min_sc_iter_2", "min_sc_iter_5", "maxiter_3_not_converged")


def _fixture():
    with open(os.path.join(GOLDEN, "adaptive_verbose_log.json")) as fh:
        return json.load(fh)


def _records(u_kn, N_k, tol, options):
    got = []

    class Grab(logging.Handler):
        def emit(self, record):
            got.append([record.levelname, record.getMessage()])

    lg = logging.getLogger(ms.logger.name)
    h, old = Grab(level=logging.DEBUG), lg.level
    lg.addHandler(h)
    lg.setLevel(logging.DEBUG)
    try:
        res = ms.adaptive(u_kn, N_k, np.zeros(len(N_k)), tol=tol, options=dict(verbose=True, **options))
    finally:
        lg.removeHandler(h)
        lg.setLevel(old)
    return got, res


def _compare(got, want, converged):
    assert len(got) == len(want), (got, want)
    # index of the last iteration's choice line (noise in the reference itself when the run converged)
    choice_lines = [i for i, (_, m) in enumerate(want) if m.startswith(("Choosing self-consistent", "Newton-Raphson used"))]
    last_choice = choice_lines[-1] if (converged and choice_lines) else -1
    for i, ((lv_g, msg_g), (lv_w, msg_w)) in enumerate(zip(got, want)):
        assert lv_g == lv_w, (i, msg_g, msg_w)
        if i == last_choice:
            assert msg_g.startswith(("Choosing self-consistent", "Newton-Raphson used")) and msg_g.split()[-1] == msg_w.split()[-1]
            continue
        if msg_w.startswith("Of ") and converged:  # (the split of the counts follows the last choice: total and wording only)
            assert NUM.sub("#", msg_g) == NUM.sub("#", msg_w) and NUM.findall(msg_g)[0] == NUM.findall(msg_w)[0]
            continue
        assert NUM.sub("#", re.sub(r"\s+", " ", msg_g)) == NUM.sub("#", re.sub(r"\s+", " ", msg_w)), (i, msg_g, msg_w)
        for a, b in zip(NUM.findall(msg_g), NUM.findall(msg_w)):
            a, b = float(a), float(b)
            if "gradient norm" in msg_w or "max_delta" in msg_w or "Converged to tolerance" in msg_w:
                if abs(b) > 1e-9:  # (below: sums of round-off)
                    assert abs(a - b) <= 2e-4 * abs(b), (i, msg_g, msg_w)
            else:
                assert a == b, (i, msg_g, msg_w)


def _run_cases():
    fx = _fixture()
    g = load_golden("config1_ho_K5_N5000.npz")
    for name in CASES:
        want = fx["config1/" + name]
        opts = dict(want["options"])
        got, res = _records(g["u_kn"], g["N_k"], want["tol"], opts)
        assert res["success"] == want["success"]
        _compare(got, want["records"], want["success"])
        if want["success"]:
            np.testing.assert_allclose(res["x"], want["x"], atol=1e-10)


def test_verbose_lines_match_the_reference_on_the_stand_in(monkeypatch):
    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", OracleMatrix)
    _run_cases()


@pytest.mark.gpu
def test_verbose_lines_match_the_reference_on_the_device():
    _run_cases()
