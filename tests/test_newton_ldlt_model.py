"""The lane-level numpy statement of the blocked LDL^T Newton solve (tools/newton_ldlt_model.py: registers as 64-vectors, the fp64
matrix instruction as its documented layout, LDS as dictionaries) against numpy.linalg.solve -- the layout algebra of
pymbar_amd/csrc/mbar_k_solver.hip::newton_body_ldlt (transposed accumulator storage, pivots from the last row upwards, the ride-along
right-hand side in row 0, the forward substitution with x_0 = -1), checked on the CPU: state counts on both sides of the 16-state block
boundaries, unsampled states, a gauge state that is not state 0."""
import importlib.util
import os

import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location(
    "newton_ldlt_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "newton_ldlt_model.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("K", [3, 16, 17, 33, 64, 65, 112, 128])
@pytest.mark.parametrize("kind", ["all sampled", "a fifth unsampled", "state 0 unsampled"])
def test_lane_level_model_solves_the_gauge_fixed_newton_system(K, kind):
    rng = np.random.default_rng(1000 * K + len(kind))
    sampled = np.ones(K, bool)
    if kind == "a fifth unsampled" and K > 4:
        sampled[rng.choice(K, K // 5, replace=False)] = False
    if kind == "state 0 unsampled":
        sampled[0] = False
    first = int(np.where(sampled)[0][0])
    A, nb, xref, live = model.build_problem(rng, K, sampled, first)
    x, piv = model.solve(A, nb)
    scale = max(1e-300, np.max(np.abs(xref)))
    assert np.max(np.abs(x[live] - xref[live])) / scale < 1e-11
    dead = ~live
    dead[0] = False
    assert np.all(x[dead] == 0.0) and x[0] == -1.0          # identity rows stay out of it; row 0 is the ride-along right-hand side
    assert np.all(piv[live] > 0.0)                           # the pivots of an SPD block, in the reversed elimination order
