"""The CPU oracle (oracle/mbar_oracle.py) against fixtures produced by the unmodified reference
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare
the HIP path with the oracle."""
import numpy as np
import pytest

from oracle import mbar_oracle as oracle
from pymbar_amd import testsystems as ts

RTOL = 1e-11  # scipy 1.7.1 (fixtures) vs scipy 1.15 (here) logsumexp agree to ~1e-15


def _l1_check(u_kn, N_k, g):
    f = g["f_eval"]
    np.testing.assert_allclose(oracle.mbar_gradient(u_kn, N_k, f), g["gradient"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(oracle.self_consistent_update(u_kn, N_k, f), g["sci"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(oracle.mbar_objective(u_kn, N_k, f), g["objective"], rtol=1e-12)
    obj, grad = oracle.mbar_objective_and_gradient(u_kn, N_k, f)
    np.testing.assert_allclose(obj, g["objective2"], rtol=1e-12)
    np.testing.assert_allclose(grad, g["gradient2"], rtol=RTOL, atol=1e-9)
    H = oracle.mbar_hessian(u_kn, N_k, f)
    np.testing.assert_allclose(H, g["hessian"], rtol=1e-10, atol=1e-9 * np.abs(g["hessian"]).max())
    logW = oracle.mbar_log_W_nk(u_kn, N_k, f)
    stride = int(g["logW_stride"])
    np.testing.assert_allclose(logW[::stride], g["logW_sample"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.exp(logW).sum(0), g["logW_colsum"], rtol=1e-11)


def test_config1_l1_and_solution(golden):
    g = golden("config1_ho_K5_N5000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config1(seed=0)
    assert np.array_equal(u_kn, g["u_kn"])  # generator == reference sampler, bit for bit
    _l1_check(u_kn, N_k, g)
    # BASELINE.md oracle sanity value
    np.testing.assert_allclose(
        g["f_k"], [0, 0.372105072072219, 0.75085733928552, 1.219887010934637, 1.537263729186455], atol=1e-13)
    pre = oracle.precondition_u_kn(u_kn, N_k, g["f_eval"])
    np.testing.assert_allclose(pre[:, ::97], g["precond_sample"], rtol=1e-12, atol=1e-12)


def test_config1_adaptive_matches_reference(golden):
    g = golden("config1_ho_K5_N5000.npz")
    u_kn, N_k = g["u_kn"], g["N_k"]
    hist = []
    f, res = oracle.solve_mbar_once_adaptive(u_kn, N_k, np.zeros(5), tol=1e-12, min_sc_iter=0, history=hist)
    assert res["success"] and bool(g["adaptive_success"])
    assert res["iterations"] == int(g["adaptive_iters"])
    assert res["nr_iter"] == int(g["adaptive_nr"]) and res["sci_iter"] == int(g["adaptive_sci"])
    assert [1 if h["choice"] == "nr" else 0 for h in hist] == list(g["adaptive_choices"])
    np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-12, atol=1e-13)
    # the default protocol (hybr) and adaptive agree on the answer
    np.testing.assert_allclose(f, g["f_k"], rtol=1e-10, atol=1e-11)


def test_config1_covariance(golden):
    g = golden("config1_ho_K5_N5000.npz")
    u_kn, N_k, f_k = g["u_kn"], g["N_k"], g["f_k"]
    for method, tag in (("svd-ew", "svd_ew"), ("svd", "svd"), ("approximate", "approximate")):
        Delta_f, dDelta_f, Theta = oracle.free_energy_differences(u_kn, N_k, f_k, method=method)
        np.testing.assert_allclose(Delta_f, g["Delta_f"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(Theta, g["Theta_" + tag], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(dDelta_f, g["dDelta_f_" + tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(g["dDelta_f_svd_ew"][0, 1:4], [0.027949397229, 0.052489132279, 0.085439757729], atol=1e-11)


def test_unsampled_state_fixture(golden):
    g = golden("ho_unsampled_K4_N2300.npz")
    u_kn, N_k = g["u_kn"], g["N_k"]
    sws = np.where(N_k != 0)[0]
    f, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(4), sws, tol=1e-12, min_sc_iter=0)
    np.testing.assert_allclose(f, g["f_k"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(g["f_k_robust"], g["f_k"], rtol=1e-9, atol=1e-10)
    Delta_f, dDelta_f, Theta = oracle.free_energy_differences(u_kn, N_k, g["f_k"])
    np.testing.assert_allclose(dDelta_f, g["dDelta_f_svd_ew"], rtol=1e-8, atol=1e-10)


def test_exponentials_l1(golden):
    g = golden("exp_K20_N1000.npz")
    _l1_check(g["u_kn"], g["N_k"], g)


def test_oscillators_regenerated_inputs(golden):
    g = golden("osc_K50_N5000.npz")
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(1, 5, 50), np.linspace(1, 3, 50), [100] * 50, seed=7)
    _l1_check(u_kn, N_k, g)
    hist = []
    f, res = oracle.solve_mbar_once_adaptive(u_kn, N_k, np.zeros(50), tol=1e-12, min_sc_iter=0, history=hist)
    assert res["iterations"] == int(g["adaptive_iters"])
    assert res["nr_iter"] == int(g["adaptive_nr"])
    np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-10, atol=1e-11)
    # reference test_mbar_solvers.py:34-41 properties hold for the reference's own solution
    np.testing.assert_allclose(oracle.mbar_gradient(u_kn, N_k, g["f_k"]), 0, atol=1e-8)
    np.testing.assert_allclose(oracle.self_consistent_update(u_kn, N_k, g["f_k"]), g["f_k"], atol=1e-10)


def test_ladder_adaptive_and_sci_counts(golden):
    g = golden("ladder_K32_N32000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config2(seed=0, K=32, N=32000)
    f, res = oracle.solve_mbar_once_adaptive(u_kn, N_k, np.zeros(32), tol=1e-12, min_sc_iter=0)
    # the NR-vs-SCI choice on the LAST iteration compares two gradient norms that are both at
    # round-off level, so only the iteration count and the earlier choices are pinned
    assert res["iterations"] == int(g["adaptive_iters"]) and abs(res["nr_iter"] - int(g["adaptive_nr"])) <= 1
    np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-10, atol=1e-11)
    f2, res2 = oracle.solve_mbar_once_adaptive(u_kn, N_k, np.zeros(32), tol=1e-12)
    assert res2["iterations"] == int(g["adaptive_iters_msc2"])
    assert abs(res2["sci_iter"] - int(g["adaptive_sci_msc2"])) <= 1
    np.testing.assert_allclose(f2, g["f_adaptive_msc2"], rtol=1e-10, atol=1e-11)
    r = oracle.sci_solve(u_kn, N_k, np.zeros(32), tol=1e-12)
    assert r["success"] and abs(r["iterations"] - int(g["sci_iters"])) <= 1
    np.testing.assert_allclose(r["x"], g["f_sci"], rtol=1e-9, atol=1e-10)


def test_config5_alchemical_shape(golden):
    g = golden("config5_alch_K40_N95000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    assert np.array_equal(N_k, g["N_k"])
    sws = np.where(N_k != 0)[0]
    f, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(40), sws, tol=1e-12, min_sc_iter=0)
    np.testing.assert_allclose(f, g["f_k"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(g["f_k_adaptive_protocol"], g["f_k"], rtol=1e-8, atol=1e-9)
    Delta_f, dDelta_f, Theta = oracle.free_energy_differences(u_kn, N_k, g["f_k"])
    np.testing.assert_allclose(Delta_f, g["Delta_f"], atol=1e-13)
    np.testing.assert_allclose(dDelta_f, g["dDelta_f_svd_ew"], rtol=1e-8, atol=1e-10)


def test_sharded_partials_identities(golden):
    """s_k / Gram identities used by the device path reproduce gradient, SCI and Hessian."""
    g = golden("exp_K20_N1000.npz")
    u_kn, N_k, f = g["u_kn"], g["N_k"], g["f_eval"]
    parts = [oracle.shard_partials(u_kn[:, a:b], N_k, f, want_gram=True) for a, b in ((0, 300), (300, 1000))]
    psum = sum(p["psum"] for p in parts)
    gram = sum(p["gram"] for p in parts)
    grad, f_sci, H = oracle.partials_to_quantities(N_k, f, psum, gram)
    np.testing.assert_allclose(grad, g["gradient"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(f_sci, g["sci"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(H, g["hessian"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(sum(p["sumlogden"] for p in parts) - np.dot(N_k, f), g["objective"], rtol=1e-13)
