"""Host-side logic of pymbar_amd (protocol driver, gauge handling, unsampled states, scipy-backed
stages, covariance) exercised on the CPU stand-in (tests/cpu_standin.py) and checked against the golden
fixtures produced by the reference.  No GPU, no libmbar_hip compute calls."""
import logging

import numpy as np
import pytest

from oracle import mbar_oracle as oracle
from pymbar_amd import mbar_solvers as ms
from pymbar_amd import testsystems as ts
from pymbar_amd.utils import ParameterError, check_w_normalized, ensure_type, kln_to_kn
from tests.cpu_standin import OracleMatrix


def test_exports_match_reference_names():
    # names a replacement module must export
    for name in ["DEFAULT_SOLVER_PROTOCOL", "ROBUST_SOLVER_PROTOCOL", "JAX_SOLVER_PROTOCOL", "BOOTSTRAP_SOLVER_PROTOCOL",
                 "solve_mbar_for_all_states", "mbar_log_W_nk", "solve_mbar", "solve_mbar_once", "adaptive",
                 "mbar_gradient", "mbar_objective", "mbar_objective_and_gradient", "mbar_hessian", "mbar_W_nk",
                 "self_consistent_update", "precondition_u_kn", "validate_inputs", "use_jit",
                 "_setup_jax_acceleration", "scipy_minimize_options", "scipy_root_options"]:
        assert hasattr(ms, name), name
    assert ms.DEFAULT_SOLVER_PROTOCOL[0]["method"] == "hybr" and ms.DEFAULT_SOLVER_PROTOCOL[1]["method"] == "adaptive"
    assert ms.ROBUST_SOLVER_PROTOCOL[0] == dict(method="adaptive", options=dict(maxiter=1000))
    assert ms.BOOTSTRAP_SOLVER_PROTOCOL == (dict(method="adaptive", options=dict(min_sc_iter=0)),)


@pytest.mark.parametrize("value,expected", [("", False), ("true", True), ("YES", True), ("1", True), ("no", False)])
def test_setup_jax_env_parsing(monkeypatch, value, expected):
    # reference tests/test_mbar_solvers.py:94-125
    monkeypatch.setenv("PYMBAR_DISABLE_JAX", value)
    assert ms._setup_jax_acceleration() is expected


def test_l1_functions_on_standin_match_golden(golden):
    g = golden("exp_K20_N1000.npz")
    h = OracleMatrix(g["u_kn"])
    N_k, f = g["N_k"], g["f_eval"]
    np.testing.assert_allclose(ms.mbar_gradient(h, N_k, f), g["gradient"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(ms.self_consistent_update(h, N_k, f), g["sci"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ms.mbar_objective(h, N_k, f), g["objective"], rtol=1e-12)
    obj, grad = ms.mbar_objective_and_gradient(h, N_k, f)
    np.testing.assert_allclose(obj, g["objective"], rtol=1e-12)
    np.testing.assert_allclose(ms.mbar_hessian(h, N_k, f), g["hessian"], rtol=1e-9, atol=1e-9)
    logW = ms.mbar_log_W_nk(h, N_k, f)
    assert logW.shape == (1000, 20) and logW.flags.f_contiguous
    np.testing.assert_allclose(logW[:: int(g["logW_stride"])], g["logW_sample"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ms.mbar_W_nk(h, N_k, f).sum(0), g["logW_colsum"], rtol=1e-11)


def test_precondition_matches_reference(golden):
    g = golden("config1_ho_K5_N5000.npz")
    h = OracleMatrix(g["u_kn"])
    pre = ms.precondition_u_kn(h, g["N_k"], g["f_eval"])
    np.testing.assert_allclose(pre[:, ::97], g["precond_sample"], rtol=1e-12, atol=1e-11)


def test_adaptive_counts_and_solution(golden):
    g = golden("config1_ho_K5_N5000.npz")
    h = OracleMatrix(g["u_kn"])
    f, res = ms.solve_mbar_once(h, g["N_k"], np.zeros(5), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
    assert res["success"] and res["iterations"] == int(g["adaptive_iters"])
    np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-10, atol=1e-11)
    # per iteration: one Gram sweep + one two-candidate sweep (+ the initial gradient)
    assert h.calls["gram"] == res["iterations"]


@pytest.mark.parametrize("method", ["adaptive", "hybr", "lm", "L-BFGS-B", "BFGS", "Newton-CG", "trust-ncg", "dogleg",
                                    "CG", "trust-exact", "trust-krylov", "SLSQP", "TNC", "self-consistent-iteration"])
def test_every_method_reaches_reference_solution(golden, method):
    # reference tests/test_mbar_solvers.py:57-91 runs every method; here each must hit the reference's f_k
    g = golden("config1_ho_K5_N5000.npz")
    h = OracleMatrix(g["u_kn"])
    f, res = ms.solve_mbar_once(h, g["N_k"], np.zeros(5), method=method, tol=1e-12, options=dict())
    tol = 2e-5 if method in ("TNC", "CG", "SLSQP") else 1e-7
    np.testing.assert_allclose(f, g["f_k"], atol=tol)
    assert f[0] == 0.0


def test_unknown_method_raises():
    h = OracleMatrix(np.random.RandomState(0).rand(3, 30))
    with pytest.raises(ParameterError):
        ms.solve_mbar_once(h, np.array([10, 10, 10]), np.zeros(3), method="no-such-method")


def test_solve_mbar_protocol_and_fallback(golden, caplog):
    g = golden("config1_ho_K5_N5000.npz")
    h = OracleMatrix(g["u_kn"])
    f, all_results = ms.solve_mbar(h, g["N_k"], np.zeros(5))  # default: hybr succeeds, adaptive never runs
    assert len(all_results) == 1
    np.testing.assert_allclose(f, g["f_k"], atol=1e-9)
    # a stage that cannot converge (maxiter=1) falls through to the next one with continuation
    proto = (dict(method="adaptive", continuation=True, options=dict(maxiter=1)),
             dict(method="adaptive", options=dict(min_sc_iter=0)))
    with caplog.at_level(logging.WARNING):
        f2, res2 = ms.solve_mbar(h, g["N_k"], np.zeros(5), solver_protocol=proto)
    assert len(res2) == 2 and not res2[0]["success"] and res2[1]["success"]
    np.testing.assert_allclose(f2, g["f_k"], atol=1e-9)
    # total failure: the stage with the smallest gradient norm wins
    proto = (dict(method="adaptive", options=dict(maxiter=1)), dict(method="adaptive", options=dict(maxiter=2)))
    f3, res3 = ms.solve_mbar(h, g["N_k"], np.zeros(5), solver_protocol=proto)
    assert not res3[0]["success"] and not res3[1]["success"]
    g3 = np.linalg.norm(oracle.mbar_gradient(g["u_kn"], g["N_k"], f3))
    g1 = np.linalg.norm(oracle.mbar_gradient(g["u_kn"], g["N_k"], res3[0]["x"]))
    assert g3 <= g1


def test_all_states_with_unsampled_state(golden):
    g = golden("ho_unsampled_K4_N2300.npz")
    h = OracleMatrix(g["u_kn"])
    N_k = g["N_k"]
    sws = np.where(N_k != 0)[0]
    f = ms.solve_mbar_for_all_states(h, N_k, np.zeros(4), sws, ms.DEFAULT_SOLVER_PROTOCOL)
    np.testing.assert_allclose(f, g["f_k"], rtol=1e-9, atol=1e-10)
    f_r = ms.solve_mbar_for_all_states(h, N_k, np.zeros(4), sws, ms.ROBUST_SOLVER_PROTOCOL)
    np.testing.assert_allclose(f_r, g["f_k"], rtol=1e-9, atol=1e-10)
    # first state unsampled: it stays the zero of the gauge (mbar_solvers.py:1013-1015)
    u2, N2 = g["u_kn"][[2, 0, 1, 3]], N_k[[2, 0, 1, 3]]
    f2 = ms.solve_mbar_for_all_states(OracleMatrix(u2), N2, np.zeros(4), np.where(N2 != 0)[0], ms.BOOTSTRAP_SOLVER_PROTOCOL)
    assert f2[0] == 0.0
    np.testing.assert_allclose(f2 - f2[1], g["f_k"][[2, 0, 1, 3]] - g["f_k"][0], rtol=1e-9, atol=1e-9)


def test_solve_mbar_once_tolerates_unsampled_states(golden):
    """The reference's solve_mbar_once does not check N_k (mbar_solvers.py:738-883): a state with N_k = 0 carries no
    weight; here it is left out of the unknowns and keeps its f_k, the sampled states get the reference's solution."""
    g = golden("ho_unsampled_K4_N2300.npz")
    N_k = g["N_k"]
    assert np.any(N_k == 0)
    sws = np.where(N_k != 0)[0]
    for method in ("adaptive", "hybr", "L-BFGS-B"):
        f, res = ms.solve_mbar_once(OracleMatrix(g["u_kn"]), N_k, np.zeros(4), method=method, tol=1e-12)
        np.testing.assert_allclose(f[sws] - f[sws[0]], g["f_k"][sws] - g["f_k"][sws[0]], rtol=1e-8, atol=1e-9)
        assert np.all(f[N_k == 0] == 0.0)
    with pytest.raises(ParameterError):
        ms.solve_mbar_once(OracleMatrix(g["u_kn"]), np.zeros(4), np.zeros(4))


def test_single_sampled_state():
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0, 1], [1, 2], [50, 0], seed=1)
    f = ms.solve_mbar_for_all_states(OracleMatrix(u_kn), N_k, np.zeros(2), np.array([0]), ms.DEFAULT_SOLVER_PROTOCOL)
    assert f[0] == 0.0
    np.testing.assert_allclose(f, oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(2), np.array([0]))[0], atol=1e-12)


def test_config5_through_protocol(golden):
    g = golden("config5_alch_K40_N95000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    h = OracleMatrix(u_kn)
    f = ms.solve_mbar_for_all_states(h, N_k, np.zeros(40), np.where(N_k != 0)[0], ms.BOOTSTRAP_SOLVER_PROTOCOL)
    Delta_f = f - f[:, None]
    rel = np.abs(Delta_f - g["Delta_f"]) / np.maximum(np.abs(g["Delta_f"]), 1e-3)
    assert rel.max() < 1e-8  # BASELINE.json: Deltaf_ij within 1e-8 relative


def test_utils_behaviour():
    # reference tests/test_utils.py:50-61, 64-205, 208-241 (spot checks)
    with pytest.raises(TypeError):
        ensure_type([1.0, 2.0], np.float64, 1, "x")
    with pytest.raises(ValueError):
        ensure_type(np.zeros((2, 2)), np.float64, 1, "x")
    with pytest.raises(ValueError):
        ensure_type(np.zeros(3), np.float64, 1, "x", shape=(4,))
    with pytest.warns(RuntimeWarning):
        out = ensure_type(np.arange(3), np.float64, 1, "x")
    assert out.dtype == np.float64 and out.flags.c_contiguous
    assert ensure_type(None, np.float64, 1, "x", can_be_none=True) is None
    W = np.full((10, 2), 0.1)
    check_w_normalized(W, np.array([5, 5]))
    with pytest.raises(ParameterError):
        check_w_normalized(W * 1.01, np.array([5, 5]))
    with pytest.raises(ParameterError):
        check_w_normalized(W, np.array([5, 6]))
    kln = np.arange(2 * 3 * 4, dtype=float).reshape(2, 3, 4)
    kn = kln_to_kn(kln, N_k=np.array([2, 3]))
    assert kn.shape == (3, 5)
    np.testing.assert_array_equal(kn[:, :2], kln[0, :, :2])
    np.testing.assert_array_equal(kn[:, 2:], kln[1, :, :3])


def test_no_gpu_means_loud_failure():
    """The product path must not silently fall back to the CPU."""
    from pymbar_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.BackendUnavailable):
        ms.mbar_gradient(np.zeros((2, 8)), np.array([4, 4]), np.zeros(2))
    import pymbar_amd

    with pytest.raises(_lib.BackendUnavailable):
        pymbar_amd.MBAR(np.zeros((2, 8)), np.array([4, 4]))


def test_bench_without_a_launcher_refuses_to_measure_fewer_ranks_than_asked():
    """``python bench.py --gpus 2`` with no WORLD_SIZE in the environment spawns its own ranks; on a box with fewer GPUs (here:
    none) it must fail loudly with exit code 3 instead of printing a one-rank line."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-400:])
    assert "GPU(s) visible" in p.stderr and not p.stdout.strip()


def test_threaded_host_helpers_and_blas_limit():
    """utils.private_copy / utils.prefault: faithful below and above the 64 MB threshold (threads only above it); the covariance
    algebra never runs with the BLAS' default thread count from 48 states on."""
    from pymbar_amd import mbar as mbar_mod
    from pymbar_amd.utils import prefault, private_copy

    small = np.arange(12.0).reshape(3, 4)
    c = private_copy(small)
    assert np.array_equal(c, small) and c is not small and c.flags.c_contiguous
    big = np.random.default_rng(0).random((9, 1_000_000))  # 72 MB
    c = private_copy(big)
    assert np.array_equal(c, big) and c is not big
    out = prefault(np.empty((9, 1_000_000)))
    assert out.shape == (9, 1_000_000) and np.all(out == 0.0)
    tiny = np.empty((2, 3))
    assert prefault(tiny) is tiny

    limits = []

    class FakeController:
        def limit(self, limits=None):
            import contextlib

            @contextlib.contextmanager
            def cm():
                limits_seen.append(limits)
                yield

            return cm()

    limits_seen = limits
    old = mbar_mod._BLAS_CONTROLLER
    mbar_mod._BLAS_CONTROLLER = FakeController()
    try:
        for K in (10, 48, 128, 256, 257, 384, 600):
            with mbar_mod._small_blas(K):
                pass
    finally:
        mbar_mod._BLAS_CONTROLLER = old
    assert limits == [1, 1, 1, 4, 4, 4]
