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


def test_host_digest_sees_every_byte():
    """``mbar_host_digest`` (host-only entry point of the C library): thread-count independent, length-sensitive, and a change
    of ONE element anywhere -- first, last, the ragged tail of a chunk -- changes it."""
    from pymbar_amd import _lib

    rng = np.random.RandomState(0)
    a = rng.standard_normal((7, 300_001))  # (16.8 MB: several 1 MiB chunks and a ragged tail)
    d = _lib.host_digest(a)
    assert len(d) == 16 and _lib.host_digest(a, threads=1) == d and _lib.host_digest(a, threads=3) == d
    for idx in [(0, 0), (6, 300_000), (3, 12_345), (5, 131_072)]:
        old = a[idx]
        a[idx] = np.nextafter(old, np.inf)
        assert _lib.host_digest(a) != d, idx
        a[idx] = old
        assert _lib.host_digest(a) == d
    assert _lib.host_digest(np.zeros(3)) != _lib.host_digest(np.zeros(4))
    assert _lib.host_digest(np.zeros(0)) == _lib.host_digest(np.zeros(0))


def test_resident_cache_never_answers_for_an_edited_matrix(monkeypatch, golden):
    """The module-level functions are pure functions of the array they are handed (mbar_solvers.py:260-292): the device copy a
    host matrix got on its first call is re-used while the array is unchanged and dropped when ANY element changed -- including
    one that no sampled digest would see."""
    import pymbar_amd.device

    uploads = []

    class Counting(OracleMatrix):
        closed = 0

        @classmethod
        def from_host(cls, u_kn, device=None, columns=None):
            uploads.append(np.shape(u_kn))
            return super().from_host(u_kn, device=device, columns=columns)

        def close(self):
            Counting.closed += 1

    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", Counting)
    ms.drop_resident_cache()
    x_n, u_kn, N_k, s_n, _, _ = ts.config2(seed=3, K=6, N=30_000)
    f = np.linspace(0.0, 0.5, 6)
    g0 = ms.mbar_gradient(u_kn, N_k, f)
    g1 = ms.mbar_gradient(u_kn, N_k, f)
    assert len(uploads) == 1 and np.array_equal(g0, g1)
    np.testing.assert_allclose(g0, oracle.mbar_gradient(u_kn, N_k, f), rtol=1e-12, atol=1e-9)
    u_kn[3, 12_345] += 1.0  # (outside every sample a spot-check digest would take)
    g2 = ms.mbar_gradient(u_kn, N_k, f)
    assert len(uploads) == 2 and Counting.closed == 1  # re-uploaded, and the stale copy went away at once
    np.testing.assert_allclose(g2, oracle.mbar_gradient(u_kn, N_k, f), rtol=1e-12, atol=1e-9)
    assert not np.array_equal(g2, g0)
    # an entry in use is not closed by eviction: the last user closes it
    monkeypatch.setenv("PYMBAR_AMD_RESIDENT_CACHE", "1")
    other = u_kn[:, :1000].copy()
    with ms._Resident(u_kn) as h:
        closed_before = Counting.closed
        ms.mbar_gradient(other, N_k, f)  # evicts u_kn's entry (one entry allowed) while h is in use
        assert Counting.closed == closed_before
        np.testing.assert_allclose(ms.mbar_gradient(h, N_k, f), g2, rtol=0, atol=0)
    assert Counting.closed == closed_before + 1
    # switched off: every call uploads its own temporary
    monkeypatch.setenv("PYMBAR_AMD_RESIDENT_CACHE", "0")
    ms.drop_resident_cache()
    n = len(uploads)
    ms.mbar_gradient(u_kn, N_k, f)
    ms.mbar_gradient(u_kn, N_k, f)
    assert len(uploads) == n + 2


def test_resident_cache_is_thread_safe(monkeypatch):
    import threading

    import pymbar_amd.device

    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", OracleMatrix)
    ms.drop_resident_cache()
    x_n, u_kn, N_k, s_n, _, _ = ts.config2(seed=4, K=5, N=5000)
    mats = [u_kn, u_kn[:, :4000].copy(), u_kn[:, 1000:].copy()]
    f = np.linspace(0.0, 0.4, 5)
    want = [oracle.mbar_gradient(m, N_k, f) for m in mats]
    errors = []

    def worker(seed):
        rng = np.random.RandomState(seed)
        try:
            for _ in range(30):
                i = rng.randint(3)
                np.testing.assert_allclose(ms.mbar_gradient(mats[i], N_k, f), want[i], rtol=1e-12, atol=1e-9)
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_state_index_groups_accepts_what_the_reference_masks_accept():
    """``x_kindices == k`` (mbar.py:424, :1958-1967) works for float-typed indices and ignores values outside [0, K)."""
    from pymbar_amd.utils import state_index_groups

    K = 4
    rng = np.random.RandomState(2)
    for x in (np.repeat(np.arange(K), 5),                          # default layout: ranges
              rng.randint(0, K, size=50),                           # shuffled
              rng.randint(0, K, size=50).astype(np.float64),        # float-typed (np.bincount would raise TypeError)
              np.concatenate([rng.randint(-2, K + 2, size=60), [K, -1]]),  # out of range (np.bincount raises on negatives)
              np.array([0.0, 1.5, 2.0, np.nan, 3.0, 1.0]),          # non-integral / NaN match no state
              np.zeros(0, dtype=int)):
        groups = state_index_groups(x, K)
        assert len(groups) == K
        for k in range(K):
            np.testing.assert_array_equal(np.asarray(list(groups[k]), dtype=np.int64), np.where(np.asarray(x) == k)[0])


def test_label_samples_paths_agree():
    """Tabulated grid, sorted cells and (beyond int64 cell numbers) sorted index tuples give the labels of a plain loop."""
    from pymbar_amd import fes

    def naive(x_n, edges):
        x_n = np.atleast_2d(np.asarray(x_n, dtype=float).T).T if np.ndim(x_n) == 1 else np.asarray(x_n, dtype=float)
        if x_n.ndim == 1:
            x_n = x_n[:, None]
        seen, labels, grid = {}, [], []
        for row in x_n:
            idx = tuple(int(np.digitize(row[d], edges[d]) - 1) for d in range(len(edges)))
            key = None if min(idx) < 0 else idx
            if key not in seen:
                seen[key] = len(seen)
                grid.append(key)
            labels.append(seen[key])
        return np.array(labels), grid

    rng = np.random.RandomState(5)
    x1 = rng.uniform(-1.5, 1.5, size=3000)
    e1 = [np.linspace(-1, 1, 11)]
    lab, grid = fes.label_samples(x1, e1[0])                    # coarse grid, many samples: tabulated
    nl, ng = naive(x1[:, None], e1)
    assert np.array_equal(lab, nl) and grid == ng
    x3 = rng.uniform(-1.2, 1.2, size=(200, 3))
    e3 = [np.linspace(-1, 1, 41)] * 3                           # 74k cells for 200 samples: sorted cells
    lab, grid = fes.label_samples(x3, e3)
    nl, ng = naive(x3, e3)
    assert np.array_equal(lab, nl) and grid == ng
    x12 = rng.uniform(-1.2, 1.2, size=(150, 12))
    e12 = [np.linspace(-1, 1, 60)] * 12                         # 61^12 cells > 2^62: sorted tuples (ravel_multi_index would raise)
    lab, grid = fes.label_samples(x12, e12)
    nl, ng = naive(x12, e12)
    assert np.array_equal(lab, nl) and grid == ng


def _mbar_like_hessian(m, rng, disconnected=False):
    """H = diag(W^T 1) - W^T W for normalised weights: positive semi-definite, null vector 1 (mbar_solvers.py:395-411)."""
    W = rng.random((4 * m, m))
    if disconnected:  # two groups of states that share no sample: a second null vector
        W[: 2 * m, m // 2:] = 0.0
        W[2 * m:, : m // 2] = 0.0
    W /= W.sum(1, keepdims=True)
    return np.diag(W.sum(0)) - W.T @ W


# (321: the first blocked size; 353: 352 unknowns = eleven full blocks, only the right-hand side's row below the last; 330 / 401 / 520: a partial last block)
@pytest.mark.parametrize("m", [2, 5, 40, 130, 321, 330, 353, 401, 520])
def test_host_newton_direction_matches_lstsq(m):
    """The K x K step of the host-driven loop (``mbar_host_newton_direction``): lstsq(H, g) minus its first component
    (mbar_solvers.py:582-583) -- plain Cholesky below 320 unknowns, column blocks shared out over host threads above; the
    blocked form gives the SAME bits for every team size."""
    from pymbar_amd import _lib

    rng = np.random.default_rng(m)
    H = _mbar_like_hessian(m, rng)
    g = rng.normal(size=m) * 1e-2
    g -= g.mean()  # (the gradient of MBAR sums to zero: g is in the range of H)
    ref = np.linalg.lstsq(H, g, rcond=-1)[0]
    ref -= ref[0]
    scale = np.abs(ref).max()
    x = _lib.host_newton_direction(H, g)
    assert x[0] == 0.0 and np.abs(x - ref).max() <= 1e-10 * scale
    teams = [_lib.host_newton_direction(H, g, threads=t) for t in (1, 2, 5, 7)]
    for xt in teams:
        assert np.array_equal(xt, teams[0]) and np.abs(xt - ref).max() <= 1e-10 * scale
    x_old = _lib.host_newton_direction(H, g, threads=-1)  # the panels-of-4 factorisation
    assert np.abs(x_old - ref).max() <= 1e-10 * scale


def test_host_team_survives_a_fork():
    """The factorisation's worker threads are parked between calls; a forked child has none of them and must start its own team on
    first use instead of waiting for threads that do not exist in it."""
    import os

    from pymbar_amd import _lib

    rng = np.random.default_rng(3)
    H = _mbar_like_hessian(500, rng)
    g = rng.normal(size=500) * 1e-2
    g -= g.mean()
    x_parent = _lib.host_newton_direction(H, g, threads=4)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child
        code = 1
        try:
            x_child = _lib.host_newton_direction(H, g, threads=4)
            code = 0 if np.array_equal(x_child, x_parent) else 2
        finally:
            os.write(w, bytes([code]))
            os._exit(code)
    os.close(w)
    import select

    ready, _, _ = select.select([r], [], [], 60.0)
    if not ready:
        os.kill(pid, 9)
    os.waitpid(pid, 0)
    assert ready, "the forked child hung in the host factorisation"
    assert os.read(r, 1) == bytes([0])
    assert np.array_equal(_lib.host_newton_direction(H, g, threads=4), x_parent)  # (and the parent's team is still there)


def test_host_team_survives_a_fork_while_another_thread_uses_it():
    """fork() while another thread of the parent is INSIDE the team (its job mutex held, workers running): the fork handlers take
    the team's locks around the fork, so the child never inherits a mutex locked by a thread it does not have; the child factors
    several times (workers re-created and placed anew), the parent's thread keeps getting the same bits throughout."""
    import os
    import select
    import threading

    from pymbar_amd import _lib

    rng = np.random.default_rng(5)
    H = _mbar_like_hessian(700, rng)
    g = rng.normal(size=700) * 1e-2
    g -= g.mean()
    x_ref = _lib.host_newton_direction(H, g, threads=4)
    stop, bad = threading.Event(), []

    def busy():
        while not stop.is_set():
            if not np.array_equal(_lib.host_newton_direction(H, g, threads=4), x_ref):
                bad.append(1)

    t = threading.Thread(target=busy, daemon=True)
    t.start()
    try:
        for _ in range(6):
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:  # child
                code = 1
                try:
                    ok = all(np.array_equal(_lib.host_newton_direction(H, g, threads=4), x_ref) for _ in range(3))
                    code = 0 if ok else 2
                finally:
                    os.write(w, bytes([code]))
                    os._exit(code)
            os.close(w)
            ready, _, _ = select.select([r], [], [], 60.0)
            if not ready:
                os.kill(pid, 9)
            os.waitpid(pid, 0)
            assert ready, "the forked child hung in the host factorisation"
            assert os.read(r, 1) == bytes([0])
            os.close(r)
    finally:
        stop.set()
        t.join(60.0)
    assert not t.is_alive() and not bad


def test_host_newton_direction_falls_back_to_the_pseudo_inverse():
    """Disconnected groups of states: the gauge-fixed block is singular, the Cholesky factorisations (both forms) report the
    breakdown and the minimum-norm solution of lstsq is returned instead."""
    from pymbar_amd import _lib

    rng = np.random.default_rng(7)
    for m in (24, 340):
        H = _mbar_like_hessian(m, rng, disconnected=True)
        g = rng.normal(size=m) * 1e-2
        g[: m // 2] -= g[: m // 2].mean()
        g[m // 2:] -= g[m // 2:].mean()
        ref = np.linalg.lstsq(H, g, rcond=-1)[0]
        ref -= ref[0]
        for threads in (0, 3):
            x = _lib.host_newton_direction(H, g, threads=threads)
            assert x[0] == 0.0 and np.all(np.isfinite(x))
            assert np.abs(H @ x - g).max() <= 1e-12
            # (each group's constant is a null direction of H: whether LAPACK keeps or drops a singular value of 1e-15 decides it in
            # the reference itself -- compare what H determines)
            d = x - ref
            d[: m // 2] -= d[: m // 2].mean()
            d[m // 2:] -= d[m // 2:].mean()
            assert np.abs(d).max() <= 1e-10 * np.abs(ref).max()
