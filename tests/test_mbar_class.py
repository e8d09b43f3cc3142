"""MBAR class (pymbar_amd/mbar.py) on the CPU stand-in: API behaviour of pymbar/tests/test_mbar.py that
touches the solver path, checked against the reference's outputs in tests/golden/."""
import numpy as np
import pytest

import pymbar_amd
import pymbar_amd.device
from pymbar_amd import testsystems as ts
from pymbar_amd.utils import ParameterError
from tests.cpu_standin import OracleMatrix


@pytest.fixture(autouse=True)
def standin(monkeypatch):
    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", OracleMatrix)


def test_config1_free_energy_differences(golden):
    g = golden("config1_ho_K5_N5000.npz")
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"])
    np.testing.assert_allclose(mbar.f_k, g["f_k"], atol=1e-10)
    for method, tag in ((None, "svd_ew"), ("svd-ew", "svd_ew"), ("svd", "svd"), ("approximate", "approximate")):
        r = mbar.compute_free_energy_differences(uncertainty_method=method, return_theta=True)
        np.testing.assert_allclose(r["Delta_f"], g["Delta_f"], atol=1e-10)
        np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_" + tag], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(r["Theta"], g["Theta_" + tag], rtol=1e-6, atol=1e-9)
    ov = mbar.compute_overlap()
    np.testing.assert_allclose(ov["matrix"], g["overlap_matrix"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(np.real(ov["scalar"]), g["overlap_scalar"], rtol=1e-8)
    np.testing.assert_allclose(ov["matrix"].sum(1), 1.0, atol=1e-8)  # test_mbar.py:451-463
    np.testing.assert_allclose(mbar.compute_effective_sample_number(), g["N_eff"], rtol=1e-8)
    # weights: columns sum to one (test_mbar.py:466-472), layout like the reference
    W = mbar.weights()
    assert W.shape == (5000, 5)
    np.testing.assert_allclose(W.sum(0), 1.0, atol=1e-8)
    np.testing.assert_allclose(W @ g["N_k"], 1.0, atol=1e-10)


def test_unsampled_state_and_protocols(golden):
    g = golden("ho_unsampled_K4_N2300.npz")
    for proto in (None, "default", "robust", (dict(method="adaptive"),), (dict(method="L-BFGS-B", tol=1e-13),)):
        mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"], solver_protocol=proto)
        np.testing.assert_allclose(mbar.f_k, g["f_k"], rtol=1e-6, atol=1e-7)
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"])
    r = mbar.compute_free_energy_differences()
    np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_svd_ew"], rtol=1e-7, atol=1e-9)


def test_input_validation_and_options(golden):
    g = golden("config1_ho_K5_N5000.npz")
    u, N_k = g["u_kn"], g["N_k"]
    with pytest.raises(ParameterError):
        pymbar_amd.MBAR(u, N_k + 1)
    with pytest.raises(ParameterError):  # test_mbar.py:127-130
        pymbar_amd.MBAR(u, N_k, initial_f_k=np.zeros(4))
    with pytest.raises(ParameterError):
        pymbar_amd.MBAR(u, N_k, initialize="nonsense")
    m0 = pymbar_amd.MBAR(u, N_k)
    for init in ("zeros", "mean-reduced-potential"):  # test_mbar.py:203-214
        m = pymbar_amd.MBAR(u, N_k, initialize=init)
        np.testing.assert_allclose(m.f_k, m0.f_k, atol=1e-9)
    m = pymbar_amd.MBAR(u, N_k, initial_f_k=m0.f_k + 3.0)  # gauge shift is removed
    np.testing.assert_allclose(m.f_k, m0.f_k, atol=1e-9)
    with pytest.raises(ParameterError):
        m0.compute_free_energy_differences(uncertainty_method="bootstrap")
    with pytest.raises(ParameterError):
        m0.compute_free_energy_differences(uncertainty_method="bad")
    # module-level protocols are not mutated (the reference leaks options: SURVEY appendix A.3)
    assert "options" not in pymbar_amd.mbar_solvers.DEFAULT_SOLVER_PROTOCOL[0]


def test_u_kln_input_and_x_kindices():
    O_k, K_k = [0.0, 1.0, 2.0], [1.0, 2.0, 4.0]
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, [40, 50, 60], seed=2)
    u_kln = np.zeros((3, 3, 60))
    start = 0
    for k, n in enumerate(N_k):
        u_kln[k, :, :n] = u_kn[:, start : start + n]
        start += n
    a = pymbar_amd.MBAR(u_kn, N_k)
    b = pymbar_amd.MBAR(u_kln, N_k)
    np.testing.assert_allclose(a.f_k, b.f_k, atol=1e-12)


def test_bootstrap_is_deterministic_under_rseed():
    # reference tests/test_mbar.py:533-545: same seed, bit-identical bootstrap free energies
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0.0, 1.0, 2.0], [1.0, 2.0, 4.0], [60, 50, 40], seed=4)
    a = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=5, rseed=123)
    b = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=5, rseed=123)
    assert np.array_equal(a.f_k_boots, b.f_k_boots)
    r = a.compute_free_energy_differences(uncertainty_method="bootstrap")
    assert r["dDelta_f"].shape == (3, 3) and np.all(np.diag(r["dDelta_f"]) == 0)
    ra = a.compute_free_energy_differences()
    assert np.all(np.abs(r["dDelta_f"] - ra["dDelta_f"]) < 0.2)


def test_bootstrap_draws_are_the_references(golden):
    """The replicate indices come out of the reference's loop (mbar.py:425-431: per state, np.where over all samples, one
    rng.integers(N_k, size=N_k)) -- here with the samples of every state grouped ONCE; same random stream, same indices, for the
    default layout (contiguous runs) and for samples interleaved through x_kindices; and the threaded private copy of a large
    matrix is a faithful copy."""
    from pymbar_amd.mbar import _private_copy, _sample_groups

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0.0, 1.0, 2.0, 3.0], [1.0, 2.0, 4.0, 3.0], [60, 0, 50, 40], seed=4)
    N = int(np.sum(N_k))

    def reference_draws(x_kindices, rseed, n_bootstraps):
        rng = np.random.default_rng(rseed)
        rng.choice(np.arange(N), min(50, N))  # (the same-state probe of the constructor draws first: mbar.py:273-279)
        out = np.zeros([n_bootstraps, N], int)
        for b in range(n_bootstraps):
            for k in range(len(N_k)):
                k_indices = np.where(x_kindices == k)[0]
                out[b, k_indices] = k_indices[rng.integers(int(N_k[k]), size=int(N_k[k]))]
        return out

    default = np.repeat(np.arange(len(N_k)), N_k)
    m = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=3, rseed=7)
    assert np.array_equal(m.bootstrap_rints, reference_draws(default, 7, 3))
    perm = np.random.default_rng(0).permutation(N)  # the same samples in another order, labelled by x_kindices
    m2 = pymbar_amd.MBAR(u_kn[:, perm], N_k, n_bootstraps=3, rseed=7, x_kindices=default[perm])
    assert np.array_equal(m2.bootstrap_rints, reference_draws(default[perm], 7, 3))
    for x in (default, default[perm], np.zeros(0, dtype=int)):
        groups = _sample_groups(x, len(N_k))
        for k in range(len(N_k)):
            assert np.array_equal(np.asarray(groups[k], dtype=int), np.where(x == k)[0])
    big = np.random.default_rng(1).random((37, 250_000))  # 74 MB: the threaded path
    assert np.array_equal(_private_copy(big), big) and _private_copy(big) is not big


def test_bootstrap_replicates_from_the_counter_based_stream():
    """bootstrap_rng="device": the replicates come from the library's counter-based stream instead of numpy's generator --
    deterministic under rseed (reference tests/test_mbar.py:533-545), every state redraws ITS OWN samples (mbar.py:425-431), the
    lazily materialised bootstrap_rints are the draws the solves saw (oracle on the gathered matrix), for the default layout and
    for samples interleaved through x_kindices; other seeds give other draws; the uncertainties agree with the reference stream's
    statistically."""
    from oracle import mbar_oracle as oracle
    from pymbar_amd import _lib

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0.0, 1.0, 2.0, 3.0], [1.0, 2.0, 4.0, 3.0], [300, 0, 250, 200], seed=4)
    N, K = int(np.sum(N_k)), len(N_k)
    default = np.repeat(np.arange(K), N_k)
    a = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=6, rseed=123, bootstrap_rng="device")
    b = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=6, rseed=123, bootstrap_rng="device")
    c = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=6, rseed=124, bootstrap_rng="device")
    assert a._bootstrap_stream is not None and a._bootstrap_rints is None   # nothing of size N was kept
    assert np.array_equal(a.f_k_boots, b.f_k_boots) and not np.array_equal(a.f_k_boots, c.f_k_boots)
    rints = a.bootstrap_rints
    assert rints.shape == (6, N) and np.array_equal(rints, b.bootstrap_rints) and not np.array_equal(rints, c.bootstrap_rints)
    assert np.array_equal(default[rints], np.tile(default, (6, 1)))         # a state draws from its own samples
    assert len({tuple(r) for r in rints}) == 6                              # replicates differ
    sws = np.where(N_k > 0)[0]
    for i in range(6):
        fr, _ = oracle.solve_mbar_for_all_states(u_kn[:, rints[i]], N_k, a.f_k.copy(), sws, tol=1e-12, min_sc_iter=0)
        np.testing.assert_allclose(a.f_k_boots[i], fr, rtol=1e-8, atol=1e-9)
    # interleaved samples: the draws follow x_kindices
    perm = np.random.default_rng(0).permutation(N)
    m2 = pymbar_amd.MBAR(u_kn[:, perm], N_k, n_bootstraps=3, rseed=123, x_kindices=default[perm], bootstrap_rng="device")
    r2 = m2.bootstrap_rints
    assert np.array_equal(default[perm][r2], np.tile(default[perm], (3, 1)))
    fr, _ = oracle.solve_mbar_for_all_states(u_kn[:, perm][:, r2[1]], N_k, m2.f_k.copy(), sws, tol=1e-12, min_sc_iter=0)
    np.testing.assert_allclose(m2.f_k_boots[1], fr, rtol=1e-8, atol=1e-9)
    # the host function: uniform within a state (chi-square of 200 000 draws over 50 positions), counts sum to N
    cum = np.array([0, 50], dtype=np.int64)
    draws = np.concatenate([_lib.bootstrap_draws(99, rep, cum) for rep in range(4000)])
    counts = np.bincount(draws, minlength=50)
    chi2 = float(np.sum((counts - counts.mean()) ** 2 / counts.mean()))
    assert counts.sum() == 200_000 and chi2 < 100.0, chi2                   # (49 degrees of freedom: 100 is p ~ 2e-5)
    # statistically the same uncertainties as with the reference's generator
    ref = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=200, rseed=5)
    dev = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=200, rseed=5, bootstrap_rng="device")
    s_ref = ref.compute_free_energy_differences(uncertainty_method="bootstrap")["dDelta_f"][0, 3]
    s_dev = dev.compute_free_energy_differences(uncertainty_method="bootstrap")["dDelta_f"][0, 3]
    assert abs(s_dev - s_ref) < 0.25 * s_ref, (s_dev, s_ref)
    with pytest.raises(ParameterError):
        pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=2, bootstrap_rng="numpy")


def test_initialize_bar_and_unnormalized_log_weights(golden):
    """initialize="BAR" feeds the solver the chained pairwise guess (tests/golden/bar_init.npz holds the reference's
    values); _computeUnnormalizedLogWeights is the reference's one-line logsumexp (mbar.py:1919-1934)."""
    from scipy.special import logsumexp

    gb = golden("bar_init.npz")
    g = golden("config1_ho_K5_N5000.npz")
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"], initialize="BAR")
    np.testing.assert_allclose(mbar.f_k, gb["config1_f_k"], atol=1e-10)
    np.testing.assert_allclose(mbar._initialize_with_bar(mbar.u_kn), gb["config1_f_init"], rtol=1e-12, atol=1e-13)
    u_n = 0.5 * (g["u_kn"][1] + g["u_kn"][3])
    ref = -1.0 * logsumexp(mbar.f_k + u_n[:, np.newaxis] - mbar.u_kn.T, b=mbar.N_k, axis=1)
    np.testing.assert_allclose(mbar._computeUnnormalizedLogWeights(u_n), ref, rtol=1e-12, atol=1e-12)


def test_host_copy_of_u_kn_is_private_by_default(golden):
    """mbar.py:243: the object owns its copy of ``u_kn`` -- a caller who recycles the array afterwards must not change what the
    object reads (the device copy is frozen at construction).  ``copy=False`` (extension) keeps a READ-ONLY reference instead."""
    g = golden("config1_ho_K5_N5000.npz")
    u = np.array(g["u_kn"])
    m = pymbar_amd.MBAR(u, g["N_k"])
    ref = m.compute_perturbed_free_energies(m.u_kn)["Delta_f"].copy()
    u += 1.0e3 * np.arange(u.shape[0])[:, None]  # the caller re-uses its buffer
    np.testing.assert_array_equal(m.u_kn, g["u_kn"])
    np.testing.assert_allclose(m.compute_perturbed_free_energies(m.u_kn)["Delta_f"], ref, atol=1e-12)
    u2 = np.array(g["u_kn"])
    m2 = pymbar_amd.MBAR(u2, g["N_k"], copy=False)
    assert np.shares_memory(m2.u_kn, u2) and not m2.u_kn.flags.writeable
    with pytest.raises(ValueError):
        m2.u_kn[0, 0] = 1.0
    np.testing.assert_allclose(m2.f_k, m.f_k, atol=1e-12)


def test_private_copy_and_upload_side_by_side(monkeypatch, golden):
    """Large matrices: the constructor's private host copy and the upload both read the caller's array and run concurrently.
    Same object as the sequential path; an argument error raised in between does not leave the started device copy behind."""
    import pymbar_amd.mbar as mbar_mod

    closed = []

    class Tracking(OracleMatrix):
        def close(self):
            closed.append(1)

    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", Tracking)
    g = golden("config1_ho_K5_N5000.npz")
    u_kn, N_k = g["u_kn"], g["N_k"]
    m_seq = pymbar_amd.MBAR(u_kn, N_k)
    monkeypatch.setattr(mbar_mod, "_EARLY_UPLOAD_BYTES", 0)
    m_par = pymbar_amd.MBAR(u_kn, N_k)
    assert m_par.u_kn is not u_kn and np.array_equal(m_par.u_kn, u_kn) and m_par.u_kn.flags.writeable
    np.testing.assert_array_equal(m_par.f_k, m_seq.f_k)
    np.testing.assert_allclose(m_par.f_k, g["f_k"], atol=1e-9)
    assert "upload_s" in m_par.upload_stats
    n_closed = len(closed)
    with pytest.raises(Exception):
        pymbar_amd.MBAR(u_kn, N_k, initial_f_k=np.zeros(3))   # wrong length: raised after the upload thread was started
    assert len(closed) == n_closed + 1
