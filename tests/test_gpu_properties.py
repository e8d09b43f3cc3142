"""Analytical / structural properties of the MBAR estimate on the GPU path -- the checks the reference's tests/test_mbar.py
makes (overlap of identical states, :425-448; row sums and leading eigenvalue of the overlap matrix, :451-463; weight
normalisation, :466-472; z-scores against analytical free energies and expectations, :140-340; effective sample number)
restated on this repository's seeded test systems and run through the real device."""
import numpy as np
import pytest

import pymbar_amd
from pymbar_amd import testsystems as ts

pytestmark = pytest.mark.gpu
N_K = np.array([1000, 500, 0, 800])
O_K, K_K = np.array([1.0, 2.0, 3.0, 4.0]), np.array([0.5, 1.0, 1.5, 2.0])


@pytest.fixture(scope="module")
def ho():
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_K, K_K, N_K, seed=12)
    m = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=40, rseed=5)
    yield dict(mbar=m, x_n=x_n, u_kn=u_kn)
    m.close()


def test_overlap_of_identical_states_is_uniform():
    d = 4
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(2.0 * np.ones(d), 0.5 * np.ones(d), [100] * d, seed=1)
    m = pymbar_amd.MBAR(u_kn, N_k)
    r = m.compute_overlap()
    np.testing.assert_allclose(r["matrix"], np.full((d, d), 1.0 / d), atol=1e-8)
    ev = np.zeros(d)
    ev[0] = 1.0
    np.testing.assert_allclose(np.real(r["eigenvalues"]), ev, atol=1e-8)
    assert abs(np.real(r["scalar"]) - 1.0) < 1e-8
    np.testing.assert_allclose(m.f_k, 0.0, atol=1e-10)  # identical states: no free energy differences


def test_overlap_rows_and_weights_are_normalised(ho):
    m = ho["mbar"]
    r = m.compute_overlap()
    np.testing.assert_allclose(r["matrix"].sum(1), 1.0, atol=1e-8)
    assert abs(np.real(r["eigenvalues"][0]) - 1.0) < 1e-8
    W = m.weights()
    np.testing.assert_allclose(W.sum(0), 1.0, atol=1e-8)
    np.testing.assert_allclose(W @ N_K, 1.0, atol=1e-10)  # sum_k N_k W_nk = 1 for every sample
    n_eff = m.compute_effective_sample_number()
    assert np.all(n_eff > 1.0) and np.all(n_eff <= N_K.sum() + 1e-6)


@pytest.mark.parametrize("uncertainty_method", [None, "svd", "svd-ew", "bootstrap"])
def test_free_energies_match_analytical_within_error(ho, uncertainty_method):
    m = ho["mbar"]
    r = m.compute_free_energy_differences(uncertainty_method=uncertainty_method)
    fa = ts.harmonic_free_energies(K_K)
    ref = fa - fa[:, None]
    err = r["dDelta_f"] + np.eye(4)
    z = (r["Delta_f"] - ref) / err
    assert np.max(np.abs(z)) < 6.0
    np.testing.assert_allclose(np.diag(r["Delta_f"]), 0.0, atol=1e-12)


def test_position_expectations_match_analytical_within_error(ho):
    m, x_n = ho["mbar"], ho["x_n"]
    r = m.compute_expectations(x_n)
    z = (r["mu"] - O_K) / r["sigma"]          # <x>_k = O_k for a harmonic oscillator
    assert np.max(np.abs(z)) < 6.0
    r2 = m.compute_expectations(x_n ** 2)
    z2 = (r2["mu"] - (O_K ** 2 + 1.0 / K_K)) / r2["sigma"]   # <x^2>_k = O_k^2 + 1/K_k (beta = 1)
    assert np.max(np.abs(z2)) < 6.0
    rd = m.compute_expectations(x_n, output="differences")
    zd = (rd["mu"] - (O_K - O_K[:, None])) / (rd["sigma"] + np.eye(4))
    assert np.max(np.abs(zd)) < 6.0


def test_entropy_enthalpy_consistency(ho):
    m = ho["mbar"]
    r = m.compute_entropy_and_enthalpy()
    np.testing.assert_allclose(r["Delta_f"], r["Delta_u"] - r["Delta_s"], atol=1e-9)   # f = u - s (reduced units)
    fa = ts.harmonic_free_energies(K_K)
    z = (r["Delta_f"] - (fa - fa[:, None])) / (r["dDelta_f"] + np.eye(4))
    assert np.max(np.abs(z)) < 6.0


def test_exponential_system_matches_analytical_within_error():
    """Second model system of the reference's tests (exponential distributions with rates 1..4): f_k = ln(rate_k),
    <x>_k = 1 / rate_k, <x^2>_k = 2 / rate_k^2."""
    rates = np.array([1.0, 2.0, 3.0, 4.0])
    x_n, u_kn, N_k, s_n = ts.exponential_u_kn(rates, N_K, seed=21)
    m = pymbar_amd.MBAR(u_kn, N_k)
    r = m.compute_free_energy_differences()
    fa = ts.exponential_free_energies(rates)
    z = (r["Delta_f"] - (fa - fa[:, None])) / (r["dDelta_f"] + np.eye(4))
    assert np.max(np.abs(z)) < 6.0
    e = m.compute_expectations(x_n)
    assert np.max(np.abs((e["mu"] - 1.0 / rates) / e["sigma"])) < 6.0
    e2 = m.compute_expectations(x_n ** 2)
    assert np.max(np.abs((e2["mu"] - 2.0 / rates ** 2) / e2["sigma"])) < 6.0
    m.close()
