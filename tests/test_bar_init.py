"""``initialize="BAR"`` (pymbar/mbar.py:1936-1988 -> other_estimators.bar, bisection) against the fixture generated
from the reference (tests/golden/make_golden_bar.py).  The chained guess is host-side code, so this runs without a GPU;
the constructor path is exercised on the GPU below."""
import os

import numpy as np
import pytest

from pymbar_amd import bar_init, testsystems as ts

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "bar_init.npz"))


def _systems():
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config1(seed=0)
    yield "config1", u_kn, np.asarray(N_k)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([1, 2, 3, 4], [0.5, 1.0, 1.5, 2.0], [1000, 500, 0, 800], seed=3)
    yield "unsampled", u_kn, np.asarray(N_k)
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    yield "config5", u_kn, np.asarray(N_k)


@pytest.mark.parametrize("tag,u_kn,N_k", list(_systems()), ids=["config1", "unsampled", "config5"])
def test_bar_zero_bisection_and_chain_match_reference(tag, u_kn, N_k):
    x = np.repeat(np.arange(len(N_k)), N_k)
    for (k, l), zeros, df in zip(GOLD[tag + "_pairs"], GOLD[tag + "_bar_zero"], GOLD[tag + "_bar_delta_f"]):
        w_F = u_kn[l, x == k] - u_kn[k, x == k]
        w_R = u_kn[k, x == l] - u_kn[l, x == l]
        got = [bar_init.bar_zero(w_F, w_R, d) for d in (-1.0, 0.0, 0.7)]
        np.testing.assert_allclose(got, zeros, rtol=1e-12, atol=1e-13)
        # same bracket, same midpoints, same stopping rule: the bisection lands on the same iterate
        np.testing.assert_allclose(bar_init.bar_bisection(w_F, w_R), df, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(bar_init.initialize_with_bar(u_kn, N_k, x), GOLD[tag + "_f_init"], rtol=1e-12, atol=1e-13)


def test_bar_edge_cases():
    rng = np.random.default_rng(0)
    w_F = rng.normal(1.0, 1.0, 400)
    w_R = rng.normal(-0.2, 1.0, 300)
    d = bar_init.bar_bisection(w_F, w_R)
    assert abs(bar_init.bar_zero(w_F, w_R, d)) < 1e-3
    # identical states: the estimate is exactly zero and the loop exits through the DeltaF == 0 branch
    assert bar_init.bar_bisection(np.zeros(10), np.zeros(7)) == 0.0
    # an iteration limit that cannot be met raises, and the chain then falls back to "no change" for that pair
    with pytest.raises(bar_init.BarConvergenceError):
        bar_init.bar_bisection(w_F, w_R, relative_tolerance=1e-15, maximum_iterations=3)
    # states without samples are skipped by the chain, unsampled neighbours are bridged
    u = rng.normal(size=(3, 30))
    N_k = np.array([10, 0, 20])
    f = bar_init.initialize_with_bar(u, N_k, np.repeat(np.arange(3), N_k))
    assert f[0] == 0.0 and f[1] == 0.0 and np.isfinite(f[2])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,u_kn,N_k", list(_systems()), ids=["config1", "unsampled", "config5"])
def test_mbar_initialize_bar_on_gpu(tag, u_kn, N_k):
    from pymbar_amd import MBAR

    m = MBAR(u_kn, N_k, initialize="BAR")
    np.testing.assert_allclose(m.f_k, GOLD[tag + "_f_k"], rtol=1e-8, atol=1e-9)


def test_bar_zero_keeps_its_digits_when_the_overlap_is_almost_nil():
    """Fermi sums in the denormal range (overlap ~ e^-720) must take the log-space form: ``log`` of a denormal sum has lost most
    of its digits (the reference works in log space throughout, other_estimators.py:129-151)."""
    from scipy.special import logsumexp

    from pymbar_amd.bar_init import bar_zero

    rng = np.random.RandomState(1)
    w_F = 725.0 + rng.standard_normal(200)
    w_R = -3.0 + rng.standard_normal(300)
    M = np.log(len(w_F) / len(w_R))
    want = logsumexp(-np.logaddexp(0.0, M + w_F - 0.0)) - logsumexp(-np.logaddexp(0.0, -(M - w_R - 0.0)))
    assert abs(bar_zero(w_F, w_R, 0.0) - want) < 1e-12 * abs(want)
    # and the fast path still agrees with it where both are fine
    w_F2 = 2.0 + rng.standard_normal(200)
    want2 = logsumexp(-np.logaddexp(0.0, M + w_F2 - 0.5)) - logsumexp(-np.logaddexp(0.0, -(M - w_R - 0.5)))
    assert abs(bar_zero(w_F2, w_R, 0.5) - want2) < 1e-12
