"""Seeded synthetic inputs for the MBAR solver path (numpy only; no GPU needed).

These are the input generators of SURVEY.md section 8(d).  The harmonic and exponential ladders
draw their samples exactly the way the reference test systems do -- legacy
``np.random.seed(seed)`` followed by one ``normal``/``exponential`` draw per state, in state
order (pymbar/testsystems/harmonic_oscillators.py:154-188,
pymbar/testsystems/exponential_distributions.py) -- so a given ``seed`` yields bit-identical
``u_kn`` here and in the reference (checked in tests/golden/make_golden.py).

Analytical answers (harmonic_oscillators.py:92-96): ``f_k = -1/2 ln(2 pi / (beta K_k))``.
"""
import numpy as np

__all__ = [
    "harmonic_u_kn",
    "harmonic_free_energies",
    "exponential_u_kn",
    "exponential_free_energies",
    "config1",
    "config2",
    "config3_params",
    "config5",
    "ladder_params",
]


def harmonic_u_kn(O_k, K_k, N_k, seed=None, beta=1.0):
    """Sample ``N_k[k]`` points from each 1-D harmonic state and evaluate every state on them.

    Returns ``(x_n, u_kn, N_k, s_n)`` with ``u_kn[l, n] = beta/2 * K_l * (x_n - O_l)**2``.
    """
    O_k = np.asarray(O_k, dtype=np.float64)
    K_k = np.asarray(K_k, dtype=np.float64)
    N_k = np.array(N_k, dtype=int)
    if not (len(O_k) == len(K_k) == len(N_k)):
        raise ValueError("O_k, K_k and N_k must have one entry per state")
    np.random.seed(seed)
    n_tot = int(N_k.sum())
    x_n = np.empty(n_tot, dtype=np.float64)
    s_n = np.empty(n_tot, dtype=int)
    start = 0
    for k, n in enumerate(N_k):
        sigma = (beta * K_k[k]) ** -0.5
        x_n[start : start + n] = np.random.normal(loc=O_k[k], scale=sigma, size=n)
        s_n[start : start + n] = k
        start += n
    u_kn = np.empty((len(O_k), n_tot), dtype=np.float64)
    for l in range(len(O_k)):
        u_kn[l] = beta * 0.5 * K_k[l] * (x_n - O_k[l]) ** 2.0
    return x_n, u_kn, N_k, s_n


def harmonic_free_energies(K_k, beta=1.0, subtract_component=0):
    fe = -0.5 * np.log(2 * np.pi / (beta * np.asarray(K_k, dtype=np.float64)))
    if subtract_component is not None:
        fe = fe - fe[subtract_component]
    return fe


def exponential_u_kn(rates, N_k, seed=None, beta=1.0):
    """Exponential-distribution ladder: ``u_kn[l, n] = beta * rate_l * x_n``."""
    rates = np.asarray(rates, dtype=np.float64)
    N_k = np.array(N_k, dtype=int)
    np.random.seed(seed)
    n_tot = int(N_k.sum())
    x_n = np.empty(n_tot, dtype=np.float64)
    s_n = np.empty(n_tot, dtype=int)
    start = 0
    for k, n in enumerate(N_k):
        x_n[start : start + n] = np.random.exponential(scale=rates[k] ** -1.0, size=n)
        s_n[start : start + n] = k
        start += n
    u_kn = np.empty((len(rates), n_tot), dtype=np.float64)
    for l in range(len(rates)):
        u_kn[l] = beta * rates[l] * x_n
    return x_n, u_kn, N_k, s_n


def exponential_free_energies(rates, beta=1.0, subtract_component=0):
    fe = np.log(beta * np.asarray(rates, dtype=np.float64))
    if subtract_component is not None:
        fe = fe - fe[subtract_component]
    return fe


def ladder_params(K):
    """Generator G of SURVEY.md 8(d): ``O_k = linspace(0, 4, K)``, ``K_k = linspace(1, 3, K)``."""
    return np.linspace(0.0, 4.0, K), np.linspace(1.0, 3.0, K)


def config1(seed=0):
    """BASELINE.json config 1: HarmonicOscillatorsTestCase defaults, K=5, N=5000."""
    O_k = (0, 1, 2, 3, 4)
    K_k = (1, 2, 4, 8, 16)
    return harmonic_u_kn(O_k, K_k, [1000] * 5, seed=seed) + (np.array(O_k, float), np.array(K_k, float))


def config2(seed=0, K=32, N=1_000_000):
    """BASELINE.json config 2: synthetic Gaussian ladder K=32, N=1e6 (equal N_k)."""
    O_k, K_k = ladder_params(K)
    return harmonic_u_kn(O_k, K_k, [N // K] * K, seed=seed) + (O_k, K_k)


def config3_params(K=128, N=10_000_000):
    """BASELINE.json config 3/4 parameters (the matrix itself is generated on the device)."""
    O_k, K_k = ladder_params(K)
    N_k = np.full(K, N // K, dtype=np.int64)
    return O_k, K_k, N_k


def config5(seed=0, K=40, n_per_state=2500, unsampled=(7, 23)):
    """BASELINE.json config 5: an alchemical-shaped ladder (K=40, N ~ 1e5) with force constants
    spaced geometrically 1 -> 16, small offsets, and two states carrying no samples."""
    K_k = np.geomspace(1.0, 16.0, K)
    O_k = np.linspace(0.0, 1.5, K)
    N_k = np.full(K, n_per_state, dtype=int)
    for k in unsampled:
        N_k[k] = 0
    return harmonic_u_kn(O_k, K_k, N_k, seed=seed) + (O_k, K_k)
