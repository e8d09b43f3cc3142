"""CPU oracle for the MBAR solver hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module restates, in plain numpy/scipy on the CPU, the algorithm of the reference
``pymbar/mbar_solvers.py`` (+ the covariance contraction of ``pymbar/mbar.py``).  It exists only
to *check* the MI355X path: it may be imported by ``tests/``, by ``__graft_entry__.smoke()`` and by
the ``cpu_baseline`` leg of ``bench.py`` -- never by anything under ``pymbar_amd/``.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py`` against
fixtures under ``tests/golden/`` that were produced by running the unmodified reference
(``/root/reference``, numpy backend) on seeded inputs -- see ``tests/golden/make_golden.py``.

All citations are relative to ``/root/reference/``.  The reference's own arithmetic is
``scipy.special.logsumexp`` + ``numpy.dot`` + ``numpy.linalg.lstsq`` (unpinned third-party
versions, pyproject.toml:27-35); the same calls are used here so the oracle is the same
floating-point algorithm, not a lookalike.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import logsumexp

__all__ = [
    "log_denominator",
    "self_consistent_update",
    "mbar_gradient",
    "mbar_objective",
    "mbar_objective_and_gradient",
    "mbar_hessian",
    "mbar_log_W_nk",
    "mbar_W_nk",
    "precondition_u_kn",
    "adaptive",
    "sci_solve",
    "solve_mbar_once_adaptive",
    "solve_mbar_for_all_states",
    "covariance_theta",
    "error_of_differences",
    "free_energy_differences",
    "shard_partials",
    "partials_to_quantities",
]


# --------------------------------------------------------------------------------------------
# L1 array math (kernels k1..k10)
# --------------------------------------------------------------------------------------------
def log_denominator(u_kn, N_k, f_k):
    """``log sum_k N_k exp(f_k - u_kn)`` per sample (MBAR Eq. 9 denominator).

    Follows pymbar/mbar_solvers.py:238 (also :290, :335, :349, :403, :447, :706): a weighted
    log-sum-exp over the state axis with weights ``b=N_k``; zero-weight (unsampled) states
    contribute nothing.
    """
    N_k = np.asarray(N_k, dtype=np.float64)
    return logsumexp(f_k - u_kn.T, b=N_k, axis=1)


def self_consistent_update(u_kn, N_k, f_k):
    """One self-consistent iteration, Eq. C3: ``f_k' = -LSE_n(-logden_n - u_kn)``.

    Follows pymbar/mbar_solvers.py:231-242.
    """
    logden_n = log_denominator(u_kn, N_k, f_k)
    return -1.0 * logsumexp(-logden_n - u_kn, axis=1)


def mbar_gradient(u_kn, N_k, f_k):
    """Gradient of the MBAR objective, Eq. C6.  Follows pymbar/mbar_solvers.py:284-292."""
    N_k = np.asarray(N_k, dtype=np.float64)
    logden_n = log_denominator(u_kn, N_k, f_k)
    lognum_k = logsumexp(-logden_n - u_kn, axis=1)
    return -1.0 * N_k * (1.0 - np.exp(f_k + lognum_k))


def mbar_objective(u_kn, N_k, f_k):
    """``sum_n logden_n - N_k . f_k``.  Follows pymbar/mbar_solvers.py:327-338."""
    N_k = np.asarray(N_k, dtype=np.float64)
    return np.sum(log_denominator(u_kn, N_k, f_k)) - np.dot(N_k, f_k)


def mbar_objective_and_gradient(u_kn, N_k, f_k):
    """Objective and gradient sharing one denominator.  Follows pymbar/mbar_solvers.py:341-355."""
    N_k = np.asarray(N_k, dtype=np.float64)
    logden_n = log_denominator(u_kn, N_k, f_k)
    lognum_k = logsumexp(-logden_n - u_kn, axis=1)
    grad = -1.0 * N_k * (1.0 - np.exp(f_k + lognum_k))
    obj = np.sum(logden_n) - np.dot(N_k, f_k)
    return obj, grad


def mbar_log_W_nk(u_kn, N_k, f_k):
    """``log W_nk = f_k - u_kn - logden_n``, shape (N, K).  pymbar/mbar_solvers.py:439-449."""
    logden_n = log_denominator(u_kn, N_k, f_k)
    return f_k - u_kn.T - logden_n[:, np.newaxis]


def mbar_W_nk(u_kn, N_k, f_k):
    """``exp(log W_nk)``.  pymbar/mbar_solvers.py:476-483."""
    return np.exp(mbar_log_W_nk(u_kn, N_k, f_k))


def mbar_hessian(u_kn, N_k, f_k):
    """Eq. C9: ``H = diag(N_k sum_n W_nk) - (N_i N_j) (W^T W)_ij``.  pymbar/mbar_solvers.py:395-411."""
    N_k = np.asarray(N_k, dtype=np.float64)
    W = mbar_W_nk(u_kn, N_k, f_k)
    H = np.dot(W.T, W)
    H *= N_k
    H *= N_k[:, np.newaxis]
    H -= np.diag(W.sum(0) * N_k)
    return -1.0 * H


def precondition_u_kn(u_kn, N_k, f_k):
    """Per-sample shift of ``u_kn`` that zeroes the objective at ``f_k``.  pymbar/mbar_solvers.py:697-707."""
    N_k = np.asarray(N_k, dtype=np.float64)
    u = u_kn - u_kn.min(0)
    u = u + logsumexp(f_k - u.T, b=N_k, axis=1) - np.dot(N_k, f_k) / N_k.sum()
    return u


# --------------------------------------------------------------------------------------------
# The adaptive Newton-Raphson / self-consistent loop
# --------------------------------------------------------------------------------------------
def _relative_change(f_new, f_old, f_sci, f_nr, tol):
    """Convergence measures of pymbar/mbar_solvers.py:627-633 (first state excluded)."""
    div = np.abs(f_new[1:])
    zeroed = np.abs(f_new[1:]) < np.min([10**-8, tol])
    div = np.where(zeroed, 1.0, div)
    max_delta = np.max(np.abs(f_new[1:] - f_old[1:]) / div)
    max_diff = np.max(np.abs(f_sci[1:] - f_nr[1:]) / div)
    return max_delta, max_diff


def adaptive(u_kn, N_k, f_k, tol=1.0e-8, maxiter=10000, min_sc_iter=2, gamma=1.0, history=None):
    """Adaptive NR / SCI loop: each iteration forms both candidates and keeps the one whose
    gradient norm is smaller.  Follows the numpy branch of pymbar/mbar_solvers.py:570-667.

    Returns ``dict(success, message, x, nr_iter, sci_iter, iterations)``.
    """
    N_k = np.asarray(N_k, dtype=np.float64)
    f_k = np.array(f_k, dtype=np.float64)
    g = mbar_gradient(u_kn, N_k, f_k)  # :570
    nr_iter = sci_iter = 0
    success = False
    message = "Did not converge."
    iterations = 0
    for iteration in range(maxiter):
        H = mbar_hessian(u_kn, N_k, f_k)  # :581
        Hinvg = np.linalg.lstsq(H, g, rcond=-1)[0]  # :582
        Hinvg -= Hinvg[0]  # :583
        f_nr = f_k - gamma * Hinvg  # :584
        f_sci = self_consistent_update(u_kn, N_k, f_k)  # :587
        f_sci = f_sci - f_sci[0]  # :588
        g_sci = mbar_gradient(u_kn, N_k, f_sci)  # :589
        gnorm_sci = np.dot(g_sci, g_sci)
        g_nr = mbar_gradient(u_kn, N_k, f_nr)  # :593
        gnorm_nr = np.dot(g_nr, g_nr)
        f_old = f_k
        if gnorm_sci < gnorm_nr or sci_iter < min_sc_iter:  # :607
            f_k, g = f_sci, g_sci
            sci_iter += 1
            choice = "sci"
        else:
            f_k, g = f_nr, g_nr
            nr_iter += 1
            choice = "nr"
        max_delta, max_diff = _relative_change(f_k, f_old, f_sci, f_nr, tol)
        iterations = iteration + 1
        if history is not None:
            history.append(dict(choice=choice, gnorm_sci=math.sqrt(gnorm_sci), gnorm_nr=math.sqrt(gnorm_nr),
                                max_delta=max_delta, max_diff=max_diff))
        if np.isnan(max_delta) or ((max_delta < tol) and max_diff < np.sqrt(tol)):  # :636
            success = True
            message = "Convergence achieved by change in f with respect to previous guess."
            break
    return dict(success=success, message=message, x=f_k, nr_iter=nr_iter, sci_iter=sci_iter,
                iterations=iterations)


def sci_solve(u_kn, N_k, f_k, tol=1.0e-12, maxiter=10000):
    """Pure self-consistent iteration (BASELINE.json config 2).

    The reference has no such MBAR method; the reference-equivalent loop is
    ``f <- self_consistent_update(u, N_k, f); f -= f[0]`` (pymbar/mbar_solvers.py:587-588) until the
    relative change of pymbar/mbar_solvers.py:627-631 drops below ``tol``.
    """
    f_k = np.array(f_k, dtype=np.float64)
    iterations = 0
    success = False
    for iteration in range(maxiter):
        f_new = self_consistent_update(u_kn, N_k, f_k)
        f_new = f_new - f_new[0]
        max_delta, _ = _relative_change(f_new, f_k, f_new, f_new, tol)
        f_k = f_new
        iterations = iteration + 1
        if np.isnan(max_delta) or max_delta < tol:
            success = True
            break
    return dict(success=success, x=f_k, iterations=iterations)


def solve_mbar_once_adaptive(u_kn, N_k, f_k, tol=1.0e-12, **options):
    """``solve_mbar_once(method="adaptive")``: gauge f_0=0, float N_k, precondition, loop.

    Follows pymbar/mbar_solvers.py:790-793 and :848-850.
    """
    N_k = 1.0 * np.asarray(N_k)
    f_k = np.asarray(f_k, dtype=np.float64)
    f_k = f_k - f_k[0]
    u = precondition_u_kn(np.ascontiguousarray(u_kn, dtype=np.float64), N_k, f_k)
    results = adaptive(u, N_k, f_k, tol=tol, **options)
    return results["x"], results


def solve_mbar_for_all_states(u_kn, N_k, f_k, states_with_samples, tol=1.0e-12, **options):
    """Solve on the sampled states, then one all-state self-consistent update fills in the
    unsampled ones and f_0 is re-zeroed.  Follows pymbar/mbar_solvers.py:999-1015 with a
    single-stage adaptive protocol.
    """
    N_k = np.asarray(N_k)
    f_k = np.array(f_k, dtype=np.float64)
    sws = np.asarray(states_with_samples)
    if len(sws) == 1:
        f_nz = np.array([0.0])
        results = None
    else:
        f_nz, results = solve_mbar_once_adaptive(u_kn[sws], N_k[sws], f_k[sws], tol=tol, **options)
    f_k[sws] = f_nz
    f_k = self_consistent_update(u_kn, N_k, f_k)
    f_k -= f_k[0]
    return f_k, results


# --------------------------------------------------------------------------------------------
# Covariance / free-energy differences
# --------------------------------------------------------------------------------------------
def covariance_theta(W, N_k, method="svd-ew"):
    """Asymptotic covariance ``Theta``.  Follows pymbar/mbar.py:1796-1864.

    ``svd-ew``: eigendecomposition of ``W^T W`` (:1849), negative eigenvalues clamped (:1851),
    ``Theta = V S pinv(I - S V^T diag(N_k) V S, rcond=1e-10) S V^T`` (:1853-1858, :1735).
    """
    N_k = np.asarray(N_k)
    K = N_k.size
    if method in (None, "bootstrap"):
        method = "svd-ew"
    if method == "approximate":
        return W.T @ W
    Ndiag = np.diag(N_k)
    ident = np.identity(K, dtype=np.float64)
    if method == "svd":
        _, S, Vt = np.linalg.svd(W, full_matrices=False)
        Sigma = np.diag(S)
        V = Vt.T
    elif method == "svd-ew":
        S2, V = np.linalg.eigh(W.T @ W)
        S2[np.where(S2 < 0.0)] = 0.0
        Sigma = np.diag(np.sqrt(S2))
    else:
        raise ValueError(f"Method {method} unrecognized.")
    inner = np.linalg.pinv(ident - Sigma @ V.T @ Ndiag @ V @ Sigma, rcond=1.0e-10)
    return V @ Sigma @ inner @ Sigma @ V.T


def error_of_differences(cov, warning_cutoff=1.0e-10):
    """``sqrt(cov_ii + cov_jj - 2 cov_ij)`` with tiny negatives zeroed.  pymbar/mbar.py:1687-1715."""
    diag = cov.diagonal()
    d2 = diag + np.vstack(diag) - 2 * cov
    cutoff = -abs(warning_cutoff)
    if np.any(d2 < 0.0) and not np.any(d2 < cutoff):
        d2[np.logical_and(0 > d2, d2 > cutoff)] = 0.0
    return np.sqrt(np.array(d2))


def free_energy_differences(u_kn, N_k, f_k, method="svd-ew"):
    """``Delta_f``, ``dDelta_f``, ``Theta`` as ``MBAR.compute_free_energy_differences``
    (pymbar/mbar.py:683, :701-703, :722-729) would return them for converged ``f_k``."""
    Delta_f = np.array(f_k - np.vstack(f_k))
    W = mbar_W_nk(u_kn, N_k, f_k)
    Theta = covariance_theta(W, N_k, method=method)
    return Delta_f, error_of_differences(Theta), Theta


# --------------------------------------------------------------------------------------------
# Sharded restatement: what one rank contributes, and how the reduced sums give g, H, f_sci.
# (identities of SURVEY.md 2a k3/k4/k7: g_k = N_k (s_k - 1), f_sci = f - log s_k,
#  H = diag(N s) - N N^T o G with s_k = sum_n W_nk, G = W^T W)
# --------------------------------------------------------------------------------------------
def shard_partials(u_shard, N_k, f_k, want_gram=False):
    """Partial sums over a column shard: ``psum_k = sum_n N_k W_nk``, ``sum_n logden_n`` and,
    optionally, ``gram_ij = sum_n (N_i W_ni)(N_j W_nj)``.  ``N_k`` is the GLOBAL count vector."""
    N_k = np.asarray(N_k, dtype=np.float64)
    logden_n = log_denominator(u_shard, N_k, f_k)
    P = N_k * np.exp(f_k - u_shard.T - logden_n[:, np.newaxis])  # (n, K), rows sum to 1
    out = dict(psum=P.sum(0), sumlogden=float(np.sum(logden_n)))
    if want_gram:
        out["gram"] = P.T @ P
    return out


def partials_to_quantities(N_k, f_k, psum, gram=None):
    """Gradient, SCI update and Hessian from all-reduced partial sums (sampled states only)."""
    N_k = np.asarray(N_k, dtype=np.float64)
    g = psum - N_k
    f_sci = f_k - np.log(psum / N_k)
    H = None
    if gram is not None:
        H = np.diag(psum) - gram
    return g, f_sci, H
