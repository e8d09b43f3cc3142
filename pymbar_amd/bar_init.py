"""``initialize="BAR"`` for :class:`pymbar_amd.MBAR`.

Host-side, O(N_k + N_l) per adjacent pair of sampled states: an initial guess for the solver built by chaining
Bennett acceptance ratio estimates between consecutive sampled states.  Mirrors the behaviour of
``MBAR._initialize_with_bar`` (pymbar/mbar.py:1936-1988), which calls ``other_estimators.bar(method="bisection",
relative_tolerance=1e-5, maximum_iterations=100, compute_uncertainty=False)`` (other_estimators.py:156-420) on
the work values of each pair; the bracket comes from the one-sided exponential averages
(other_estimators.py:614-617).  The implicit BAR equation is the one of other_estimators.py:56-153, written here
with ``logaddexp`` (log of the Fermi function) instead of the explicit max-shift.
"""
import logging

import numpy as np
from scipy.special import logsumexp

logger = logging.getLogger(__name__)


from .utils import BoundsError as BarBoundsError  # noqa: E402  (pymbar's own classes when pymbar is installed)
from .utils import ConvergenceError as BarConvergenceError  # noqa: E402


def exp_delta_f(w):
    """One-sided exponential average ``-ln <exp(-w)>`` (other_estimators.py:614-617)."""
    w = np.asarray(w, dtype=np.float64)
    return -(logsumexp(-w) - np.log(float(w.size)))


def bar_zero(w_F, w_R, DeltaF):
    """Function whose root in ``DeltaF`` is the BAR estimate (other_estimators.py:56-153):
    ``ln sum_F f(M + w_F - DeltaF) - ln sum_R f(-(M - w_R - DeltaF))`` with ``f(x) = 1/(1 + e^x)`` and
    ``M = ln(T_F / T_R)``;  ``ln f(x) = -logaddexp(0, x)`` never overflows."""
    w_F = np.asarray(w_F, dtype=np.float64)
    w_R = np.asarray(w_R, dtype=np.float64)
    M = np.log(float(w_F.size) / float(w_R.size))
    # fast path: the Fermi functions summed directly (one exponential per work value; e^x = inf gives f = 0); only when a sum
    # underflows towards zero / the denormal range -- (almost) no overlap at this DeltaF -- the log-space form below is needed
    with np.errstate(over="ignore"):
        s_F = np.sum(1.0 / (1.0 + np.exp(M + w_F - DeltaF)))
        s_R = np.sum(1.0 / (1.0 + np.exp(-(M - w_R - DeltaF))))
    floor = np.finfo(np.float64).tiny / np.finfo(np.float64).eps  # below it a sum is made of denormals: log(s) has lost digits
    if s_F > floor and s_R > floor and np.isfinite(s_F) and np.isfinite(s_R):
        return np.log(s_F) - np.log(s_R)
    log_f_F = -np.logaddexp(0.0, M + w_F - DeltaF)
    log_f_R = -np.logaddexp(0.0, -(M - w_R - DeltaF))
    return logsumexp(log_f_F) - logsumexp(log_f_R)


def bar_bisection(w_F, w_R, relative_tolerance=1.0e-5, maximum_iterations=100):
    """BAR free energy difference by bisection, with the bracketing, stopping rule and failure modes of
    ``other_estimators.bar(method="bisection")`` (other_estimators.py:253-372)."""
    upper = exp_delta_f(w_F)
    lower = -exp_delta_f(w_R)
    f_upper = bar_zero(w_F, w_R, upper)
    f_lower = bar_zero(w_F, w_R, lower)
    if np.isnan(f_upper) or np.isnan(f_lower):
        logger.warning("BAR is likely to be inaccurate because of poor overlap; guessing a free energy difference of 0.")
        return 0.0
    while f_upper * f_lower > 0:  # widen until the signs differ (:276-285)
        mid = (upper + lower) / 2
        upper = upper - max(abs(upper - mid), 0.1)
        lower = lower + max(abs(lower - mid), 0.1)
        f_upper = bar_zero(w_F, w_R, upper)
        f_lower = bar_zero(w_F, w_R, lower)
    delta = 0.0
    relative_change = np.inf
    iteration = 0
    for iteration in range(maximum_iterations + 1):
        delta_old = delta
        delta = (upper + lower) / 2
        f_new = bar_zero(w_F, w_R, delta)
        if delta == 0.0:
            break
        relative_change = abs((delta - delta_old) / delta)
        if iteration > 0 and relative_change < relative_tolerance:
            break
        if f_upper * f_new < 0:
            lower, f_lower = delta, f_new
        elif f_lower * f_new <= 0:
            upper, f_upper = delta, f_new
        else:
            raise BarBoundsError("Cannot determine bound on free energy")
    if iteration >= maximum_iterations:
        raise BarConvergenceError(
            "Did not converge to within specified tolerance. max_delta = {:f}, TOLERANCE = {:f}, MAX_ITS = {:d}".format(
                relative_change, relative_tolerance, maximum_iterations))
    return delta


def initialize_with_bar(u_kn, N_k, x_kindices, f_k_init=None):
    """Chain BAR estimates over consecutive sampled states (pymbar/mbar.py:1936-1988).  Only useful when the
    states are in order.  Returns the (K,) initial guess (not yet shifted to f_0 = 0)."""
    N_k = np.asarray(N_k)
    K = len(N_k)
    order = np.where(N_k > 0)[0]
    f_k_init = np.zeros(K) if f_k_init is None else np.array(f_k_init, dtype=np.float64)
    # the samples of every state, grouped once (the reference builds a boolean mask over all N samples per pair, :1958-1967)
    from .utils import state_index_groups

    groups = state_index_groups(x_kindices, K)

    def samples_of(k):  # a slice (a view of the row) in the default layout, an index array otherwise
        g = groups[k]
        return slice(g.start, g.stop) if isinstance(g, range) else g

    for k, l in zip(order[:-1], order[1:]):
        from_k = samples_of(k)
        from_l = samples_of(l)
        w_F = u_kn[l, from_k] - u_kn[k, from_k]
        w_R = u_kn[k, from_l] - u_kn[l, from_l]
        if len(w_F) > 0 and len(w_R) > 0:
            try:
                f_k_init[l] = f_k_init[k] + bar_bisection(w_F, w_R, relative_tolerance=1.0e-5, maximum_iterations=100)
            except BarConvergenceError:
                logger.warning("WARNING: BAR did not converge to within tolerance")
                f_k_init[l] = f_k_init[k]
        else:
            f_k_init[l] = 0
    return f_k_init
