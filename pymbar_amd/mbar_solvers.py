"""MI355X-backed drop-in for ``pymbar.mbar_solvers`` (the solver layer L2 + the array math L1).

Same exported names, argument meaning, return conventions and error behaviour as the reference
module (pymbar/mbar_solvers.py, 1017 lines), but every sweep over the K x N reduced-potential
matrix runs in ``libmbar_hip.so`` on the GPU (include/mbar_hip.h).  There is no numpy fallback: on
a machine without the library or without a gfx950 device these functions raise
``pymbar_amd._lib.BackendUnavailable``.

``u_kn`` may be a numpy array (uploaded on first sight and kept in a small cache of device copies, so that the reference's
unchanged ``MBAR`` class uploads its matrix once per object, not once per call; a copy is re-used only while a digest of the
whole host array still matches, so an array edited in place between two calls is simply uploaded again -- see ``_ResidentCache``) or a resident
:class:`pymbar_amd.device.DeviceMatrix` (or any object with the same methods: the tests drive the
protocol logic below with a CPU stand-in built from the oracle).

Identities used: with ``psum_k = sum_n N_k W_nk`` and ``gram = sum_n p p^T``,
``p_nk = N_k W_nk``:
    mbar_gradient            = psum - N_k                      (mbar_solvers.py:284-292)
    self_consistent_update   = -lognum (log space, all states) (mbar_solvers.py:231-242)
    mbar_objective           = sum_n logden_n - N_k . f_k      (mbar_solvers.py:327-338)
    mbar_hessian             = diag(psum) - gram               (mbar_solvers.py:395-411)
"""
import copy
import logging
import os
import warnings

import numpy as np
import scipy.optimize

from .utils import ParameterError, check_w_sums, ensure_type

logger = logging.getLogger(__name__)

# API parity with the reference's backend switch (mbar_solvers.py:14-97): this backend never jits.
use_jit = False


def _setup_jax_acceleration():
    """Parse ``PYMBAR_DISABLE_JAX`` exactly like mbar_solvers.py:17-19 (kept for API parity)."""
    return os.environ.get("PYMBAR_DISABLE_JAX", "").lower() in ("true", "yes", "1")


force_no_jax = _setup_jax_acceleration()

# Solver protocols: ordered stages, each a dict(method, [tol], [continuation], [options]).
# Same content as mbar_solvers.py:102-117.
JAX_SOLVER_PROTOCOL = (
    dict(method="BFGS", continuation=True),
    dict(method="adaptive", options=dict(min_sc_iter=0)),
)
DEFAULT_SOLVER_PROTOCOL = (
    dict(method="hybr", continuation=True),
    dict(method="adaptive", options=dict(min_sc_iter=0)),
)
ROBUST_SOLVER_PROTOCOL = (
    dict(method="adaptive", options=dict(maxiter=1000)),
    dict(method="L-BFGS-B", options=dict(maxiter=1000)),
)
BOOTSTRAP_SOLVER_PROTOCOL = (dict(method="adaptive", options=dict(min_sc_iter=0)),)
# Extension (SURVEY.md 3.3 / BASELINE.json config 2): pure self-consistent iteration on the device.
SCI_SOLVER_PROTOCOL = (dict(method="self-consistent-iteration"),)

scipy_minimize_options = ["L-BFGS-B", "dogleg", "CG", "BFGS", "Newton-CG", "TNC", "trust-ncg", "trust-krylov",
                          "trust-exact", "SLSQP"]
scipy_nohess_options = ["L-BFGS-B", "BFGS", "CG", "TNC", "SLSQP"]
scipy_root_options = ["hybr", "lm"]


# --------------------------------------------------------------------------------------------
# matrix handles
# --------------------------------------------------------------------------------------------
def _is_handle(u_kn):
    return hasattr(u_kn, "eval") and hasattr(u_kn, "set_Nk")


class _ResidentCache:
    """Device copies of recently used HOST matrices, so that the literal drop-in -- the reference's unchanged ``MBAR`` class,
    which calls this module several times with the same ``self.u_kn`` (mbar.py:413 ``solve_mbar_for_all_states``, :455
    ``mbar_log_W_nk``, :910 again per expectation) -- uploads the matrix ONCE instead of once per call.

    The functions of this module stay pure functions of their arguments (mbar_solvers.py:260-292: every call of the reference
    reads the array it is handed): a device copy is re-used only when a digest of EVERY byte of the host array
    (``mbar_host_digest``: 128 bits, all host cores, memory speed -- ~0.1 s for config 3's 10 GB against the 0.2 s upload it
    saves; a change of one element always changes it) equals the digest taken when the copy was uploaded.  An in-place edit
    of any kind therefore costs a fresh upload, never a stale answer.
    An entry in use is never closed: handles are reference-counted (an evicted entry is closed by its last user), every cache
    operation runs under a lock, and a handle -- a context is not thread-safe -- is used by one thread at a time.
    Bounded: ``PYMBAR_AMD_RESIDENT_CACHE`` entries (default 2: the unchanged class alternates between ``self.u_kn`` and one
    bootstrap replicate's matrix; 0 switches the cache off) and ``PYMBAR_AMD_RESIDENT_CACHE_GB`` (default 12: one matrix of
    config 3's size, no more -- a drop-in must not sit on a large part of a shared GPU) of device memory; least recently used first out;
    :func:`drop_resident_cache` releases everything."""

    class Entry:
        __slots__ = ("handle", "nbytes", "users", "evicted", "lock")

        def __init__(self, handle, nbytes):
            import threading

            self.handle, self.nbytes, self.users, self.evicted = handle, nbytes, 0, False
            self.lock = threading.RLock()  # (one thread at a time on the context; re-entrant for nested calls of one thread)

    def __init__(self):
        import threading
        from collections import OrderedDict

        self.entries = OrderedDict()  # key -> Entry
        self.uploads = 0              # (counted for the boundary test)
        self.hits = 0
        self.mutex = threading.RLock()

    @staticmethod
    def limits():
        return (int(os.environ.get("PYMBAR_AMD_RESIDENT_CACHE", "2")),
                float(os.environ.get("PYMBAR_AMD_RESIDENT_CACHE_GB", "12")) * 1e9)

    @staticmethod
    def digest(a):
        """128-bit digest of every byte of the C-contiguous array ``a`` (the C library's threaded hash; ``hashlib`` when the
        library cannot be loaded -- only the stand-in device of the CPU tests gets there)."""
        try:
            from . import _lib

            return _lib.host_digest(a)
        except Exception:  # pragma: no cover - library not built
            import hashlib

            return hashlib.blake2b(a.reshape(-1).view(np.uint8), digest_size=16).digest()

    def key_of(self, a):
        return (a.ctypes.data, a.shape, a.strides, self.digest(a))

    def acquire(self, u_kn):
        """A resident entry for the host array (its ``users`` count raised, its lock held), or ``None`` when it cannot be
        cached (the caller uploads a temporary)."""
        max_entries, max_bytes = self.limits()
        a = u_kn
        if (max_entries <= 0 or not isinstance(a, np.ndarray) or a.ndim != 2 or a.dtype != np.float64
                or not a.flags.c_contiguous or a.size == 0 or a.nbytes > max_bytes):
            return None
        key = self.key_of(a)
        with self.mutex:
            entry = self.entries.get(key)
            if entry is not None:
                self.entries.move_to_end(key)
                self.hits += 1
            else:
                # a different content behind the same address / shape: the old copy can never be hit again
                for old_key in [k for k in self.entries if k[:3] == key[:3]]:
                    self._evict(old_key)
                from .device import DeviceMatrix  # deferred: importing this module must not need a GPU

                entry = self.Entry(DeviceMatrix.from_host(a), a.nbytes)
                self.uploads += 1
                self.entries[key] = entry
                for old_key in list(self.entries):
                    if len(self.entries) <= max_entries and sum(e.nbytes for e in self.entries.values()) <= max_bytes:
                        break
                    if old_key != key:
                        self._evict(old_key)
            entry.users += 1
        entry.lock.acquire()
        return entry

    def release(self, entry):
        entry.lock.release()
        with self.mutex:
            entry.users -= 1
            if entry.evicted and entry.users == 0:
                entry.handle.close()

    def _evict(self, key):
        entry = self.entries.pop(key)
        entry.evicted = True
        if entry.users == 0:
            entry.handle.close()

    def clear(self):
        with self.mutex:
            for key in list(self.entries):
                self._evict(key)


_resident_cache = _ResidentCache()


def drop_resident_cache():
    """Release the device copies this module keeps of recently used host matrices (see :class:`_ResidentCache`)."""
    _resident_cache.clear()


class _Resident:
    """Context manager yielding a device-resident handle for ``u_kn``: handles pass through untouched, a float64 C-contiguous
    host array gets (or re-uses) a cached device copy, anything else is uploaded for the duration of the block."""

    def __init__(self, u_kn):
        self.u_kn = u_kn
        self.owned = None
        self.entry = None

    def __enter__(self):
        if _is_handle(self.u_kn):
            return self.u_kn
        self.entry = _resident_cache.acquire(self.u_kn)
        if self.entry is not None:
            return self.entry.handle
        from .device import DeviceMatrix  # deferred: importing this module must not need a GPU

        self.owned = DeviceMatrix.from_host(self.u_kn)
        return self.owned

    def __exit__(self, *exc):
        if self.entry is not None:
            _resident_cache.release(self.entry)
            self.entry = None
        if self.owned is not None:
            self.owned.close()


def validate_inputs(u_kn, N_k, f_k):
    """Type/shape checks of mbar_solvers.py:174-203: float64 C-contiguous ``u_kn`` (K, N), float
    ``N_k`` (K,), float ``f_k`` (K,).  A device handle passes through unchanged."""
    if _is_handle(u_kn):
        n_states = u_kn.shape[0]
    else:
        n_states, n_samples = u_kn.shape
        u_kn = ensure_type(u_kn, "float", 2, "u_kn or Q_kn", shape=(n_states, n_samples))
    N_k = ensure_type(N_k, "float", 1, "N_k", shape=(n_states,), warn_on_cast=False)
    f_k = ensure_type(f_k, "float", 1, "f_k", shape=(n_states,))
    return u_kn, N_k, f_k


def _prep(h, N_k, f_k):
    N_k = np.asarray(N_k, dtype=np.float64)
    f_k = np.asarray(f_k, dtype=np.float64)
    if N_k.shape != (h.shape[0],) or f_k.shape != (h.shape[0],):
        raise ValueError(f"N_k and f_k must have shape ({h.shape[0]},)")
    h.set_Nk(N_k)
    return N_k, f_k


# --------------------------------------------------------------------------------------------
# L1 array math
# --------------------------------------------------------------------------------------------
def self_consistent_update(u_kn, N_k, f_k, states_with_samples=None):
    """Improved guess of the free energies, Eq. C3 (mbar_solvers.py:206-257).

    Only ``states_with_samples`` (default: all) enter the denominator and are returned."""
    with _Resident(u_kn) as h:
        N_k = np.asarray(N_k, dtype=np.float64)
        f_k = np.asarray(f_k, dtype=np.float64)
        if states_with_samples is None:
            h.set_Nk(N_k)
            return -1.0 * h.lognum(f_k)
        mask = np.zeros_like(N_k)
        mask[states_with_samples] = N_k[states_with_samples]
        h.set_Nk(mask)
        return -1.0 * h.lognum(f_k)[states_with_samples]


def mbar_gradient(u_kn, N_k, f_k):
    """Gradient of the MBAR objective, Eq. C6 (mbar_solvers.py:260-292)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        psum, _, _ = h.eval(f_k)
        return psum[0] - N_k


def mbar_objective(u_kn, N_k, f_k):
    """``sum_n logden_n - N_k . f_k`` (mbar_solvers.py:295-338)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        _, sld, _ = h.eval(f_k)
        return sld[0] - np.dot(N_k, f_k)


def mbar_objective_and_gradient(u_kn, N_k, f_k):
    """Objective and gradient from one sweep (mbar_solvers.py:341-392)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        psum, sld, _ = h.eval(f_k)
        return sld[0] - np.dot(N_k, f_k), psum[0] - N_k


def mbar_hessian(u_kn, N_k, f_k):
    """Hessian of the objective, Eq. C9 (mbar_solvers.py:395-436); the W^T W contraction runs on
    the fp64 matrix cores."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        psum, _, gram = h.eval(f_k, gram=True)
        return np.diag(psum[0]) - gram


def mbar_log_W_nk(u_kn, N_k, f_k):
    """Normalised log weights, Eq. 9, shape (N, K), F-ordered like the reference's result
    (mbar_solvers.py:439-473)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        return h.logw_kn(f_k).T


def mbar_W_nk(u_kn, N_k, f_k):
    """``exp(mbar_log_W_nk)`` (mbar_solvers.py:476-507), the exponential taken on the device (``mbar_w``)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        return h.w_kn(f_k).T


def precondition_u_kn(u_kn, N_k, f_k):
    """``u_kn`` shifted per sample so that the objective is zero at ``f_k`` (mbar_solvers.py:697-735).

    Returned as a new host array for API parity.  The solvers below never materialise it: a per-sample
    offset kept on the device gives the same objective (see ``DeviceMatrix.set_objective_offset``)."""
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        logden = h.logden(f_k)
        u_host = u_kn if not _is_handle(u_kn) else h.to_host()
        return np.asarray(u_host, dtype=np.float64) + (logden - np.dot(N_k, f_k) / N_k.sum())


# --------------------------------------------------------------------------------------------
# the adaptive loop
# --------------------------------------------------------------------------------------------
def adaptive(u_kn, N_k, f_k, tol=1.0e-8, options=None):
    """Newton-Raphson / self-consistent iteration keeping, per iteration, the candidate with the
    smaller gradient norm (mbar_solvers.py:510-667).  The loop runs in ``mbar_solve_adaptive``:
    one MFMA Gram sweep + one two-candidate sweep per iteration.

    Returns ``dict(success, message, x)`` like the reference, plus ``nr_iter``, ``sci_iter``,
    ``iterations``.  States with ``N_k == 0`` are ignored (their ``f_k`` is returned unchanged)."""
    if options is None:
        options = dict()
    options.setdefault("verbose", False)
    options.setdefault("maxiter", 10000)
    options.setdefault("print_warning", False)
    options.setdefault("gamma", 1.0)
    options.setdefault("min_sc_iter", 2)
    verbose = options["verbose"] is True
    if verbose:
        logger.info("Determining dimensionless free energies by Newton-Raphson / self-consistent iteration.")
    if tol < 4.0 * np.finfo(float).eps:
        logger.info("Tolerance may be too close to machine precision to converge.")
    maxiter = int(options["maxiter"])
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        x, res = h.solve_adaptive(f_k, tol=tol, maxiter=maxiter, min_sc_iter=int(options["min_sc_iter"]),
                                  gamma=float(options["gamma"]), history_rows=maxiter if verbose else 0)
    if verbose:
        # the reference's per-iteration lines (mbar_solvers.py:598-624), replayed from the rows the device loop recorded:
        # {choice, |g_sci|, |g_nr|, max_delta} per iteration; its two wordings of a self-consistent choice depend on the running
        # count AFTER the increment (:609-620)
        min_sc_iter, sci_seen = int(options["min_sc_iter"]), 0
        for it, row in enumerate(res["history"]):
            logger.info("self consistent iteration gradient norm is %10.5g, Newton-Raphson gradient norm is %10.5g"
                        % (row[1], row[2]))
            if row[0] == 0:
                sci_seen += 1
                if sci_seen < min_sc_iter:
                    logger.info(f"Choosing self-consistent iteration on iteration {it:d} because min_sci_iter={min_sc_iter:d}")
                else:
                    logger.info(f"Choosing self-consistent iteration for lower gradient on iteration {it:d}")
            else:
                logger.info(f"Newton-Raphson used on iteration {it:}")
    if res["success"]:
        message = "Convergence achieved by change in f with respect to previous guess."
        if verbose:
            logger.info(f"Converged to tolerance of {res['max_delta']:e} in {res['iterations']:d} iterations.")
            logger.info(f"Of {res['iterations']:d} iterations, {res['nr_iter']:d} were Newton-Raphson iterations "
                        f"and {res['sci_iter']:d} were self-consistent iterations")
            if np.all(x == 0.0):
                logger.info("WARNING: All f_k appear to be zero.")
    else:
        message = "Did not converge."
        logger.warning("WARNING: Did not converge to within specified tolerance.")
        if maxiter <= 0:
            logger.warning(f"No iterations ran be cause maximum_iterations was <= 0 ({maxiter})!")
        else:
            # (the reference prints the index of the last iteration, mbar_solvers.py:655-657: one less than the count)
            logger.warning(f"max_delta = {res['max_delta']:e}, tol = {tol:e}, maximum_iterations = {maxiter:d}, "
                           f"iterations completed = {max(0, res['iterations'] - 1):d}")
    results = dict(success=res["success"], message=message, x=x, nr_iter=res["nr_iter"], sci_iter=res["sci_iter"],
                   iterations=res["iterations"], wall_ms=res["wall_ms"])
    if res.get("psum") is not None and np.all(np.isfinite(res["psum"])):
        results["psum"] = res["psum"]  # (extension: sum_n N_k W_nk at x, from the solver's last sweep)
    return results


def self_consistent_iteration(u_kn, N_k, f_k, tol=1.0e-12, options=None):
    """Extension: the loop ``f <- self_consistent_update(f); f -= f[0]`` until the relative change of
    mbar_solvers.py:627-631 drops below ``tol``, run device-resident (``mbar_solve_sci``)."""
    options = dict() if options is None else options
    maxiter = int(options.get("maxiter", 10000))
    with _Resident(u_kn) as h:
        N_k, f_k = _prep(h, N_k, f_k)
        x, res = h.solve_sci(f_k, tol=tol, maxiter=maxiter)
    message = "Convergence achieved by change in f with respect to previous guess." if res["success"] else "Did not converge."
    if not res["success"]:
        logger.warning("WARNING: Did not converge to within specified tolerance.")
    return dict(success=res["success"], message=message, x=x, nr_iter=0, sci_iter=res["iterations"],
                iterations=res["iterations"], wall_ms=res["wall_ms"])


# --------------------------------------------------------------------------------------------
# solver drivers
# --------------------------------------------------------------------------------------------
def _solve_once_resident(h, N_k, f_k, sampled, method, tol, options):
    """One protocol stage on a resident matrix.  ``N_k`` (float, full length, zeros for unsampled
    states), ``f_k`` full length with ``f_k[sampled[0]] == 0``; the unknowns are ``f_k[sampled[1:]]``
    (the reduced coordinates of mbar_solvers.py:791, 795-818)."""
    K = h.shape[0]
    free = np.asarray(sampled[1:], dtype=np.int64)
    sub = np.ix_(free, free)
    h.set_Nk(N_k)

    def full(x):
        f = f_k.copy()
        f[free] = x
        return f

    def grad(x):
        psum, _, _ = h.eval(full(x))
        return (psum[0] - N_k)[free]

    def grad_and_obj(x):
        f = full(x)
        psum, sld, _ = h.eval(f, use_offset=True)
        # preconditioned objective (mbar_solvers.py:793): sum_n (logden_n(f) - logden_n(f0)) + N.f0 - N.f
        return sld[0] + obj_const - np.dot(N_k, f), (psum[0] - N_k)[free]

    def hess(x):
        psum, _, gram = h.eval(full(x), gram=True)
        return (np.diag(psum[0]) - gram)[sub]

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        if method == "adaptive":
            results = adaptive(h, N_k, f_k, tol=tol, options=options)
            f_out = results["x"]
        elif method == "self-consistent-iteration":
            results = self_consistent_iteration(h, N_k, f_k, tol=tol, options=options)
            f_out = results["x"]
        elif method in scipy_minimize_options:
            h.set_objective_offset(f_k)
            obj_const = np.dot(N_k, f_k)
            try:
                results = scipy.optimize.minimize(
                    grad_and_obj, f_k[free], jac=True, hess=None if method in scipy_nohess_options else hess,
                    method=method, tol=tol, options=options)
            finally:
                h.set_objective_offset(None)
            f_out = full(results["x"])
        elif method in scipy_root_options:
            results = scipy.optimize.root(grad, f_k[free], jac=hess, method=method, tol=tol, options=options)
            f_out = full(results["x"])
        else:
            raise ParameterError(f"Method {method} for solution of free energies not recognized")

    # surface runtime warnings; anything other than "Unknown solver options" triggers the weight check
    # of mbar_solvers.py:861-881
    can_ignore = True
    for msg in w:
        if "Unknown solver options" in str(msg.message):
            continue
        warnings.showwarning(msg.message, msg.category, msg.filename, msg.lineno, msg.file, "")
        can_ignore = False
    if not can_ignore:
        psum, _, _ = h.eval(f_out)
        colsum = np.where(N_k > 0, psum[0] / np.where(N_k > 0, N_k, 1.0), 1.0)
        check_w_sums(colsum[sampled], abs(psum[0].sum() - N_k.sum()) / max(1.0, N_k.sum()))
        logger.warning("MBAR weights converged within tolerance, despite the SciPy Warnings. Please validate your results.")
    return f_out, results


def solve_mbar_once(u_kn_nonzero, N_k_nonzero, f_k_nonzero, method="adaptive", tol=1e-12, continuation=None,
                    options=None):
    """Solve the MBAR equations with one solver (mbar_solvers.py:738-883).

    Works in the gauge ``f[0] = 0``; returns ``(f_k, results)``.  ``method`` is "adaptive", "hybr",
    "lm", any gradient-based ``scipy.optimize.minimize`` method, or the extension
    "self-consistent-iteration"."""
    u_kn_nonzero, N_k_nonzero, f_k_nonzero = validate_inputs(u_kn_nonzero, np.asarray(N_k_nonzero), f_k_nonzero)
    f_k_nonzero = f_k_nonzero - f_k_nonzero[0]
    N_k_nonzero = 1.0 * N_k_nonzero
    options = dict() if options is None else options
    # The reference does not check N_k here (its caller passes sampled states only, mbar_solvers.py:1002-1006); states
    # with N_k <= 0 are tolerated like it tolerates them: they carry no weight in any sum, are left out of the unknowns
    # and keep the f_k they came with.
    sampled = np.where(N_k_nonzero > 0)[0]
    if sampled.size == 0:
        raise ParameterError("solve_mbar_once needs at least one state with N_k > 0")
    N_k_nonzero = np.where(N_k_nonzero > 0, N_k_nonzero, 0.0)
    f_k_nonzero = f_k_nonzero - f_k_nonzero[sampled[0]]  # gauge on the first state that has samples
    with _Resident(u_kn_nonzero) as h:
        return _solve_once_resident(h, N_k_nonzero, f_k_nonzero, sampled, method, tol, options)


def _gnorm(h, N_k, f, sampled):
    psum, _, _ = h.eval(f)
    return float(np.linalg.norm((psum[0] - N_k)[sampled]))


def _solve_protocol_resident(h, N_k, f_k, sampled, solver_protocol):
    """Run the stages in order, stop at the first success, otherwise keep the result with the
    smallest gradient norm (mbar_solvers.py:928-974)."""
    if solver_protocol is None:
        solver_protocol = DEFAULT_SOLVER_PROTOCOL
    all_fks, all_gnorms, all_results = [], [], []
    f_cur = f_k
    results = dict(success=False)
    for solver in solver_protocol:
        stage = dict(solver)
        method = stage.pop("method", "adaptive")
        tol = stage.pop("tol", 1e-12)
        continuation = stage.pop("continuation", None)
        options = stage.pop("options", None)
        options = dict() if options is None else options
        f_stage = f_cur - f_cur[sampled[0]]
        f_result, results = _solve_once_resident(h, N_k, f_stage, sampled, method, tol, options)
        all_fks.append(f_result)
        # gradient norm at the stage's result (mbar_solvers.py:939): the adaptive loop hands its per-state sums back, any other
        # method costs one evaluation sweep
        ps = results.get("psum") if isinstance(results, dict) else None
        all_gnorms.append(float(np.linalg.norm((ps - N_k)[sampled])) if ps is not None else _gnorm(h, N_k, f_result, sampled))
        all_results.append(results)
        if results["success"]:
            logger.info(f"Reached a solution to within tolerance with {method}")
            break
        logger.warning(f"Failed to reach a solution to within tolerance with {method}: trying next method")
        logger.info(f"Ending gnorm of method {method} = {all_gnorms[-1]:e}")
        if continuation:
            f_cur = f_result
            logger.info("Will continue with results from previous method")
    if results["success"]:
        logger.info("Solution found within tolerance!")
        best, best_gnorm = all_fks[-1], all_gnorms[-1]
    else:
        i_best = int(np.argmin(all_gnorms))
        logger.warning("No solution found to within tolerance.")
        best, best_gnorm = all_fks[i_best], all_gnorms[i_best]
        logger.warning(f"The solution with the smallest gradient {best_gnorm:e} norm is "
                       f"{solver_protocol[i_best]['method']}")
        logger.warning("Please exercise caution with this solution and consider alternative methods or a different tolerance.")
    logger.info(f"Final gradient norm: {best_gnorm:.3g}")
    best_psum = None
    if results["success"] and isinstance(all_results[-1], dict):
        best_psum = all_results[-1].get("psum")
    return best, all_results, best_psum


def solve_mbar(u_kn_nonzero, N_k_nonzero, f_k_nonzero, solver_protocol=None):
    """Solve with a sequence of solvers (mbar_solvers.py:886-974).  Returns ``(f_k, all_results)``."""
    u_kn_nonzero, N_k_f, f_k_nonzero = validate_inputs(u_kn_nonzero, np.asarray(N_k_nonzero), f_k_nonzero)
    N_k_f = 1.0 * N_k_f
    with _Resident(u_kn_nonzero) as h:
        sampled = np.arange(h.shape[0])
        return _solve_protocol_resident(h, N_k_f, f_k_nonzero - f_k_nonzero[0], sampled, solver_protocol)[:2]


def solve_mbar_for_all_states(u_kn, N_k, f_k, states_with_samples, solver_protocol):
    """Solve on the states with samples, then one all-state self-consistent update gives the
    unsampled states and ``f_k[0]`` is re-zeroed (mbar_solvers.py:977-1017).

    Unlike the reference no ``u_kn[states_with_samples]`` copy is made: unsampled rows are masked by
    ``N_k = 0`` on the device."""
    states_with_samples = np.asarray(states_with_samples, dtype=np.int64)
    N_k = np.asarray(N_k)
    f_k = np.array(f_k, dtype=np.float64)
    with _Resident(u_kn) as h:
        Nf = np.asarray(N_k, dtype=np.float64)
        if len(states_with_samples) == 1:
            f_k[states_with_samples] = 0.0
        else:
            f_start = f_k.copy()
            f_start[states_with_samples] -= f_start[states_with_samples[0]]
        psum_solved = None
        if len(states_with_samples) != 1:
            f_solved, _, psum_solved = _solve_protocol_resident(h, Nf, f_start, states_with_samples,
                                                                copy.deepcopy(solver_protocol) if solver_protocol is not None else None)
            f_k[states_with_samples] = f_solved[states_with_samples]
        h.set_Nk(Nf)
        if np.all(Nf > 0):
            # every state is sampled: the all-state update -lognum_k equals f_k - log(psum_k / N_k) -- with the per-state sums the
            # adaptive loop hands back at its solution no sweep at all, otherwise one single-candidate sweep (instead of the
            # log-denominator sweep + the log-space reduction sweep)
            if psum_solved is not None and np.all(psum_solved > 0):
                f_k = f_k - np.log(psum_solved / Nf)
            else:
                psum, _, _ = h.eval(f_k)
                f_k = f_k - np.log(psum[0] / Nf)
        else:
            f_k = -1.0 * h.lognum(f_k)
    f_k -= f_k[0]
    return f_k
