"""``MBAR(u_kn, N_k, solver_protocol=...)`` on the MI355X solver path.

Host-side mirror of the part of ``pymbar.mbar.MBAR`` that touches the hot path (SURVEY.md 3.1,
3.4): construction = solve (pymbar/mbar.py:85-465), ``Log_W_nk`` / ``W_nk`` (:455-478),
``compute_effective_sample_number`` (:495-560), ``compute_overlap`` (:563-617) and
``compute_free_energy_differences`` with the asymptotic covariance ``Theta`` (:620-729, :1687-1864).
The reduced-potential matrix is uploaded once and stays resident; ``W^T W`` is contracted on the
fp64 matrix cores and only K x K linear algebra (eigh / pinv) runs on the host.

The ``Log_W_nk`` consumers (``compute_expectations*``, ``compute_perturbed_free_energies``,
``compute_entropy_and_enthalpy``) live in :mod:`pymbar_amd.expectations` and are bound as
methods below.  Not mirrored (not on the K x N solver path): FES, timeseries, the other estimators.
"""
import copy
import contextlib
import logging
import os

import numpy as np

from . import expectations as _expectations
from . import mbar_solvers
from .mbar_solvers import (BOOTSTRAP_SOLVER_PROTOCOL, DEFAULT_SOLVER_PROTOCOL, JAX_SOLVER_PROTOCOL,
                           ROBUST_SOLVER_PROTOCOL)
from .utils import ParameterError, check_w_sums, kln_to_kn

logger = logging.getLogger(__name__)


def _resolve_protocol(protocol, default, robust, maximum_iterations, verbose, pname):
    """Protocol resolution of pymbar/mbar.py:370-406, on a private copy (the reference mutates the
    module-level constants in place, which leaks options between MBAR objects)."""
    if protocol is None or protocol == "default":
        protocol = default
    elif protocol == "robust":
        protocol = robust
    elif protocol == "jax":
        protocol = JAX_SOLVER_PROTOCOL
    else:
        for solver in protocol:
            if not isinstance(solver, dict):
                logger.warning(f"{pname} is not 'robust','default' or a tuple/list dictionaries, setting to 'default'")
                protocol = default
                break
    protocol = copy.deepcopy(tuple(protocol))
    for solver in protocol:
        solver.setdefault("options", dict())
        if solver["options"] is None:
            solver["options"] = dict()
        solver.setdefault("continuation", None)
        opts = solver["options"]
        if "maxiter" not in opts or maximum_iterations > opts["maxiter"]:
            opts["maxiter"] = maximum_iterations
        opts.setdefault("verbose", verbose)
    return protocol


@contextlib.contextmanager
def _small_blas(K):
    """A few hundred states: the eigendecomposition / pseudo-inverse / products of the covariance are K x K host matrices, and a
    BLAS that wakes every core of a large host for them spends tens of milliseconds on its threads (16-95 ms instead of 3 at
    K = 120 on a 256-thread box; 130 instead of 23 ms at 384, 267 instead of 60 at 512).  One thread while they run up to 256
    states, four beyond (measured best there: 34.7 / 26.8 / 23.3 / 23.9 ms at 1 / 2 / 4 / 8 threads for 384), if threadpoolctl is
    there to ask."""
    global _BLAS_CONTROLLER
    if K < 48:  # (below ~50 states the products are too small for the BLAS to thread at all)
        yield
        return
    if _BLAS_CONTROLLER is None:
        try:
            from threadpoolctl import ThreadpoolController

            _BLAS_CONTROLLER = ThreadpoolController()  # (scans the loaded libraries once; threadpool_limits() does it per call)
        except Exception:  # pragma: no cover - optional dependency
            _BLAS_CONTROLLER = False
    if not _BLAS_CONTROLLER:
        yield
        return
    with _BLAS_CONTROLLER.limit(limits=1 if K <= 256 else 4):
        yield


_BLAS_CONTROLLER = None


from .utils import private_copy as _private_copy  # noqa: E402


_EARLY_UPLOAD_BYTES = 8 << 20  # from this size on the private host copy of u_kn and its upload run side by side (a thread start is ~0.1 ms)


def _sample_groups(x_kindices, K):
    """The samples of every state (see :func:`pymbar_amd.utils.state_index_groups`)."""
    from .utils import state_index_groups

    return state_index_groups(x_kindices, K)


class MBAR:
    """Multistate Bennett acceptance ratio estimator; free energies are solved on construction.

    Parameters follow pymbar/mbar.py:85-99.  ``u_kn`` is (K, N) [or (K, L, N_max) ``u_kln``];
    ``N_k`` (K,) may contain zeros.  Extra keywords: ``device`` selects the GPU; ``copy=False`` keeps a read-only reference to a
    float64 C-contiguous ``u_kn`` instead of the reference's host copy (the caller must then leave the array alone);
    ``bootstrap_rng``: ``"reference"`` (default) draws the bootstrap replicates with ``numpy.random.default_rng(rseed)`` exactly as
    the reference does (mbar.py:417-449: the same ``bootstrap_rints`` and ``f_k_boots`` for the same ``rseed``), ``"device"`` draws
    them on the GPU from a counter-based stream keyed by ``rseed`` (the same statistics, deterministic under ``rseed``, not numpy's
    numbers; a replicate then costs its solve -- no pass over N integers on the host -- and ``bootstrap_rints`` is materialised on
    first access)."""

    def __init__(self, u_kn, N_k, maximum_iterations=10000, relative_tolerance=1.0e-7, verbose=False,
                 initial_f_k=None, solver_protocol=None, initialize="zeros", x_kindices=None, n_bootstraps=0,
                 bootstrap_solver_protocol=None, rseed=None, device=None, copy=True, bootstrap_rng="reference"):
        from .device import DeviceMatrix

        if bootstrap_rng not in ("reference", "device"):  # (before the upload and the main solve: a typo must not cost either)
            raise ParameterError("bootstrap_rng must be 'reference' or 'device'")
        self.N_k = np.array(N_k, dtype=np.int64)
        self.N = int(np.sum(self.N_k))
        if len(np.shape(u_kn)) == 3:
            u_kn = kln_to_kn(u_kn, N_k=self.N_k)
        # Like the reference (mbar.py:243) the object keeps its OWN host copy of the matrix by default: the device copy is
        # frozen at construction, and a caller who goes on to modify or recycle the array must not make the two diverge (the
        # host copy is read again by initialize="BAR" in the bootstrap loop, by "mean-reduced-potential" and as the default
        # ``u_kn`` / ``A_n`` of the expectation family).  ``copy=False`` (extension; for matrices of many GB, where a second host
        # copy is 10 GB and seconds) REFERENCES a float64 C-contiguous input instead, through a read-only view.
        src = np.ascontiguousarray(u_kn, dtype=np.float64)   # (a conversion already yields a private array)
        shared = src is u_kn or src.base is not None
        early_upload = None
        if copy and shared:
            if src.ndim == 2 and src.nbytes >= _EARLY_UPLOAD_BYTES and int(np.sum(self.N_k)) == src.shape[1]:
                # many GB: the private host copy (several threads, memory-bound) and the PCIe upload (one ctypes call that
                # releases the interpreter lock) both only READ the caller's array -- side by side instead of one after the
                # other (config 3: 0.08 s of copy hidden behind the 0.20 s upload)
                import threading

                early_upload = dict(dm=None, err=None, dt=0.0)

                def _upload():
                    import time as _t

                    t0 = _t.perf_counter()
                    try:
                        early_upload["dm"] = DeviceMatrix.from_host(src, device=device)
                    except BaseException as exc:  # noqa: BLE001  (re-raised on the constructor's thread below)
                        early_upload["err"] = exc
                    early_upload["dt"] = _t.perf_counter() - t0

                early_upload["thread"] = threading.Thread(target=_upload, daemon=True)
                early_upload["thread"].start()
            try:
                self.u_kn = _private_copy(src)
            except BaseException:  # (a MemoryError on a matrix of many GB: the upload beside it must not leave its device copy behind)
                if early_upload is not None:
                    early_upload["thread"].join()
                    if early_upload["dm"] is not None:
                        early_upload["dm"].close()
                raise
        elif shared:
            self.u_kn = src.view()
            self.u_kn.setflags(write=False)
        else:
            self.u_kn = src
        try:
            if self.u_kn.ndim != 2:
                raise ParameterError("u_kn must be a K x N (or K x L x N_max) array.")
            K, N = self.u_kn.shape
            if verbose:
                logger.info("K (total states) = {:d}, total samples = {:d}".format(K, N))
            if np.sum(self.N_k) != N:
                raise ParameterError(
                    "The sum of all N_k must equal the total number of samples (length of second dimension of u_kn.")
            self.K, self.N = K, N
            if x_kindices is not None:
                self.x_kindices = x_kindices
            else:
                self.x_kindices = np.repeat(np.arange(K, dtype=np.int64), self.N_k)
            self.verbose = verbose
            if rseed is None:
                rseed = np.random.randint(np.iinfo(np.int32).max)
            self.rng = np.random.default_rng(rseed)

            # same-energy state detection on <= 50 random samples, verbose only (mbar.py:273-317); the random
            # draw happens regardless of verbosity so that bootstraps are reproducible under rseed
            self.samestates = []
            maxpoint = min(50, self.N)
            indices = self.rng.choice(np.arange(self.N), maxpoint)
            if self.verbose:
                sub = self.u_kn[:, indices]
                for k in range(K):
                    for l in range(k):
                        d = sub[k] - sub[l]
                        if np.dot(d, d) < relative_tolerance:
                            self.samestates += [[k, l], [l, k]]
                            logger.warning(f"States {l:d} and {k:d} have the same energies on the dataset. They are "
                                           "therefore likely to to be the same thermodynamic state.")
                logger.info("N_k = ")
                logger.info(self.N_k)

            self.states_with_samples = np.where(self.N_k != 0)[0].astype(np.int64)
            self.K_nonzero = self.states_with_samples.size
            if verbose:
                logger.info("There are {:d} states with samples.".format(self.K_nonzero))

            self.f_k = np.zeros([K], dtype=np.float64)
            if initial_f_k is not None:
                initial_f_k = np.array(initial_f_k, dtype=np.float64)
                if initial_f_k.shape != self.f_k.shape:
                    raise ParameterError("initial_f_k must be a {:d}-dimensional np array.".format(K))
                self.f_k = initial_f_k - initial_f_k[0]
            else:
                self._initializeFreeEnergies(verbose, method=initialize)

            solver_protocol = _resolve_protocol(solver_protocol, DEFAULT_SOLVER_PROTOCOL, ROBUST_SOLVER_PROTOCOL,
                                                maximum_iterations, verbose, "solver_protocol")
            bootstrap_solver_protocol = _resolve_protocol(bootstrap_solver_protocol, BOOTSTRAP_SOLVER_PROTOCOL,
                                                          ROBUST_SOLVER_PROTOCOL, maximum_iterations, verbose,
                                                          "bootstrap_solver_protocol")

        except BaseException:
            if early_upload is not None:  # (an argument error after the upload was started: do not leave its device copy behind)
                early_upload["thread"].join()
                if early_upload["dm"] is not None:
                    early_upload["dm"].close()
            raise
        # the matrix goes to HBM once and stays there for the lifetime of the object
        self._device = device
        self._bootstrap_protocol = bootstrap_solver_protocol
        import time as _time

        _t0 = _time.perf_counter()
        if early_upload is not None:
            early_upload["thread"].join()
            if early_upload["err"] is not None:
                raise early_upload["err"]
            self._dm = early_upload["dm"]
            _dt = early_upload["dt"]
        else:
            self._dm = DeviceMatrix.from_host(self.u_kn, device=device)
            _dt = _time.perf_counter() - _t0
        self.upload_stats = dict(upload_s=_dt, upload_GBps=8.0 * K * N / _dt * 1e-9 if _dt > 0 else float("inf"))
        self.f_k = mbar_solvers.solve_mbar_for_all_states(self._dm, self.N_k, self.f_k, self.states_with_samples,
                                                          solver_protocol)

        self.n_bootstraps = 0
        # which random stream the replicates were actually drawn from: "device" only when the device stream could be used
        # (group sizes equal to N_k and a backend that has it); a request that could not be honoured is logged, not silent
        self.bootstrap_rng_used = None if n_bootstraps <= 0 else "reference"
        self._bootstrap_stream = None   # (seed, cumN, order) of the device stream, or None: numpy's, rows stored
        self._bootstrap_rints = None
        if n_bootstraps > 0 and bootstrap_rng == "device" and hasattr(self._dm, "draw_bootstrap_weights"):
            groups = _sample_groups(self.x_kindices, K)
            if all(len(groups[k]) == int(self.N_k[k]) for k in range(K)):
                self.n_bootstraps = n_bootstraps
                self.f_k_boots = np.zeros([n_bootstraps, K])
                cumN = np.concatenate(([0], np.cumsum(self.N_k))).astype(np.int64)
                default_layout = all(isinstance(g, range) for g in groups)
                order = None if default_layout else np.concatenate([np.asarray(g, dtype=np.int64) for g in groups])
                seed = int(self.rng.integers(np.iinfo(np.int64).max))   # (deterministic under rseed)
                self._bootstrap_stream = (seed, cumN, order)
                self.bootstrap_rng_used = "device"
                for b in range(n_bootstraps):
                    self._set_bootstrap_weights(self._dm, b)
                    f_k_init = self.f_k.copy()
                    if initialize == "BAR":
                        f_k_init = self._initialize_with_bar(self.u_kn[:, self._bootstrap_row(b)], f_k_init=self.f_k.copy())
                    try:
                        self.f_k_boots[b, :] = mbar_solvers.solve_mbar_for_all_states(
                            self._dm, self.N_k, f_k_init, self.states_with_samples, bootstrap_solver_protocol)
                    finally:
                        self._dm.set_sample_weights(None)
                n_bootstraps = 0  # (done)
        if n_bootstraps > 0 and bootstrap_rng == "device":
            logger.warning("bootstrap_rng='device' is not available here (x_kindices group sizes differ from N_k, or the backend has no "
                           "device stream): the replicates are drawn from numpy's stream (bootstrap_rng_used = 'reference')")
        if n_bootstraps > 0:
            self.n_bootstraps = n_bootstraps
            self.f_k_boots = np.zeros([n_bootstraps, K])
            self.bootstrap_rints = np.zeros([n_bootstraps, self.N], int)
            # the samples of every state, ascending (what np.where(x_kindices == k)[0] yields in the reference's loop,
            # mbar.py:425-431, found there K times per replicate: O(K N) each): grouped ONCE; the random stream -- one
            # rng.integers(N_k, size=N_k) per state and replicate, in state order -- is the reference's
            groups = _sample_groups(self.x_kindices, K)
            for b in range(n_bootstraps):
                rints = self._bootstrap_rints[b]  # (the row is filled in place: every pass over N integers counts at N = 4e6)
                for k in range(K):
                    k_indices = groups[k]
                    draw = self.rng.integers(int(self.N_k[k]), size=int(self.N_k[k]))
                    if isinstance(k_indices, range):  # default layout: the samples of state k are one contiguous run
                        np.add(draw, k_indices.start, out=rints[k_indices.start:k_indices.stop])
                    else:
                        rints[k_indices] = k_indices[draw]
                # a replicate is the vector of draw counts: the resident matrix is re-used, nothing is gathered
                self._dm.set_sample_weights(np.bincount(rints, minlength=self.N))
                f_k_init = self.f_k.copy()
                if initialize == "BAR":  # the reference re-runs the BAR chain on every resampled matrix (mbar.py:435-436)
                    # (on a copy: the reference hands over self.f_k itself, which its BAR chain then overwrites in place)
                    f_k_init = self._initialize_with_bar(self.u_kn[:, rints], f_k_init=self.f_k.copy())
                try:
                    self.f_k_boots[b, :] = mbar_solvers.solve_mbar_for_all_states(
                        self._dm, self.N_k, f_k_init, self.states_with_samples, bootstrap_solver_protocol)
                finally:
                    self._dm.set_sample_weights(None)
        elif n_bootstraps < 0:
            logger.warning("n_bootstraps must be an integer >= 0")

        self._Log_W_nk = None
        if self.verbose:
            logger.info("Final dimensionless free energies")
            logger.info("f_k = ")
            logger.info(self.f_k)
            logger.info("MBAR initialization complete.")

    # ---- bootstrap replicates -----------------------------------------------------------------------
    @property
    def bootstrap_rints(self):
        """(n_bootstraps, N) resampled sample indices (mbar.py:420): stored rows with the reference's generator; with
        ``bootstrap_rng="device"`` regenerated from the counter-based stream on first access (n_bootstraps x N integers)."""
        if self._bootstrap_rints is None and self._bootstrap_stream is not None:
            self._bootstrap_rints = np.stack([self._bootstrap_row(b) for b in range(self.n_bootstraps)])
        return self._bootstrap_rints

    @bootstrap_rints.setter
    def bootstrap_rints(self, value):
        self._bootstrap_rints = value

    def _bootstrap_row(self, b):
        """The resampled indices of replicate ``b`` (one row of ``bootstrap_rints``)."""
        if self._bootstrap_stream is None or self._bootstrap_rints is not None:
            return self._bootstrap_rints[b]
        from . import _lib

        seed, cumN, order = self._bootstrap_stream
        return _lib.bootstrap_draws(seed, b, cumN, order)

    def _set_bootstrap_weights(self, dm, b):
        """Make replicate ``b`` the sample multiplicities of ``dm`` (this object's matrix, or an augmented one over the same
        samples): drawn on the device from the counter-based stream, or the draw counts of the stored row."""
        if self._bootstrap_stream is not None and self._bootstrap_rints is None and hasattr(dm, "draw_bootstrap_weights"):
            seed, cumN, order = self._bootstrap_stream
            # (the layout is constant for this object's lifetime: it goes to a matrix's context once, keyed by the stream tuple)
            dm.draw_bootstrap_weights(seed, b, cumN, order, layout_key=self._bootstrap_stream)
        else:
            dm.set_sample_weights(np.bincount(self._bootstrap_row(b), minlength=self.N))

    # ---- weights ----------------------------------------------------------------------------------
    @property
    def Log_W_nk(self):
        """(N, K) log weights (mbar.py:455), computed on the device on first access and then cached
        (the reference materialises them eagerly in the constructor)."""
        if self._Log_W_nk is None:
            self._Log_W_nk = mbar_solvers.mbar_log_W_nk(self._dm, self.N_k, self.f_k)
        return self._Log_W_nk

    @Log_W_nk.setter
    def Log_W_nk(self, value):  # a plain attribute in the reference (mbar.py:455): assignment must work
        self._Log_W_nk = value

    @property
    def W_nk(self):
        """(N, K) weights: ``exp(Log_W_nk)`` taken on the device (not cached: 10 GB at config 3)."""
        if self._Log_W_nk is not None:  # (already on the host, or assigned by the caller: the reference's exp(self.Log_W_nk))
            return np.exp(self._Log_W_nk)
        self._dm.set_Nk(self.N_k)
        return self._dm.w_kn(self.f_k).T

    def weights(self):
        return self.W_nk

    def close(self):
        """Release the device copy of ``u_kn``."""
        if getattr(self, "_dm", None) is not None:
            self._dm.close()
            self._dm = None

    def _gram_w(self):
        self._dm.set_Nk(self.N_k)
        if hasattr(self._dm, "gram_w_cached"):  # (kept while the matrix, N_k, the multiplicities and f_k stay what they are)
            return self._dm.gram_w_cached(self.f_k)
        return self._dm.gram_w(self.f_k)

    def compute_effective_sample_number(self, verbose=False):
        """Kish effective sample number ``1 / sum_n W_nk^2`` per state (mbar.py:495-560): the diagonal of
        ``W^T W``."""
        G, _ = self._gram_w()
        N_eff = 1.0 / np.diag(G)
        if verbose:
            for k in range(self.K):
                logger.info("Effective number of sample in state {:d} is {:10.3f}".format(k, N_eff[k]))
                logger.info("Efficiency for state {:d} is {:6f}/{:d} = {:10.4f}".format(k, N_eff[k], self.N, N_eff[k] / self.N))
        return N_eff

    def compute_overlap(self):
        """Overlap matrix ``O = N_k * (W^T W)``, its eigenvalues and ``1 - second largest`` (mbar.py:563-617)."""
        G, _ = self._gram_w()
        O = self.N_k * G
        with _small_blas(self.K):
            eigenvals = np.sort(np.linalg.eigvals(O))[::-1]
        return dict(scalar=1 - eigenvals[1], eigenvalues=eigenvals, matrix=O)

    # ---- free energy differences --------------------------------------------------------------------
    def compute_free_energy_differences(self, compute_uncertainty=True, uncertainty_method=None, warning_cutoff=1.0e-10,
                                        return_theta=False):
        """``Delta_f[i, j] = f_j - f_i`` and its asymptotic (or bootstrap) uncertainty (mbar.py:620-729)."""
        Deltaf_ij = np.array(self.f_k - np.vstack(self.f_k))
        self._zerosamestates(Deltaf_ij)
        result_vals = dict(Delta_f=Deltaf_ij)
        if uncertainty_method == "bootstrap" and (self.n_bootstraps is None or self.n_bootstraps <= 0):
            raise ParameterError("Cannot request bootstrap sampling of free energy differences without any bootstraps.")
        Theta_ij = None
        if (compute_uncertainty and uncertainty_method != "bootstrap") or return_theta:
            Theta_ij = self._computeAsymptoticCovarianceMatrix(None, self.N_k, method=uncertainty_method)
        if compute_uncertainty:
            if uncertainty_method == "bootstrap":
                diffm = self.f_k_boots[:, np.newaxis, :] - self.f_k_boots[:, :, np.newaxis]
                result_vals["dDelta_f"] = np.std(diffm, axis=0)
            else:
                dDeltaf_ij = np.array(self._ErrorOfDifferences(Theta_ij, warning_cutoff=warning_cutoff))
                self._zerosamestates(dDeltaf_ij)
                result_vals["dDelta_f"] = dDeltaf_ij
        if return_theta:
            result_vals["Theta"] = Theta_ij
        return result_vals

    # ---- private -----------------------------------------------------------------------------------
    def _ErrorOfDifferences(self, cov, warning_cutoff=1.0e-10):
        """``sqrt(cov_ii + cov_jj - 2 cov_ij)``; small negative squares are zeroed (mbar.py:1687-1715)."""
        diag = cov.diagonal()
        d2 = diag + np.vstack(diag) - 2 * cov
        cutoff = -abs(warning_cutoff)
        if np.any(d2 < 0.0):
            if np.any(d2 < cutoff):
                logger.warning("A squared uncertainty is negative. Largest Magnitude = {0:f}".format(abs(np.min(d2[d2 < cutoff]))))
            else:
                d2[np.logical_and(0 > d2, d2 > cutoff)] = 0.0
        return np.sqrt(np.array(d2))

    @staticmethod
    def _pseudoinverse(A, tol=1.0e-10):
        """``np.linalg.pinv(A, rcond=tol)`` (mbar.py:1717-1754 ``_pseudoinverse``).  The one matrix it is applied to here,
        ``I - Sigma V^T N V Sigma`` of the covariance (mbar.py:1851-1858), is symmetric by construction (to 1e-17 as computed), so
        the pseudo-inverse is taken through its eigendecomposition (``hermitian=True``: same cut-off on the same magnitudes, |eigenvalue| =
        singular value) -- Theta equal to the SVD route's to ~3e-15 relative, in 40 % of its time at order 256, where it was a third of a call of
        the expectation family.  ``PYMBAR_AMD_PINV=svd`` restores the literal call."""
        if os.environ.get("PYMBAR_AMD_PINV", "") == "svd" or not np.allclose(A, A.T, rtol=0.0, atol=1e-12 * max(1.0, float(np.abs(A).max()))):
            return np.linalg.pinv(A, rcond=tol)
        return np.linalg.pinv(A, rcond=tol, hermitian=True)

    def _zerosamestates(self, A):
        for pair in self.samestates:
            A[pair[0], pair[1]] = 0
            A[pair[1], pair[0]] = 0

    def _computeAsymptoticCovarianceMatrix(self, W, N_k, method=None):
        """Asymptotic covariance ``Theta`` (mbar.py:1756-1864).  ``W`` may be ``None``: ``W^T W`` and the
        column sums then come straight from the device (methods "svd-ew" and "approximate"); "svd"
        needs the explicit (N, K) matrix."""
        if method is None or method == "bootstrap":
            method = "svd-ew"
        if method not in ("approximate", "svd", "svd-ew"):
            raise ParameterError(f"Method {method} unrecognized.")
        N_k = np.asarray(N_k)
        if W is None and method != "svd":
            G, wsum = self._gram_w()
            return self._theta_from_gram(G, N_k, method, wsum=wsum)
        if W is None:
            W = self.W_nk
        N, Kw = W.shape
        if Kw != N_k.size:
            raise ParameterError("W must be NxK, where N_k is a K-dimensional array.")
        if np.sum(N_k) != N:
            raise ParameterError("W must be NxK, where N = sum_k N_k.")
        from .utils import check_w_normalized

        check_w_normalized(W, N_k)
        return self._theta_from_gram(W.T @ W, N_k, method, W=W)

    def _theta_from_gram(self, G, N_k, method=None, wsum=None, W=None, dm=None, f_full=None):
        """``Theta`` from the Gram matrix ``G = W^T W`` (device MFMA sweep).  "svd-ew" (default):
        eigendecomposition of ``G``, negative eigenvalues clamped, ``Theta = V S pinv(I - S V^T diag(N_k) V S,
        rcond=1e-10) S V^T`` (mbar.py:1838-1858); "approximate": ``G`` itself (:1816); "svd" needs ``W`` (taken
        from ``dm`` if not given) (:1818-1836)."""
        if method is None or method == "bootstrap":
            method = "svd-ew"
        if method not in ("approximate", "svd", "svd-ew"):
            raise ParameterError(f"Method {method} unrecognized.")
        N_k = np.asarray(N_k)
        K = N_k.size
        if wsum is not None:
            check_w_sums(wsum, 0.0)
        if method == "approximate":
            return G
        Ndiag = np.diag(N_k)
        ident = np.identity(K, dtype=np.float64)
        with _small_blas(K):
            return self._theta_dense(G, N_k, method, W, dm, f_full, Ndiag, ident)

    def _theta_dense(self, G, N_k, method, W, dm, f_full, Ndiag, ident):
        if method == "svd":
            if W is None:
                if dm is None:
                    W = self.W_nk
                else:
                    W = np.exp(dm.logw_kn(f_full)).T
            _, S, Vt = np.linalg.svd(W, full_matrices=False)
            Sigma, V = np.diag(S), Vt.T
        else:
            S2, V = np.linalg.eigh(G)
            S2[np.where(S2 < 0.0)] = 0.0
            Sigma = np.diag(np.sqrt(S2))
        # V Sigma pinv(I - Sigma V^T N V Sigma) Sigma V^T (mbar.py:1851-1858) with the diagonal factors applied as row / column
        # scalings: the same numbers as the products with the dense diagonal matrices (those only add exact zeros), two K^3
        # products instead of six -- at K + S = 256 they were a third of compute_entropy_and_enthalpy's host time
        sg, nk = Sigma.diagonal(), Ndiag.diagonal()
        VS = V * sg[None, :]
        inner = ((sg[:, None] * V.T) * nk[None, :]) @ V * sg[None, :]
        return ((VS @ self._pseudoinverse(ident - inner)) * sg[None, :]) @ V.T

    # ---- Log_W_nk consumers (pymbar_amd/expectations.py) -------------------------------------------
    compute_expectations_inner = _expectations.compute_expectations_inner
    compute_covariance_of_sums = _expectations.compute_covariance_of_sums
    compute_expectations = _expectations.compute_expectations
    compute_multiple_expectations = _expectations.compute_multiple_expectations
    compute_perturbed_free_energies = _expectations.compute_perturbed_free_energies
    compute_entropy_and_enthalpy = _expectations.compute_entropy_and_enthalpy

    def _computeUnnormalizedLogWeights(self, u_n):
        """``-ln sum_k N_k exp(f_k - (u_k(x_n) - u(x_n)))`` for one target potential ``u_n`` (mbar.py:1919-1934):
        ``-(u_n + logden_n)`` with the per-sample log-denominator from one device sweep."""
        self._dm.set_Nk(self.N_k)
        return -(np.asarray(u_n, dtype=np.float64) + self._dm.logden(self.f_k))

    def _initialize_with_bar(self, u_kn, f_k_init=None):
        """Chained pairwise BAR guess (mbar.py:1936-1988); see :mod:`pymbar_amd.bar_init`."""
        from .bar_init import initialize_with_bar

        return initialize_with_bar(np.asarray(u_kn), self.N_k, self.x_kindices, f_k_init)

    def _initializeFreeEnergies(self, verbose=False, method="zeros"):
        """Initial guess (mbar.py:1868-1917): zeros or the per-state mean reduced potential."""
        if method == "zeros":
            self.f_k[:] = 0.0
        elif method == "mean-reduced-potential":
            means = np.zeros([self.K], float)
            for k in self.states_with_samples:
                means[k] = self.u_kn[k, 0 : self.N_k[k]].mean()
            if np.max(np.abs(means)) < 0.000001:
                logger.warning("Warning: All mean reduced potentials are close to zero.")
            self.f_k = means
        elif method == "BAR":  # chained pairwise BAR estimates on the host copy (mbar.py:1936-1988)
            from .bar_init import initialize_with_bar

            self.f_k = initialize_with_bar(self.u_kn, self.N_k, self.x_kindices)
        else:
            raise ParameterError("Method " + method + " unrecognized.")
        self.f_k[:] = self.f_k[:] - self.f_k[0]
