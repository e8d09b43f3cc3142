"""Helpers shared by the solver mirror and the MBAR class (host side, numpy only).

Behavioural mirror of the pieces of pymbar/utils.py that the solver path imports
(pymbar/mbar_solvers.py:10): ``ensure_type`` (utils.py:117-232), ``check_w_normalized``
(utils.py:340-393), ``kln_to_kn`` (utils.py:41-76) and the exception classes (utils.py:401-422).
"""
import warnings

import numpy as np


try:
    # With pymbar itself installed the SAME exception classes are raised, so callers that switch the backend keep
    # their ``except pymbar.utils.ParameterError`` clauses (and the reference's own tests pass unchanged).
    from pymbar.utils import (BoundsError, ConvergenceError, DataError, ParameterError,  # noqa: F401
                              TypeCastPerformanceWarning)
except Exception:  # pymbar absent (or not importable in this interpreter): same names, same meaning (utils.py:401-422)

    class ParameterError(Exception):
        """An error in the input parameters has been detected."""

    class ConvergenceError(Exception):
        """Convergence could not be achieved."""

    class BoundsError(Exception):
        """Could not determine bounds on free energy."""

    class DataError(Exception):
        """Data is inconsistent."""

    class TypeCastPerformanceWarning(RuntimeWarning):
        pass


def ensure_type(val, dtype, ndim, name, length=None, can_be_none=False, shape=None, warn_on_cast=True,
                add_newaxis_on_deficient_ndim=False):
    """Check dtype / ndim / shape of an array, casting (with a warning) when needed; the result is
    always C-contiguous.  Same contract and error types as pymbar/utils.py:117-232."""
    if can_be_none and val is None:
        return None
    if not isinstance(val, np.ndarray):
        if add_newaxis_on_deficient_ndim and ndim == 1 and np.isscalar(val):
            val = np.array([val])
        else:
            raise TypeError(f"{name} must be numpy array.  You supplied type {type(val)}")
    if warn_on_cast and val.dtype != dtype:
        warnings.warn(f"Casting {name} dtype={val.dtype} to {dtype} ", TypeCastPerformanceWarning)
    if val.ndim != ndim:
        if add_newaxis_on_deficient_ndim and val.ndim + 1 == ndim:
            val = val[np.newaxis, ...]
        else:
            raise ValueError(f"{name} must be ndim {ndim}. You supplied {val.ndim}")
    val = np.ascontiguousarray(val, dtype=dtype)
    if length is not None and len(val) != length:
        raise ValueError(f"{name} must be length {length}. You supplied {len(val)}.")
    if shape is not None:
        want = str(shape).replace("None", "Any")
        if len(shape) != val.ndim or any(b is not None and a != b for a, b in zip(val.shape, shape)):
            raise ValueError(f"{name} must be shape {want}. You supplied  {val.shape}")
    return val


def check_w_normalized(W, N_k, tolerance=1.0e-4):
    """``sum_n W_nk = 1`` and ``sum_k N_k W_nk = 1`` within ``tolerance`` else ParameterError
    (pymbar/utils.py:340-393)."""
    N, K = W.shape
    column_sums = np.sum(W, axis=0)
    bad = np.abs(column_sums - 1) > tolerance
    if np.any(bad):
        first = int(np.arange(K)[bad][0])
        raise ParameterError(
            f"Warning: Should have \\sum_n W_nk = 1. Actual column sum for state {first:d} was "
            f"{column_sums[first]:f}. {int(np.sum(bad)):d} other columns have similar problems. \n"
            "This generally indicates the free energies are not converged.")
    row_sums = np.sum(W * N_k, axis=1)
    bad = np.abs(row_sums - 1) > tolerance
    if np.any(bad):
        first = int(np.arange(N)[bad][0])
        raise ParameterError(
            f"Warning: Should have \\sum__k N_k W_nk = 1. Actual row sum for state {first:d} was "
            f"{row_sums[first]:f}. {int(np.sum(bad)):d} other columns have similar problems. \n"
            "This generally indicates the free energies are not converged.")


def check_w_sums(column_sums, row_sum_dev, tolerance=1.0e-4):
    """Device-side variant of :func:`check_w_normalized`: takes the per-state sums ``sum_n W_nk``
    and the largest per-sample deviation ``max_n |sum_k N_k W_nk - 1|`` computed on the GPU."""
    column_sums = np.asarray(column_sums)
    bad = np.abs(column_sums - 1) > tolerance
    if np.any(bad):
        first = int(np.where(bad)[0][0])
        raise ParameterError(
            f"Warning: Should have \\sum_n W_nk = 1. Actual column sum for state {first:d} was "
            f"{column_sums[first]:f}. {int(np.sum(bad)):d} other columns have similar problems. \n"
            "This generally indicates the free energies are not converged.")
    if row_sum_dev > tolerance:
        raise ParameterError("Warning: Should have \\sum__k N_k W_nk = 1.")


def kn_to_n(kn, N_k=None, cleanup=False):
    """(K, N_max) -> (N_total,) concatenation of the first N_k[k] entries of each row (pymbar/utils.py:78-114)."""
    kn = np.asarray(kn)
    K, N_max = kn.shape
    if N_k is None:
        N_k = N_max * np.ones([K], dtype=np.int64)
    out = np.concatenate([kn[k, : int(N_k[k])] for k in range(K)]).astype(np.float64)
    if cleanup:
        del kn
    return out


def kln_to_kn(kln, N_k=None, cleanup=False):
    """(K, L, N_max) -> (L, N_total) concatenation of the first N_k[k] samples of each k
    (pymbar/utils.py:41-76)."""
    K, L, N_max = np.shape(kln)
    if N_k is None:
        N_k = N_max * np.ones([K], dtype=np.int64)
    N = int(np.sum(N_k))
    kn = np.zeros([L, N], dtype=np.float64)
    i = 0
    for k in range(K):
        if N_k[k] > 0:
            kn[:, i : i + N_k[k]] = kln[k, :, 0 : N_k[k]]
            i += N_k[k]
    if cleanup:
        del kln
    return kn


def state_index_groups(x_kindices, K):
    """``[np.where(x_kindices == k)[0] for k in range(K)]`` (pymbar/mbar.py:424, :1958-1967) in one pass: ``range`` objects
    when the samples are already ordered by state (the default layout: no index arrays at all), otherwise slices of a stable
    argsort.  Like the reference's ``==`` masks this accepts float-typed indices (integral values match their state,
    anything else matches none) and values outside ``[0, K)`` (they belong to no state)."""
    x = np.asarray(x_kindices)
    if x.size == 0:
        return [range(0, 0) for _ in range(K)]
    if x.dtype.kind in "iu":
        xi = x.astype(np.int64, copy=False)
        valid = (xi >= 0) & (xi < K)
    else:
        xf = np.asarray(x, dtype=np.float64)
        with np.errstate(invalid="ignore"):
            valid = (xf >= 0) & (xf < K) & (xf == np.floor(xf))
        xi = np.where(valid, xf, 0).astype(np.int64)
    if valid.all():
        counts = np.bincount(xi, minlength=K)[:K]
        offs = np.concatenate(([0], np.cumsum(counts)))
        if np.all(xi[:-1] <= xi[1:]):
            return [range(int(offs[k]), int(offs[k + 1])) for k in range(K)]
        order = np.argsort(xi, kind="stable")
        return [order[offs[k]:offs[k + 1]] for k in range(K)]
    idx = np.nonzero(valid)[0]
    counts = np.bincount(xi[idx], minlength=K)[:K]
    offs = np.concatenate(([0], np.cumsum(counts)))
    order = idx[np.argsort(xi[idx], kind="stable")]
    return [order[offs[k]:offs[k + 1]] for k in range(K)]


def _row_chunks(nrows, nthreads):
    import numpy as _np

    b = _np.linspace(0, nrows, nthreads + 1).astype(int)
    return [(int(b[i]), int(b[i + 1])) for i in range(nthreads) if b[i + 1] > b[i]]


def _host_threads(nrows):
    import os as _os

    return max(1, min(16, int(nrows), (_os.cpu_count() or 1)))


# (glibc serves blocks up to 32 MB from its heap -- pages that have been touched before -- and maps larger ones afresh: above that
# size a copy is bound by the first touch of its pages on ONE thread (15 GB/s on the GPU box's host against 75 GB/s below it),
# which several threads take in parallel; below it numpy's own copy is the fastest: 30 MB in 0.42 ms against 0.58 ms threaded)
_THREADED_COPY_BYTES = 32 << 20


def private_copy(src):
    """``np.array(src)`` of a float64 C-contiguous matrix; from 32 MB on the rows are copied by several threads (numpy releases
    the GIL in ``copyto``): a fresh multi-GB array is bound by the first touch of its pages on one thread (14 GB/s on the GPU
    box's host), which several threads take in parallel (120-145 GB/s)."""
    import numpy as _np

    if src.nbytes < _THREADED_COPY_BYTES or src.ndim != 2 or src.shape[0] < 2:
        return _np.array(src, dtype=_np.float64)
    from concurrent.futures import ThreadPoolExecutor

    out = _np.empty_like(src)
    # (at least ~4 MB per thread)
    chunks = _row_chunks(src.shape[0], min(_host_threads(src.shape[0]), max(2, src.nbytes >> 22)))
    with ThreadPoolExecutor(len(chunks)) as ex:
        list(ex.map(lambda c: _np.copyto(out[c[0]:c[1]], src[c[0]:c[1]]), chunks))
    return out


def prefault(out):
    """Touch the pages of a freshly allocated 2-D array from several threads (from 64 MB on): a device-to-host copy into
    untouched pageable memory runs at the page-fault rate of the ONE driver thread that performs it (17 GB/s measured for the
    10 GB ``Log_W_nk`` of config 3), into touched memory at the PCIe rate."""
    if out.nbytes < (64 << 20) or out.ndim != 2 or out.shape[0] < 2:
        return out
    from concurrent.futures import ThreadPoolExecutor

    chunks = _row_chunks(out.shape[0], _host_threads(out.shape[0]))
    with ThreadPoolExecutor(len(chunks)) as ex:
        list(ex.map(lambda c: out[c[0]:c[1]].fill(0.0), chunks))
    return out
