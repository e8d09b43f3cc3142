"""Histogram free energy surfaces on the MI355X path: the weight extraction of ``pymbar.FES``.

The reference's ``FES`` pulls two things out of its ``MBAR`` object (pymbar/fes.py:403-416, 1383-1406):

* the unnormalised log weights of the samples in the target potential ``u_n``,
  ``log_w_n = mbar._computeUnnormalizedLogWeights(u_n)`` (:410), from which the bin free energies are
  ``f_i = -logsumexp(log_w_n[samples of bin i])`` (:585);
* for analytical uncertainties, an ``N x (K + nbins)`` weight matrix -- ``exp(Log_W_nk)`` plus one column per populated
  bin, ``W[n, K+i] = exp(log_w_n + f_i)`` on the bin's samples and 0 elsewhere (:1388-1402) -- handed to
  ``_computeAsymptoticCovarianceMatrix`` (:1406).

Here neither matrix exists on the host.  A bin is an *unsampled state* of an augmented reduced-potential matrix whose
row is ``u_n`` on the bin's own samples and ``+inf`` (weight zero) everywhere else; the rows are built on the device from
one vector and one label array (``mbar_ctx_fill_masked_rows``), the bin free energies are the all-state log-space
reduction the solver already has (``mbar_lognum``: ``f_i = -log sum_{n in bin i} exp(-u_n - logden_n)``), and the
covariance input ``W^T W`` of the augmented weights is one MFMA Gram sweep (``mbar_gram_w``).

Only the histogram estimator's weight extraction is mirrored (binning conventions, reference points and the
uncertainty formula of ``FES._get_fes_histogram``); kernel density / spline surfaces and the Monte Carlo sampler of the
reference consume the same ``log_w_n`` vector and are not part of the K x N path.
"""
import logging

import numpy as np

from .utils import DataError, ParameterError

logger = logging.getLogger(__name__)


def unnormalized_log_weights(mbar, u_n):
    """``log_w_n`` of pymbar/fes.py:410 (``-u_n - logden_n``, one device sweep)."""
    return mbar._computeUnnormalizedLogWeights(np.asarray(u_n, dtype=np.float64))


def label_samples(x_n, bin_edges):
    """Bin labels the way pymbar/fes.py:519-563 assigns them on a regular grid: ``sample_label[n]`` numbers the POPULATED
    bins in order of first appearance (the reference's ``bin_order``).  Like the reference, the two overflow regions are
    bins of their own when populated: samples left of the first edge (grid index -1 in some dimension; the reference
    labels them -1 and still gives that label a free energy and a covariance column, fes.py:537-547) and samples right
    of the last edge (``np.digitize`` index ``len(edges) - 1``).  Returns ``(sample_label, grid_of_label)`` with
    ``grid_of_label[i]`` the tuple of per-dimension grid indices of label ``i`` (``None`` for the left-overflow bin)."""
    x_n = np.asarray(x_n, dtype=np.float64)
    if x_n.ndim == 1:
        x_n = x_n[:, None]
    if np.ndim(bin_edges[0]) == 0:
        bin_edges = [bin_edges]
    dims = len(bin_edges)
    if x_n.shape[1] != dims:
        raise DataError("x_n and bin_edges have inconsistent dimension")
    N = len(x_n)
    if N == 0:
        return np.zeros(0, dtype=np.int64), []
    bin_n = [np.digitize(x_n[:, d], bin_edges[d]) - 1 for d in range(dims)]  # each in -1 .. len(edges) - 1
    sizes = tuple(len(bin_edges[d]) + 1 for d in range(dims))
    # one integer per grid cell (0 = "off the grid to the left in some dimension": all such samples share the reference's
    # label -1), then the populated cells numbered in order of first appearance -- without a Python loop over the samples
    left = np.zeros(N, dtype=bool)
    for b in bin_n:
        left |= b < 0
    P = float(np.prod(sizes, dtype=np.float64)) + 1.0
    if P >= 2.0 ** 62:
        # more cells than an int64 can number (many dimensions): sort the samples' index TUPLES instead of flat cell numbers
        stacked = np.stack([np.where(left, -1, b) for b in bin_n], axis=1)  # (all left-overflow samples share one tuple)
        tuples, first, inv = np.unique(stacked, axis=0, return_index=True, return_inverse=True)
        by_first = np.argsort(first, kind="stable")
        rank = np.empty(len(tuples), dtype=np.int64)
        rank[by_first] = np.arange(len(tuples))
        sample_label = rank[np.asarray(inv).reshape(-1)]
        grid = [None if t[0] < 0 else tuple(int(v) for v in t) for t in tuples[by_first]]
        return sample_label, grid
    flat = np.ravel_multi_index(tuple(b + 1 for b in bin_n), sizes) + 1
    flat[left] = 0
    P = int(P)
    if P <= 50_000_000 and P <= 8 * N + 1024:
        # a grid no larger than a few cells per sample: tabulate it (two int64 tables over the grid, no sort)
        first = np.full(P, N, dtype=np.int64)
        first[flat[::-1]] = np.arange(N - 1, -1, -1)  # (duplicates: the last write wins, so the smallest n is kept)
        cells = np.nonzero(first < N)[0]
        cells = cells[np.argsort(first[cells], kind="stable")]
        rank = np.zeros(P, dtype=np.int64)
        rank[cells] = np.arange(len(cells))
        sample_label = rank[flat]
    else:  # a fine grid and few samples (or a huge grid): sort the samples' cells instead of tabulating the grid
        cells, first, inv = np.unique(flat, return_index=True, return_inverse=True)
        by_first = np.argsort(first, kind="stable")
        rank = np.empty(len(cells), dtype=np.int64)
        rank[by_first] = np.arange(len(cells))
        sample_label = rank[inv]
        cells = cells[by_first]
    grid = [None if c == 0 else tuple(int(v) - 1 for v in np.unravel_index(int(c) - 1, sizes)) for c in cells]
    return sample_label, grid


def histogram_fes(mbar, u_n, sample_label, reference="from-lowest", reference_label=None, uncertainty_method="analytical",
                  theta_method=None):
    """Free energies of the populated histogram bins and their uncertainties, relative to a reference bin.

    ``sample_label[n]`` in ``[0, nbins)`` is the bin of sample ``n`` (-1: not in any bin); every label below ``nbins =
    max + 1`` must occur.  ``reference``: "from-lowest" (the bin of lowest free energy) or "from-specified"
    (``reference_label``) -- pymbar/fes.py:1362-1376.  ``uncertainty_method``: "analytical" (fes.py:1381-1415) or None.

    Returns ``dict(f_i, df_i, f_raw, reference, Theta)``; ``f_raw`` are the bin free energies before the reference
    is subtracted (``histogram_data["f"]`` of the reference)."""
    from .device import DeviceMatrix
    from .expectations import _augmented_solve

    u_n = np.ascontiguousarray(u_n, dtype=np.float64)
    sample_label = np.asarray(sample_label, dtype=np.int64)
    K, N = mbar.K, mbar.N
    if u_n.shape != (N,) or sample_label.shape != (N,):
        raise ParameterError("u_n and sample_label must have one entry per sample")
    nbins = int(sample_label.max()) + 1 if N > 0 else 0
    if nbins < 1:
        raise DataError("no sample falls into any bin")
    counts = np.bincount(sample_label[sample_label >= 0], minlength=nbins)
    if np.any(counts == 0):
        raise DataError(f"WARNING: bin {int(np.where(counts == 0)[0][0])} has no samples -- all bins must have at least one sample.")
    if uncertainty_method not in (None, "analytical"):
        raise ParameterError(f"Uncertainty_method {uncertainty_method} is not a valid option")

    dm = DeviceMatrix.empty(K + nbins, N, device=getattr(mbar, "_device", None))
    try:
        dm.copy_rows_from(mbar._dm, 0, 0, K)
        dm.fill_masked_rows(K, nbins, u_n, sample_label)  # one "state" per bin: u_n on its samples, +inf elsewhere
        N_aug = np.zeros(K + nbins, dtype=np.float64)
        N_aug[:K] = mbar.N_k
        dm.set_Nk(N_aug)
        f_full, _ = _augmented_solve(dm, K, nbins, mbar.f_k)
        f_raw = f_full[K:].copy()  # = -logsumexp(log_w_n[bin])   (fes.py:585)
        if reference == "from-lowest":
            j = int(np.argmin(f_raw))
        elif reference == "from-specified":
            if reference_label is None or not (0 <= int(reference_label) < nbins):
                raise ParameterError("Specified reference point for FES not given")
            j = int(reference_label)
        else:
            raise ParameterError(f"reference point method {reference} is not supported for histogram surfaces here")
        out = dict(f_i=f_raw - f_raw[j], f_raw=f_raw, reference=j)
        if uncertainty_method == "analytical":
            G, wsum = dm.gram_w(f_full)
            Theta = mbar._theta_from_gram(G, N_aug.astype(np.int64), theta_method, wsum=wsum, dm=dm, f_full=f_full)
            d = np.diag(Theta)[K:]
            var = d + d[j] - 2.0 * Theta[K:, K + j]
            out["df_i"] = np.sqrt(np.maximum(var, 0.0))
            out["Theta"] = Theta
    finally:
        dm.close()
    return out
