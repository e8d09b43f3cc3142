"""Expectations, perturbed free energies and entropy/enthalpy on the MI355X path.

Host-side mirror of the ``Log_W_nk`` consumers of ``pymbar.MBAR``:
``compute_expectations_inner`` (pymbar/mbar.py:732-1001), ``compute_expectations`` (:1039-1312),
``compute_multiple_expectations`` (:1315-1439), ``compute_perturbed_free_energies`` (:1442-1521),
``compute_entropy_and_enthalpy`` (:1524-1681), ``compute_covariance_of_sums`` (:1005-1036).

The reference materialises an ``N x (K + NL + S)`` log-weight matrix on the host and runs ``logsumexp`` over
the sample axis for every extra column.  Here every extra column is an *unsampled state* of an augmented
reduced-potential matrix:

* a new state ``l`` with potentials ``u_ln``                         ->  row ``u_ln``,
* an observable ``A_i`` (shifted to be positive) at state ``l``      ->  row ``u_ln - log A_i``,

so that, with ``N_k = 0`` for the extra rows,

* their normalisers are the all-state log-space reduction the solver already uses for unsampled states
  (``mbar_lognum``:  ``f_row = -log sum_n exp(-row_n - logden_n)``),
* ``<A_i>_l = exp(lognum(A row) - lognum(l row))``, and
* the covariance input ``W^T W`` of the augmented weight matrix is one MFMA Gram sweep (``mbar_gram_w``).

No N x K array is formed on the host; only (K + NL + S)-sized linear algebra runs there.
"""
import logging

import numpy as np

from .utils import DataError, ParameterError, kln_to_kn, kn_to_n

logger = logging.getLogger(__name__)



def _augmented_matrix(mbar, u_ln, L_list, state_rows, A_n, obs_rows, device, dedup=False, extend=False):
    """Assemble ``[u_kn; u_ln[L_list]; u_ln[state_rows[s]] - log(A[obs_rows[s]] - shift)]`` ON THE DEVICE, the extra rows as
    unsampled states (N_k = 0: they do not enter the denominator).  The resident ``u_kn`` is copied device to device; a
    new-state row that IS a resident row (``u_ln is mbar.u_kn``, the default of compute_expectations) likewise, and so are
    observables that are resident rows (entropy / enthalpy: the reduced potentials themselves); only genuinely new rows and
    the RAW observables cross PCIe -- never the N x (K + NL + S) host array of mbar.py:886-903.  The shift that makes an
    observable positive (mbar.py:858-867: ``A - (min A - |4 eps min A|)``) and the logarithm are taken on the device
    (``rows_logshift`` / ``vec_logshift``); returns ``(dm, N_aug, shift, col)`` with ``shift[i]`` of observable ``i`` and
    ``col[l]`` the row that holds state ``l``.

    ``dedup`` (only when the new states ARE resident rows): the copies are not made at all -- state ``l`` is row ``l`` -- and the
    matrix has K + S rows instead of K + NL + S (compute_expectations and the entropy / enthalpy decomposition at 128 states:
    256 rows instead of 384, which keeps the sweeps on the one-read kernels); the caller expands the Gram matrix.

    ``extend`` (round 6): the resident rows are not copied either -- the extra rows go into an extension of the resident matrix
    (``DeviceMatrix.extend``: the library sweeps ``[resident rows | appended rows]`` as one panel) when it can hold them (up to
    128 resident states, 129 .. 256 rows in total, one rank); observables that are resident rows at resident states (entropy /
    enthalpy) are then written in one pass, ``u_l - log(u_i - shift)`` straight from the resident rows."""
    from .device import DeviceMatrix

    K, N = mbar.K, mbar.N
    NL, S = len(L_list), len(state_rows)
    resident = u_ln is mbar.u_kn
    dedup = dedup and resident
    if dedup:
        NL = 0
    dm = None
    if extend and hasattr(mbar._dm, "extend") and NL + S > 0:
        dm = mbar._dm.extend(NL + S)
    if dm is None:
        dm = DeviceMatrix.empty(K + NL + S, N, device=device)
        dm.copy_rows_from(mbar._dm, 0, 0, K)
    shift = np.zeros(len(A_n) if S > 0 else 0, dtype=np.float64)

    def runs(pairs):
        """(dst, src) row pairs -> maximal (dst0, src0, n) runs with both indices consecutive: one device call per run."""
        out = []
        for d, s_ in pairs:
            if out and out[-1][0] + out[-1][2] == d and out[-1][1] + out[-1][2] == s_:
                out[-1][2] += 1
            else:
                out.append([d, s_, 1])
        return out

    if dedup:
        col = {int(l): int(l) for l in L_list}
    else:
        if resident:
            for d0, s0, n in runs([(K + j, int(l)) for j, l in enumerate(L_list)]):
                dm.copy_rows_from(mbar._dm, d0, s0, n)
        else:
            for j, l in enumerate(L_list):
                dm.upload_rows(K + j, u_ln[int(l)][np.newaxis, :])
        col = {int(l): K + j for j, l in enumerate(L_list)}
    uniq = np.unique(obs_rows) if S > 0 else []
    if len(uniq) > 2 and len(uniq) * 2 > S:
        # mostly DIFFERENT observables (entropy / enthalpy: the K reduced potentials, each at its own state): the raw rows go
        # into the observable rows (device to device when they are rows of the resident matrix, one transfer per run of
        # consecutive observables otherwise), become log(A - shift) there and then u - log(A - shift), one launch per run
        if A_n is mbar.u_kn and hasattr(dm, "rows_obs_from_base") and all(col[int(l)] < K for l in state_rows):
            # (resident observables at resident states into appended rows: one pass per run, no copy of the observable rows)
            trip = []
            for s in range(S):
                d, o_, st = K + NL + s, int(obs_rows[s]), col[int(state_rows[s])]
                if trip and trip[-1][0] + trip[-1][3] == d and trip[-1][1] + trip[-1][3] == o_ and trip[-1][2] + trip[-1][3] == st:
                    trip[-1][3] += 1
                else:
                    trip.append([d, o_, st, 1])
            for d0, o0, s0, n in trip:
                shift[o0:o0 + n] = dm.rows_obs_from_base(d0, s0, o0, n)
        else:
            for d0, o0, n in runs([(K + NL + s, int(obs_rows[s])) for s in range(S)]):
                if A_n is mbar.u_kn:
                    dm.copy_rows_from(mbar._dm, d0, o0, n)
                else:
                    dm.upload_rows(d0, A_n[o0:o0 + n])
            shift_rows = dm.rows_logshift(K + NL, S)
            shift[np.asarray(obs_rows, dtype=int)] = shift_rows
            for d0, s0, n in runs([(K + NL + s, col[int(state_rows[s])]) for s in range(S)]):
                dm.rows_rsub(d0, s0, n)
        uniq = []
    for i in uniq:  # one upload of A_i; its log, shifted, is subtracted at every state it is evaluated at
        shift[int(i)] = dm.vec_logshift(A_n[int(i)])
        # observable rows = state rows - log(A_i - shift), written in one pass per run of consecutive rows
        for d0, s0, n in runs([(K + NL + s, col[int(state_rows[s])]) for s in range(S) if int(obs_rows[s]) == int(i)]):
            dm.rows_sub(d0, s0, n, None)
    N_aug = np.zeros(K + NL + S, dtype=np.float64)
    N_aug[:K] = mbar.N_k
    dm.set_Nk(N_aug)
    return dm, N_aug, shift, col


def _augmented_solve(dm, K, R, f_k):
    """Normalisers of the augmented matrix at ``f_k``: (f_full, lognum) with f_full[K:] = -lognum[K:]."""
    f_full = np.zeros(K + R, dtype=np.float64)
    f_full[:K] = f_k
    lognum = dm.lognum(f_full)
    f_full[K:] = -lognum[K:]
    return f_full, lognum


def compute_expectations_inner(mbar, A_n, u_ln, state_map, uncertainty_method=None, warning_cutoff=1.0e-10,
                               return_theta=False):
    """Expectations of observables ``A_i`` at the states of ``u_ln`` selected by ``state_map``
    (``[[states...],[observables...]]``, or a 1-D list of states for free energies only), with the covariance
    ``Theta`` of the log normalisers.  Same inputs and result keys as pymbar/mbar.py:732-1001:
    ``observables``, ``f``, ``Theta``, ``Amin`` (+ ``bootstrapped_observables`` / ``bootstrapped_f``)."""
    state_map = np.asarray(state_map)
    if state_map.ndim < 2:
        state_list = np.array(state_map, dtype=int)
        obs_list = np.zeros(0, dtype=int)
    else:
        state_list = np.array(state_map[0, :], dtype=int)
        obs_list = np.array(state_map[1, :], dtype=int)
    S = len(obs_list)
    u_ln = np.asarray(u_ln, dtype=np.float64)
    if u_ln.ndim == 1:
        u_ln = u_ln.reshape(1, -1)
    A_n = np.asarray(A_n, dtype=np.float64)  # (never modified: the shift to positive values and the log are taken on the device)
    if A_n.ndim == 1:
        A_n = A_n.reshape(1, -1)
    K, N = mbar.K, mbar.N
    L_list = np.unique(state_list)
    NL = len(L_list)
    col_of_state = {int(l): j for j, l in enumerate(L_list)}  # column K + j of the reference's augmented matrix (:886-903)
    # New states that ARE resident rows (the default of compute_expectations, the entropy / enthalpy decomposition) are not
    # duplicated on the device: W_l = W_k exp(f_l - f_k) column for column, so the Gram matrix of the reference's
    # (K + NL + S)-column layout is the (K + S)-row one with rows / columns repeated and scaled.  ("svd" needs W itself.)
    dedup = u_ln is mbar.u_kn and uncertainty_method != "svd"

    result_vals = dict()
    bootstrap = uncertainty_method == "bootstrap"
    n_total = mbar.n_bootstraps + 1 if bootstrap else 1
    if bootstrap:
        A_i_bootstrap = np.zeros([mbar.n_bootstraps, S])
        f_bootstrap = np.zeros([mbar.n_bootstraps, len(state_list)])
    Theta_ij = None
    # ONE observable at resident states, no bootstrap replicates: no augmented matrix at all (see _single_observable_moments)
    if (dedup and not bootstrap and S > 0 and np.all(obs_list == obs_list[0]) and len(state_list) == S
            and len(np.unique(state_list)) == S and getattr(mbar, "_dm", None) is not None
            and hasattr(mbar._dm, "weights_from_vec") and getattr(mbar._dm, "nranks", 1) == 1):
        try:
            return _single_observable_moments(mbar, A_n[int(obs_list[0])], state_list, int(obs_list[0]), len(A_n), col_of_state,
                                              L_list, uncertainty_method, return_theta)
        except _LinearWeightsOverflow:
            pass  # an observable spanning more than ~1e154: its square has no linear-space weights; the log-space path below has no such limit
    # The augmented matrix goes to the device once.  A bootstrap replicate (mbar.py:905-912 gathers
    # u_kn[:, bootstrap_rints[n]]) is the same matrix with per-sample multiplicities = draw counts.
    # (observables are made strictly positive so that they can live in log space, mbar.py:858-867: shift[i] is the reference's
    # A_min[i] - logfactors[i], found on the device)
    # (an extension of the resident matrix instead of an augmented copy where the library can sweep the two as one panel; bootstrap
    # replicates put multiplicities on the augmented matrix and "svd" downloads its weights: those keep the copy)
    dm, N_dm, shift, row_of_state = _augmented_matrix(mbar, u_ln, L_list, state_list[:S] if S > 0 else [], A_n, obs_list,
                                                      getattr(mbar, "_device", None), dedup=dedup,
                                                      extend=not bootstrap and uncertainty_method != "svd")
    R_dm = len(N_dm) - K      # extra rows on the device: S when the state rows are not duplicated, NL + S otherwise
    obs_row0 = len(N_dm) - S  # first observable row
    try:
        for n in range(n_total):
            if n == 0:
                f_k = mbar.f_k
            else:
                f_k = mbar.f_k_boots[n - 1, :]
                if hasattr(mbar, "_set_bootstrap_weights"):  # (this repository's class: drawn on the device when its stream lives there)
                    mbar._set_bootstrap_weights(dm, n - 1)
                else:
                    dm.set_sample_weights(np.bincount(mbar.bootstrap_rints[n - 1], minlength=N))
            f_full, lognum = _augmented_solve(dm, K, R_dm, f_k)
            f_states = np.array([-lognum[row_of_state[int(l)]] for l in state_list])
            A_i = np.array([np.exp(lognum[obs_row0 + s] - lognum[row_of_state[int(state_list[s])]]) for s in range(S)])
            if n == 0:
                if S > 0:
                    result_vals["observables"] = A_i + shift[obs_list]
                if return_theta:
                    # (the column sums go along: like the reference's check_w_normalized on the augmented W, weights that
                    # do not sum to one raise instead of silently producing a Theta)
                    G, wsum = dm.gram_w(f_full)
                    if dedup:
                        # the reference's layout [K sampled | NL state copies | S observables] from the rows on the device:
                        # copy l is row l scaled by c_l = exp(f_l - f_k[l]), f_l = -lognum[l] its normaliser as an unsampled state
                        # (N = 0).  Theta = W^T (I - W N W^T)^+ W (Eq. D6) is bilinear in the weight columns, so Theta of the
                        # (K + S) DISTINCT columns is computed (an eigendecomposition of that order) and the copies' rows /
                        # columns are those of column l times c_l.
                        src = np.concatenate((np.arange(K), L_list.astype(int), K + np.arange(S))).astype(int)
                        scale = np.concatenate((np.ones(K), np.exp(-lognum[L_list.astype(int)] - f_k[L_list.astype(int)]), np.ones(S)))
                        Theta_red = mbar._theta_from_gram(G, N_dm.astype(np.int64), uncertainty_method, wsum=wsum)
                        Theta_ij = (scale[:, None] * Theta_red[np.ix_(src, src)]) * scale[None, :]
                    else:
                        Theta_ij = mbar._theta_from_gram(G, N_dm.astype(np.int64), uncertainty_method, wsum=wsum, dm=dm,
                                                         f_full=f_full)
                result_vals["f"] = f_states
            else:
                A_i_bootstrap[n - 1, :] = A_i + shift[obs_list] if S > 0 else 0.0
                f_bootstrap[n - 1, :] = f_states
    finally:
        dm.close()
    if bootstrap:
        result_vals["bootstrapped_observables"] = A_i_bootstrap
        result_vals["bootstrapped_f"] = f_bootstrap
    if return_theta:
        si = K + NL + np.arange(S)
        li = K + np.array([col_of_state[int(l)] for l in state_list], dtype=int)
        idx = np.concatenate((si, li)).astype(int)
        result_vals["Theta"] = Theta_ij[np.ix_(idx, idx)]
        if S > 0:
            result_vals["Amin"] = shift[obs_list]
    return result_vals


class _LinearWeightsOverflow(Exception):
    """``(A - shift)**p`` overflowed on the device (mbar_ctx_weights_from_vec, MBAR_ERR_NUMERIC)."""


def _weights_from_vec(dm, power):
    from ._lib import MbarHipError

    try:
        dm.weights_from_vec(power)
    except MbarHipError as exc:
        if exc.code == -6:  # MBAR_ERR_NUMERIC
            raise _LinearWeightsOverflow() from exc
        raise


def _single_observable_moments(mbar, A_row, state_list, obs_index, n_obs, col_of_state, L_list, uncertainty_method, return_theta):
    """``compute_expectations_inner`` for ONE observable evaluated at S distinct RESIDENT states (``compute_expectations(A_n)``,
    mbar.py:1039-1312, the common call) WITHOUT an augmented matrix.  The weight column of "A at state l" is
    ``W_s = A'_n W_nl exp(f_s - f_l)`` with ``A' = A - shift > 0`` (mbar.py:858-867), so

    * its normaliser is ``f_s = -log sum_n A'_n exp(-u_ln - logden_n)``: the log-space reduction of the RESIDENT matrix with the
      per-sample weights ``A'`` (``mbar_lognum`` after ``mbar_ctx_weights_from_vec(1)``), one read;
    * the covariance input ``W^T W`` of the reference's N x (K + NL + S) matrix (mbar.py:886-903) is made of the three weighted
      Gram matrices ``G_p = sum_n A'^p W W^T`` (p = 0, 1, 2) of the resident matrix: ``[G_0 | G_1 d ; d G_1 | d G_2 d]`` with
      ``d_s = exp(f_s - f_l(s))`` -- three sweeps of K rows on the matrix cores instead of a device-to-device copy of the
      resident rows, a rewrite of S observable rows and sweeps over K + S rows (at 128 states: 256-row kernels, twice the
      blocks of three 128-row sweeps; config 3: 20 GB less device memory and half the sweep time).

    Same result keys as the general path."""
    K, N = mbar.K, mbar.N
    dm = mbar._dm
    S = len(state_list)
    NL = len(L_list)
    states = np.asarray(state_list, dtype=int)
    f_k = mbar.f_k
    result_vals = dict()
    shift = np.zeros(n_obs, dtype=np.float64)
    try:
        dm.set_sample_weights(None)
        cached = hasattr(dm, "lognum_cached")
        lognum0 = dm.lognum_cached(f_k) if cached else dm.lognum(f_k)   # resident rows as states: -f_l (kept across calls at the same f_k)
        shift[obs_index] = dm.vec_logshift(A_row)       # log(A - shift) stays on the device
        _weights_from_vec(dm, 1.0)
        lognum1 = dm.lognum(f_k)                       # log sum_n A'_n exp(-u_ln - logden_n)
        f_states = -lognum0[states]
        A_i = np.exp(lognum1[states] - lognum0[states])
        result_vals["observables"] = A_i + shift[obs_index]
        result_vals["f"] = f_states
        if return_theta:
            G1, ws1 = dm.gram_w(f_k)
            _weights_from_vec(dm, 2.0)
            G2, _ = dm.gram_w(f_k)
            dm.set_sample_weights(None)
            G0, ws0 = dm.gram_w_cached(f_k) if cached else dm.gram_w(f_k)
            # Theta on the (K + S) DISTINCT weight columns [K resident | S observables], observable s = A' W_l(s) scaled by
            # d_s = exp(f_s - f_k[l(s)]), f_s = -lognum1[l(s)].  The reference's layout [K sampled | NL state copies | S observables]
            # (mbar.py:886-903) holds, besides these, NL columns that are exact multiples c_l W_l of resident ones, c_l =
            # exp(f_l - f_k[l]), with N = 0: Theta = W^T (I - W N W^T)^+ W (Eq. D6; what the eigen-form of mbar.py:1838-1858
            # evaluates) is bilinear in the columns, so their rows / columns of Theta are those of column l times c_l -- an
            # eigendecomposition of order K + S instead of K + NL + S (256 instead of 384 at 128 states: 7 ms instead of 23).
            Ls = L_list.astype(int)
            c_l = np.exp(-lognum0[Ls] - f_k[Ls])
            d_s = np.exp(-lognum1[states] - f_k[states])
            n_red = K + S
            G = np.empty((n_red, n_red), dtype=np.float64)
            G[:K, :K] = G0
            G[:K, K:] = G1[:, states] * d_s[None, :]
            G[K:, K:] = (d_s[:, None] * G2[np.ix_(states, states)]) * d_s[None, :]
            iu = np.triu_indices(n_red, 1)
            G[(iu[1], iu[0])] = G[iu]                     # (symmetric: the lower triangle from the upper)
            wsum = np.concatenate((ws0, d_s * ws1[states]))
            N_red = np.zeros(n_red, dtype=np.int64)
            N_red[:K] = mbar.N_k
            Theta_red = mbar._theta_from_gram(G, N_red, uncertainty_method, wsum=wsum)
            src = np.concatenate((np.arange(K), Ls, K + np.arange(S))).astype(int)
            scale = np.concatenate((np.ones(K), c_l, np.ones(S)))
            Theta_ij = (scale[:, None] * Theta_red[np.ix_(src, src)]) * scale[None, :]
            si = K + NL + np.arange(S)
            li = K + np.array([col_of_state[int(l)] for l in state_list], dtype=int)
            idx = np.concatenate((si, li)).astype(int)
            result_vals["Theta"] = Theta_ij[np.ix_(idx, idx)]
            result_vals["Amin"] = shift[np.full(S, obs_index, dtype=int)]
    finally:
        dm.set_sample_weights(None)
    return result_vals


def compute_covariance_of_sums(mbar, d_ij, K, a):
    """Covariance of ``sum_k a_k (x_ik - x_jk)`` from the matrix of standard deviations ``d_ij`` of the
    (nK x nK) differences (pymbar/mbar.py:1005-1036), vectorised."""
    var = np.square(np.asarray(d_ij, dtype=np.float64))
    a = np.asarray(a, dtype=np.float64)
    n = len(a)
    d2 = np.zeros([K, K], float)
    ii = np.arange(K)
    for k in range(n):
        bk = var[k * K : (k + 1) * K]
        d2 += a[k] ** 2 * bk[:, k * K : (k + 1) * K]
        for l in range(n):
            blk = bk[:, l * K : (l + 1) * K]                       # var[i + kK, j + lK]
            blk_t = var[k * K : (k + 1) * K, l * K : (l + 1) * K].T   # var[j + kK, i + lK]
            diag = blk[ii, ii]                                        # var[i + kK, i + lK]
            d2 += a[k] * a[l] * (-diag[:, None] + blk + blk_t - diag[None, :])
    return np.sqrt(d2)


def _difference_covariance(inner, K):
    """``cov(A_i, A_j)`` from the log-space Theta of (A-rows, state-rows) (mbar.py:1268-1281)."""
    diag = np.ones(2 * K, dtype=np.float64)
    diag[0:K] = diag[K : 2 * K] = inner["observables"] - inner["Amin"]
    Theta = (diag[:, None] * inner["Theta"]) * diag[None, :]
    cov = np.array(Theta[0:K, 0:K] + Theta[K : 2 * K, K : 2 * K] - Theta[0:K, K : 2 * K] - Theta[K : 2 * K, 0:K])
    return Theta, cov


def compute_expectations(mbar, A_n, u_kn=None, output="averages", state_dependent=False, compute_uncertainty=True,
                         uncertainty_method=None, warning_cutoff=1.0e-10, return_theta=False):
    """Expectation of an observable at every state of ``u_kn`` (default: the sampled states), as averages or as
    differences between states, with uncertainties (pymbar/mbar.py:1039-1312)."""
    if uncertainty_method == "bootstrap" and (mbar.n_bootstraps is None or mbar.n_bootstraps <= 0):
        raise ParameterError("Cannot request bootstrap sampling of expectations without any bootstraps.")
    dims = len(np.shape(A_n))
    if dims > 2:
        logger.warning("dim=3 (state_dependent) / dim=2 observables in K x N_max form are deprecated; use N-shaped inputs.")
    if not state_dependent:
        if dims == 2:
            A_n = kn_to_n(A_n, N_k=mbar.N_k)
            if u_kn is not None:
                if len(np.shape(u_kn)) == 3:
                    u_kn = kln_to_kn(u_kn, N_k=mbar.N_k)
                elif len(np.shape(u_kn)) == 2:
                    u_kn = kn_to_n(u_kn, N_k=mbar.N_k)
    else:
        if dims == 3:
            A_n = kln_to_kn(A_n, N_k=mbar.N_k)
            if u_kn is not None:
                if len(np.shape(u_kn)) == 3:
                    u_kn = kln_to_kn(u_kn, N_k=mbar.N_k)
                elif len(np.shape(u_kn)) == 2:
                    u_kn = kn_to_n(u_kn, N_k=mbar.N_k)
    if u_kn is None:
        u_kn = mbar.u_kn
    K = 1 if len(np.shape(u_kn)) == 1 else np.shape(u_kn)[0]
    state_map = np.zeros([2, K], int)
    state_map[0, :] = np.arange(K)
    if state_dependent:
        state_map[1, :] = np.arange(K)
    inner = compute_expectations_inner(mbar, A_n, u_kn, state_map, return_theta=compute_uncertainty,
                                       uncertainty_method=uncertainty_method, warning_cutoff=warning_cutoff)
    result_vals = dict()
    Theta = covA_ij = None
    if (compute_uncertainty and uncertainty_method != "bootstrap") or return_theta:
        if "Theta" not in inner:
            inner = compute_expectations_inner(mbar, A_n, u_kn, state_map, return_theta=True,
                                               uncertainty_method=uncertainty_method, warning_cutoff=warning_cutoff)
        Theta, covA_ij = _difference_covariance(inner, K)
    if output == "averages":
        result_vals["mu"] = inner["observables"]
        if compute_uncertainty:
            if uncertainty_method == "bootstrap":
                result_vals["sigma"] = np.std(inner["bootstrapped_observables"], axis=0)
            else:
                result_vals["sigma"] = np.sqrt(covA_ij[0:K, 0:K].diagonal())
    if output == "differences":
        A_im = inner["observables"]
        result_vals["mu"] = A_im - np.vstack(A_im)
        if compute_uncertainty:
            if uncertainty_method == "bootstrap":
                Ab = inner["bootstrapped_observables"]
                result_vals["sigma"] = np.std(Ab[:, np.newaxis, :] - Ab[:, :, np.newaxis], axis=0)
            else:
                result_vals["sigma"] = mbar._ErrorOfDifferences(covA_ij, warning_cutoff=warning_cutoff)
    if return_theta:
        result_vals["Theta"] = Theta
    return result_vals


def compute_multiple_expectations(mbar, A_in, u_n, compute_uncertainty=True, compute_covariance=False,
                                  uncertainty_method=None, warning_cutoff=1.0e-10, return_theta=False):
    """Several observables at one state ``u_n`` with uncertainties / covariances (pymbar/mbar.py:1315-1439)."""
    A_in = np.asarray(A_in, dtype=np.float64)
    I = A_in.shape[0]
    if A_in.ndim == 3:
        A_in = np.array([kn_to_n(A_in[i], N_k=mbar.N_k) for i in range(I)])
    if len(np.shape(u_n)) == 2:
        u_n = kn_to_n(u_n, N_k=mbar.N_k)
    state_map = np.zeros([2, I], int)
    state_map[1, :] = np.arange(I)
    inner = compute_expectations_inner(mbar, A_in, u_n, state_map, return_theta=(compute_uncertainty or compute_covariance or return_theta),
                                       uncertainty_method=uncertainty_method, warning_cutoff=warning_cutoff)
    result_vals = dict(mu=inner["observables"])
    if compute_uncertainty or compute_covariance or return_theta:
        Theta, covA_ij = _difference_covariance(inner, I)
        if compute_uncertainty:
            result_vals["sigma"] = np.sqrt(covA_ij[0:I, 0:I].diagonal())
        if compute_covariance:
            result_vals["covariances"] = inner["Theta"][0:I, 0:I]
        if return_theta:
            result_vals["Theta"] = Theta
    if uncertainty_method == "bootstrap":
        if compute_uncertainty:
            result_vals["sigma"] = np.std(inner["bootstrapped_observables"], axis=0)
        if compute_covariance:
            result_vals["covariances"] = np.cov(inner["bootstrapped_observables"].T)
    return result_vals


def compute_perturbed_free_energies(mbar, u_ln, compute_uncertainty=True, uncertainty_method=None, warning_cutoff=1.0e-10):
    """Free energy differences among new states ``u_ln`` (L, N) (pymbar/mbar.py:1442-1521)."""
    if len(np.shape(u_ln)) == 3:
        u_ln = kln_to_kn(u_ln, N_k=mbar.N_k)
    u_ln = np.asarray(u_ln, dtype=np.float64)
    L, N = u_ln.shape
    if N < mbar.N:
        raise DataError("There seems to be too few samples in u_kn. You must evaluate at the new potential with all "
                        "of the samples used originally.")
    inner = compute_expectations_inner(mbar, np.array([0]), u_ln, np.arange(L), return_theta=compute_uncertainty,
                                       uncertainty_method=uncertainty_method, warning_cutoff=warning_cutoff)
    f_k = inner["f"]
    result_vals = dict(Delta_f=f_k - np.vstack(f_k))
    if compute_uncertainty:
        if uncertainty_method == "bootstrap":
            # (the reference returns the per-state spread here, not the spread of differences: mbar.py:1514)
            result_vals["dDelta_f"] = np.std(inner["bootstrapped_f"], axis=0)
        else:
            result_vals["dDelta_f"] = mbar._ErrorOfDifferences(inner["Theta"], warning_cutoff=warning_cutoff)
    return result_vals


def compute_entropy_and_enthalpy(mbar, u_kn=None, uncertainty_method=None, verbose=False, warning_cutoff=1.0e-10):
    """Decomposition of free energy differences into reduced enthalpy and entropy differences
    (pymbar/mbar.py:1524-1681)."""
    if verbose:
        logger.info("Computing average energy and entropy by MBAR.")
    if u_kn is not None and len(np.shape(u_kn)) == 3:
        u_kn = kln_to_kn(u_kn, N_k=mbar.N_k)
    if u_kn is None:
        u_kn = mbar.u_kn
    K, N = np.shape(u_kn)
    state_map = np.vstack([np.arange(K), np.arange(K)])
    inner = compute_expectations_inner(mbar, u_kn, u_kn, state_map, return_theta=True,  # (the observables are copied there)
                                       uncertainty_method=uncertainty_method, warning_cutoff=warning_cutoff)
    # covariance of (ln c_Ua, ln c_a, ln c_a again) -> u, f and s = u - f   (mbar.py:1600-1610)
    Theta = np.zeros([3 * K, 3 * K], dtype=np.float64)
    Theta[0 : 2 * K, 0 : 2 * K] = inner["Theta"]
    Theta[2 * K : 3 * K, :] = Theta[K : 2 * K, :]
    Theta[:, 2 * K : 3 * K] = Theta[:, K : 2 * K]
    diag = np.ones(3 * K, dtype=np.float64)
    diag[0:K] = diag[K : 2 * K] = inner["observables"] - inner["Amin"]
    Theta = (diag[:, None] * Theta) * diag[None, :]
    f_k = inner["f"]
    u_k = inner["observables"]
    s_k = u_k - f_k
    result_vals = dict(Delta_f=f_k - np.vstack(f_k), Delta_u=u_k - np.vstack(u_k), Delta_s=s_k - np.vstack(s_k))
    if uncertainty_method == "bootstrap":
        fb = mbar.f_k_boots
        ub = inner["bootstrapped_observables"]
        sb = ub - fb
        for name, arr in (("dDelta_f", fb), ("dDelta_u", ub), ("dDelta_s", sb)):
            result_vals[name] = np.std(arr[:, np.newaxis, :] - arr[:, :, np.newaxis], axis=0)
    else:
        covf = Theta[2 * K : 3 * K, 2 * K : 3 * K]
        covu = Theta[0:K, 0:K] + Theta[K : 2 * K, K : 2 * K] - Theta[0:K, K : 2 * K] - Theta[K : 2 * K, 0:K]
        covs = (covu + covf + Theta[0:K, 2 * K : 3 * K] + Theta[2 * K : 3 * K, 0:K]
                - Theta[K : 2 * K, 2 * K : 3 * K] - Theta[2 * K : 3 * K, K : 2 * K])
        result_vals["dDelta_f"] = mbar._ErrorOfDifferences(covf, warning_cutoff=warning_cutoff)
        result_vals["dDelta_u"] = mbar._ErrorOfDifferences(covu, warning_cutoff=warning_cutoff)
        result_vals["dDelta_s"] = mbar._ErrorOfDifferences(covs, warning_cutoff=warning_cutoff)
    return result_vals
