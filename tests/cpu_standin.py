"""CPU stand-in for ``pymbar_amd.device.DeviceMatrix`` -- TEST INFRASTRUCTURE ONLY.

Implements the handle interface that ``pymbar_amd.mbar_solvers`` drives, on top of the numpy oracle
(``oracle/mbar_oracle.py``), so that the host-side protocol logic and the N-sharded algorithm
(partial sums + all-reduce, the same decomposition ``libmbar_hip.so`` uses across GPUs) can be tested
without a GPU, including world_size-2 ``gloo`` runs.  Nothing under ``pymbar_amd/`` imports this.
"""
import numpy as np
from scipy.special import logsumexp

from oracle import mbar_oracle as oracle


class OracleMatrix:
    def __init__(self, u_shard, allreduce=None):
        self.u = np.ascontiguousarray(u_shard, dtype=np.float64)
        self.K, self.N_local = self.u.shape
        self.allreduce = allreduce  # fn(array, "sum"|"max") in place, or None
        self.Nk = None
        self.offset = None
        self.calls = dict(eval=0, gram=0, lognum=0)

    shape = property(lambda self: (self.K, self.N_local))

    @classmethod
    def empty(cls, K, N_local, device=None):
        return cls(np.zeros((K, N_local)))

    @classmethod
    def from_host(cls, u_kn, device=None, columns=None):
        u_kn = np.asarray(u_kn, dtype=np.float64)
        if columns is not None:
            u_kn = u_kn[:, columns[0] : columns[1]]
        return cls(np.array(u_kn))  # (an upload is a copy: the device matrix must not follow the host array afterwards)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        pass

    def upload_rows(self, row0, rows):
        rows = np.atleast_2d(np.asarray(rows, dtype=np.float64))
        self.u[row0:row0 + rows.shape[0]] = rows

    def copy_rows_from(self, src, dst_row0=0, src_row0=0, nrows=None):
        nrows = src.K - src_row0 if nrows is None else nrows
        self.u[dst_row0:dst_row0 + nrows] = src.u[src_row0:src_row0 + nrows]

    def row_sub(self, row, v_n=None):
        if v_n is not None:
            self._last_v = np.asarray(v_n, dtype=np.float64).copy()
        self.u[row] -= self._last_v

    def rows_sub(self, dst_row0, src_row0, nrows, v_n=None):
        if v_n is not None:
            self._last_v = np.asarray(v_n, dtype=np.float64).copy()
        self.u[dst_row0:dst_row0 + nrows] = self.u[src_row0:src_row0 + nrows] - self._last_v

    def rows_rsub(self, dst_row0, src_row0, nrows):
        self.u[dst_row0:dst_row0 + nrows] = self.u[src_row0:src_row0 + nrows] - self.u[dst_row0:dst_row0 + nrows]

    def rows_logshift(self, row0, nrows):
        rows = self.u[row0:row0 + nrows]
        amin = rows.min(axis=1)
        shift = amin - np.abs(4.0 * np.finfo(np.float64).eps * amin)
        with np.errstate(divide="ignore"):
            rows[...] = np.log(rows - shift[:, None])
        return shift

    def vec_logshift(self, A_n):
        A_n = np.asarray(A_n, dtype=np.float64)
        amin = A_n.min()
        shift = amin - np.abs(4.0 * np.finfo(np.float64).eps * amin)
        with np.errstate(divide="ignore"):
            self._last_v = np.log(A_n - shift)
        return float(shift)

    def fill_masked_rows(self, row0, nrows, v_n, label_n):
        v_n = np.asarray(v_n, dtype=np.float64)
        label_n = np.asarray(label_n)
        for i in range(nrows):
            self.u[row0 + i] = np.where(label_n == i, v_n, np.inf)

    def to_host(self):
        return self.u.copy()

    def set_Nk(self, N_k):
        self.Nk = np.asarray(N_k, dtype=np.float64).copy()

    def set_sample_weights(self, c_n):
        """Multiplicities are emulated by materialising the resampled matrix (column n repeated c_n times)."""
        if not hasattr(self, "u_full"):
            self.u_full = self.u
        self._wreal = None
        if c_n is None:
            self.u = self.u_full
        else:
            c = np.asarray(c_n)
            assert np.all(c == np.round(c)) and c.shape == (self.u_full.shape[1],)
            self.u = np.repeat(self.u_full, c.astype(int), axis=1)

    def draw_bootstrap_weights(self, seed, replicate, cumN, order=None, n_global0=0, layout_key=None):
        """Model of mbar_ctx_draw_bootstrap_weights: the draw counts of the counter-based stream (the host function of the C
        library yields the draws; it needs no GPU)."""
        from pymbar_amd import _lib

        rints = _lib.bootstrap_draws(seed, replicate, cumN, order)
        self.set_sample_weights(np.bincount(rints, minlength=self.u_full.shape[1] if hasattr(self, "u_full") else self.u.shape[1]))

    def weights_from_vec(self, power):
        """Real-valued per-sample weights (A - shift)**power from the observable vec_logshift left behind (model of
        mbar_ctx_weights_from_vec): they enter the log-space numerators and the W^T W sums."""
        self._wreal = np.exp(float(power) * self._last_v)

    def _reduce(self, arr, op="sum"):
        if self.allreduce is not None:
            self.allreduce(arr, op)
        return arr

    def _logden(self, f):
        return oracle.log_denominator(self.u, self.Nk, f)

    def eval(self, f, gram=False, use_offset=False):
        f = np.atleast_2d(np.asarray(f, dtype=np.float64))
        nf = f.shape[0]
        psum = np.zeros((nf, self.K))
        sld = np.zeros(nf)
        G = None
        self.calls["eval"] += 1
        for i in range(nf):
            part = oracle.shard_partials(self.u, self.Nk, f[i], want_gram=(gram and i == 0))
            psum[i] = part["psum"]
            if use_offset:
                sld[i] = np.sum(self._logden(f[i]) - self.offset)
            else:
                sld[i] = part["sumlogden"]
            if gram and i == 0:
                G = part["gram"]
                self.calls["gram"] += 1
        buf = np.concatenate([psum.ravel(), sld] + ([G.ravel()] if G is not None else []))
        self._reduce(buf)
        psum = buf[: nf * self.K].reshape(nf, self.K).copy()
        sld = buf[nf * self.K : nf * self.K + nf].copy()
        if G is not None:
            G = buf[nf * self.K + nf :].reshape(self.K, self.K).copy()
        return psum, sld, G

    def set_objective_offset(self, f0):
        self.offset = None if f0 is None else self._logden(np.asarray(f0, dtype=np.float64))

    def logden(self, f):
        return self._logden(np.asarray(f, dtype=np.float64))

    def lognum(self, f):
        self.calls["lognum"] += 1
        f = np.asarray(f, dtype=np.float64)
        x = -self._logden(f) - self.u  # (K, n)
        if getattr(self, "_wreal", None) is not None:
            with np.errstate(divide="ignore"):
                x = x + np.log(self._wreal)[None, :]
        m = np.max(x, axis=1)
        mg = self._reduce(m.copy(), "max")
        s = np.sum(np.exp(x - mg[:, None]), axis=1)
        self._reduce(s)
        return mg + np.log(s)

    def logw_kn(self, f):
        f = np.asarray(f, dtype=np.float64)
        return f[:, None] - self.u - self._logden(f)[None, :]

    def w_kn(self, f):
        return np.exp(self.logw_kn(f))

    def gram_w(self, f):
        W = np.exp(self.logw_kn(f)).T
        if getattr(self, "_wreal", None) is not None:
            Ww = W * self._wreal[:, None]
            buf = np.concatenate([(Ww.T @ W).ravel(), Ww.sum(0)])
        else:
            buf = np.concatenate([(W.T @ W).ravel(), W.sum(0)])
        self._reduce(buf)
        return buf[: self.K * self.K].reshape(self.K, self.K).copy(), buf[self.K * self.K :].copy()

    # ---- model of mbar_solve_adaptive / mbar_solve_sci (pymbar_amd/csrc/mbar_capi.cpp) -------------
    def solve_adaptive(self, f, tol=1e-12, maxiter=10000, min_sc_iter=2, gamma=1.0, check_convergence=True,
                       history_rows=0):
        f = np.array(f, dtype=np.float64)
        Nk = self.Nk
        sampled = np.where(Nk > 0)[0]
        first = sampled[0]
        psum, _, _ = self.eval(f)
        psum = psum[0]
        res = dict(iterations=0, nr_iter=0, sci_iter=0, success=False, max_delta=np.nan, gnorm=np.nan, wall_ms=0.0)
        hist = []
        for it in range(maxiter):
            _, _, gram = self.eval(f, gram=True)
            g = (psum - Nk)[sampled]
            H = (np.diag(psum) - gram)[np.ix_(sampled, sampled)]
            x = np.zeros(len(sampled))
            if len(sampled) > 1:
                try:
                    L = np.linalg.cholesky(H[1:, 1:])
                    x[1:] = np.linalg.solve(L.T, np.linalg.solve(L, g[1:]))
                except np.linalg.LinAlgError:
                    y = np.linalg.pinv(H) @ g
                    x = y - y[0]
            f_nr = f.copy()
            f_nr[sampled] = f[sampled] - gamma * x
            f_sci = f.copy()
            f_sci[sampled] = f[sampled] - np.log(psum[sampled] / Nk[sampled])
            f_sci[sampled] -= f_sci[first]
            psum2, _, _ = self.eval(np.stack([f_sci, f_nr]))
            gs = np.sum((psum2[0] - Nk)[sampled] ** 2)
            gn = np.sum((psum2[1] - Nk)[sampled] ** 2)
            f_old = f
            if gs < gn or res["sci_iter"] < min_sc_iter:
                f, psum, choice = f_sci, psum2[0], 0
                res["sci_iter"] += 1
            else:
                f, psum, choice = f_nr, psum2[1], 1
                res["nr_iter"] += 1
            rest = sampled[1:]
            div = np.abs(f[rest])
            div = np.where(div < min(1e-8, tol), 1.0, div)
            max_delta = np.max(np.abs(f[rest] - f_old[rest]) / div) if len(rest) else 0.0
            max_diff = np.max(np.abs(f_sci[rest] - f_nr[rest]) / div) if len(rest) else 0.0
            res["iterations"] = it + 1
            res["max_delta"] = max_delta
            hist.append([choice, np.sqrt(gs), np.sqrt(gn), max_delta])
            if check_convergence and (np.isnan(max_delta) or (max_delta < tol and max_diff < np.sqrt(tol))):
                res["success"] = True
                break
        res["gnorm"] = float(np.linalg.norm((psum - Nk)[sampled]))
        res["history"] = np.array(hist[:history_rows]).reshape(-1, 4)
        return f, res

    def solve_sci(self, f, tol=1e-12, maxiter=10000, check_convergence=True):
        f = np.array(f, dtype=np.float64)
        Nk = self.Nk
        sampled = np.where(Nk > 0)[0]
        res = dict(iterations=0, nr_iter=0, sci_iter=0, success=False, max_delta=np.nan, gnorm=np.nan, wall_ms=0.0)
        for it in range(maxiter):
            psum, _, _ = self.eval(f)
            fn = f.copy()
            fn[sampled] = f[sampled] - np.log(psum[0][sampled] / Nk[sampled])
            fn[sampled] -= fn[sampled[0]]
            rest = sampled[1:]
            div = np.abs(fn[rest])
            div = np.where(div < min(1e-8, tol), 1.0, div)
            delta = np.max(np.abs(fn[rest] - f[rest]) / div) if len(rest) else 0.0
            f = fn
            res["iterations"] = res["sci_iter"] = it + 1
            res["max_delta"] = delta
            if check_convergence and (np.isnan(delta) or delta < tol):
                res["success"] = True
                break
        return f, res


class OracleExtended:
    """Model of ``pymbar_amd.device.ExtendedMatrix`` (rows appended to a resident matrix without a copy of it): the resident rows
    are read through ``base``, the appended ones are a numpy block; ``lognum`` / ``gram_w`` stack the two for the oracle."""

    def __init__(self, base, nrows):
        self.base, self.Kb, self.K, self.N_local, self.nranks = base, base.K, base.K + nrows, base.N_local, 1
        self.ext = np.zeros((nrows, base.N_local))
        self._last_v = None

    def close(self):
        pass

    def _rows(self, row0, n):
        if row0 + n <= self.Kb:
            return self.base.u[row0:row0 + n]
        assert row0 >= self.Kb
        return self.ext[row0 - self.Kb:row0 - self.Kb + n]

    def upload_rows(self, row0, rows):
        rows = np.atleast_2d(np.asarray(rows, dtype=np.float64))
        assert row0 >= self.Kb
        self.ext[row0 - self.Kb:row0 - self.Kb + rows.shape[0]] = rows

    def copy_rows_from(self, src, dst_row0=0, src_row0=0, nrows=None):
        nrows = src.K - src_row0 if nrows is None else nrows
        if src is self.base and dst_row0 == 0 and src_row0 == 0 and nrows == self.Kb:
            return
        assert dst_row0 >= self.Kb
        self.ext[dst_row0 - self.Kb:dst_row0 - self.Kb + nrows] = src.u[src_row0:src_row0 + nrows]

    def rows_sub(self, dst_row0, src_row0, nrows, v_n=None):
        if v_n is not None:
            self._last_v = np.asarray(v_n, dtype=np.float64).copy()
        self.ext[dst_row0 - self.Kb:dst_row0 - self.Kb + nrows] = self._rows(src_row0, nrows) - self._last_v

    def rows_rsub(self, dst_row0, src_row0, nrows):
        d = self.ext[dst_row0 - self.Kb:dst_row0 - self.Kb + nrows]
        d[...] = self._rows(src_row0, nrows) - d

    def rows_obs_from_base(self, dst_row0, state_row0, obs_row0, nrows):
        obs = self.base.u[obs_row0:obs_row0 + nrows]
        amin = obs.min(axis=1)
        shift = amin - np.abs(4.0 * np.finfo(np.float64).eps * amin)
        with np.errstate(divide="ignore"):
            self.ext[dst_row0 - self.Kb:dst_row0 - self.Kb + nrows] = self.base.u[state_row0:state_row0 + nrows] - np.log(obs - shift[:, None])
        return shift

    def rows_logshift(self, row0, nrows):
        rows = self.ext[row0 - self.Kb:row0 - self.Kb + nrows]
        amin = rows.min(axis=1)
        shift = amin - np.abs(4.0 * np.finfo(np.float64).eps * amin)
        with np.errstate(divide="ignore"):
            rows[...] = np.log(rows - shift[:, None])
        return shift

    def vec_logshift(self, A_n):
        A_n = np.asarray(A_n, dtype=np.float64)
        amin = A_n.min()
        shift = amin - np.abs(4.0 * np.finfo(np.float64).eps * amin)
        with np.errstate(divide="ignore"):
            self._last_v = np.log(A_n - shift)
        return float(shift)

    def set_Nk(self, N_k):
        N_k = np.asarray(N_k, dtype=np.float64)
        assert np.all(N_k[self.Kb:] == 0) and np.array_equal(N_k[:self.Kb], self.base.Nk)

    def _stacked(self):
        full = OracleMatrix(np.vstack([self.base.u, self.ext]))
        full.set_Nk(np.concatenate([self.base.Nk, np.zeros(self.K - self.Kb)]))
        return full

    def lognum(self, f):
        return self._stacked().lognum(f)

    def gram_w(self, f):
        return self._stacked().gram_w(f)


def _oracle_extend(self, nrows):
    """(as the library: up to 128 resident states, 129 .. 256 rows in total; the tests lower the bounds through EXTEND_ROWS)"""
    lo, hi = OracleMatrix.EXTEND_ROWS
    Kp = (self.K + 15) // 16 * 16
    if self.allreduce is not None or not (lo < Kp + (nrows + 15) // 16 * 16 <= hi) or Kp > 128:
        return None
    return OracleExtended(self, nrows)


OracleMatrix.EXTEND_ROWS = (128, 256)
OracleMatrix.extend = _oracle_extend
