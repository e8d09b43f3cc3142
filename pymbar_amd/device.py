"""Device-resident reduced-potential matrix: the Python face of one ``mbar_ctx`` (include/mbar_hip.h).

A :class:`DeviceMatrix` holds this rank's column shard of ``u_kn`` in HBM for as long as the object
lives, so the many sweeps of a solve touch PCIe only with K-sized vectors.  It can be passed to every
function of :mod:`pymbar_amd.mbar_solvers` in place of the numpy ``u_kn`` (they upload a temporary
one when handed a numpy array).  The reference keeps ``u_kn`` in host RAM and re-sends it on every
jitted call (pymbar/mbar_solvers.py:255-257).
"""
import ctypes as C

import numpy as np

from . import _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class DeviceMatrix:
    """(K x N_local) fp64 matrix on one MI355X plus the solver state attached to it."""

    def __init__(self, K, N_local, device=None):
        _lib.require_device()
        self._lib = _lib.load_library()
        self.K = int(K)
        self.N_local = int(N_local)
        if device is None:
            import os

            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, _lib.device_count())
        self.device = int(device)
        self._ctx = C.c_void_p()
        rc = self._lib.mbar_ctx_create(C.byref(self._ctx), self.device, self.K, self.N_local)
        if rc != _lib.MBAR_OK:
            raise _lib.MbarHipError(rc, _lib.last_error(None))
        self._Nk = None
        self._version = 0  # bumped by everything that changes the matrix, N_k or the transport
        self._wtag = None  # None = unit sample multiplicities; otherwise a token of the weights last installed
        self._lognum_cache = None
        self._gram_w_cache = None
        self._cb = None  # keeps the host all-reduce callback alive
        self.nranks = 1
        self.rank = 0
        self.allreduce_kind = "none"

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def empty(cls, K, N_local, device=None):
        """A zero-filled resident matrix, to be assembled row by row (upload_rows / copy_rows_from / row_sub)."""
        return cls(K, N_local, device=device)

    @classmethod
    def from_host(cls, u_kn, device=None, columns=None):
        """Upload a host ``u_kn`` (K, N) -- or only its ``columns=(n0, n1)`` slice (this rank's shard)."""
        u_kn = np.asarray(u_kn)
        if u_kn.ndim != 2:
            raise ValueError("u_kn must be 2-D (K, N)")
        if u_kn.dtype != np.float64 or not u_kn.flags.c_contiguous:
            u_kn = np.ascontiguousarray(u_kn, dtype=np.float64)
        K, N = u_kn.shape
        n0, n1 = (0, N) if columns is None else columns
        dm = cls(K, n1 - n0, device=device)
        dm._check(dm._lib.mbar_ctx_upload_u(dm._ctx, _dptr(u_kn), N, n0, n1 - n0, 0))
        return dm

    @classmethod
    def harmonic(cls, O_k, K_k, N_k_global, seed=0, n_global0=0, N_local=None, device=None):
        """Generate the synthetic harmonic ladder of SURVEY.md 8(d) directly in HBM."""
        O_k = np.ascontiguousarray(O_k, dtype=np.float64)
        K_k = np.ascontiguousarray(K_k, dtype=np.float64)
        N_k_global = np.ascontiguousarray(N_k_global, dtype=np.int64)
        K = len(O_k)
        if N_local is None:
            N_local = int(N_k_global.sum()) - n_global0
        dm = cls(K, N_local, device=device)
        dm._check(dm._lib.mbar_ctx_generate_harmonic(
            dm._ctx, C.c_uint64(seed), _dptr(O_k), _dptr(K_k),
            N_k_global.ctypes.data_as(C.POINTER(C.c_int64)), n_global0))
        return dm

    # ---- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _lib.MBAR_OK:
            raise _lib.MbarHipError(rc, _lib.last_error(self._ctx))

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.mbar_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def shape(self):
        return (self.K, self.N_local)

    # ---- row-level assembly of an augmented matrix on the device (expectation family) ----------------------
    def upload_rows(self, row0, rows):
        """Whole rows ``[row0, row0 + len(rows))`` from a host array ``rows`` (R, N_local)."""
        self._version += 1
        rows = np.ascontiguousarray(np.atleast_2d(rows), dtype=np.float64)
        if rows.shape[1] != self.N_local:
            raise ValueError("rows must have N_local columns")
        self._check(self._lib.mbar_ctx_upload_rows(self._ctx, int(row0), rows.shape[0], _dptr(rows), rows.shape[1]))

    def copy_rows_from(self, src, dst_row0=0, src_row0=0, nrows=None):
        """Device-to-device copy of rows of another resident matrix (same device, same N_local)."""
        self._version += 1
        nrows = src.K - src_row0 if nrows is None else nrows
        self._check(self._lib.mbar_ctx_copy_rows(self._ctx, int(dst_row0), src._ctx, int(src_row0), int(nrows)))

    def row_sub(self, row, v_n=None):
        """``u[row, :] -= v_n`` on the device (v = log A_n turns a state row into an observable row).
        ``v_n=None`` subtracts the vector of the previous call again (no upload)."""
        self._version += 1
        if v_n is None:
            self._check(self._lib.mbar_ctx_row_sub(self._ctx, int(row), None))
            return
        v_n = np.ascontiguousarray(v_n, dtype=np.float64)
        if v_n.shape != (self.N_local,):
            raise ValueError("v_n must have N_local entries")
        self._check(self._lib.mbar_ctx_row_sub(self._ctx, int(row), _dptr(v_n)))

    def rows_sub(self, dst_row0, src_row0, nrows, v_n=None):
        """``u[dst_row0 + r, :] = u[src_row0 + r, :] - v_n`` for ``r < nrows`` in one launch (``v_n=None``: the vector of the
        previous call): an observable evaluated at a run of states, the state rows copied and shifted in the same pass."""
        self._version += 1
        if v_n is None:
            self._check(self._lib.mbar_ctx_rows_sub(self._ctx, int(dst_row0), int(src_row0), int(nrows), None))
            return
        v_n = np.ascontiguousarray(v_n, dtype=np.float64)
        if v_n.shape != (self.N_local,):
            raise ValueError("v_n must have N_local entries")
        self._check(self._lib.mbar_ctx_rows_sub(self._ctx, int(dst_row0), int(src_row0), int(nrows), _dptr(v_n)))

    def rows_rsub(self, dst_row0, src_row0, nrows):
        """``u[dst_row0 + r, :] = u[src_row0 + r, :] - u[dst_row0 + r, :]`` for ``r < nrows`` in one launch: rows uploaded as
        ``log A`` become observable rows ``u - log A``."""
        self._version += 1
        self._check(self._lib.mbar_ctx_rows_rsub(self._ctx, int(dst_row0), int(src_row0), int(nrows)))

    def rows_logshift(self, row0, nrows):
        """Rows ``[row0, row0 + nrows)`` hold raw observable values and become ``log(A - shift)`` in place, ``shift = min A -
        |4 eps min A|`` per row (returned): the reference's shift-to-positive + log (mbar.py:858-867, :886-903) on the device."""
        self._version += 1
        shift = np.empty(int(nrows), dtype=np.float64)
        self._check(self._lib.mbar_ctx_rows_logshift(self._ctx, int(row0), int(nrows), _dptr(shift)))
        return shift

    def vec_logshift(self, A_n):
        """One observable ``A_n`` (N_local,) into the staging vector as ``log(A - shift)``; returns ``shift``.  ``rows_sub`` /
        ``row_sub`` with ``v_n=None`` subtract it."""
        A_n = np.ascontiguousarray(A_n, dtype=np.float64)
        if A_n.shape != (self.N_local,):
            raise ValueError("A_n must have N_local entries")
        shift = np.empty(1, dtype=np.float64)
        self._check(self._lib.mbar_ctx_vec_logshift(self._ctx, _dptr(A_n), _dptr(shift)))
        return float(shift[0])

    def fill_masked_rows(self, row0, nrows, v_n, label_n):
        """Rows ``row0 + i`` (``i < nrows``) become ``v_n`` on the samples with ``label_n == i`` and ``+inf`` (weight zero)
        elsewhere: one extra "state" per histogram bin of a free energy surface, built on the device."""
        self._version += 1
        v_n = np.ascontiguousarray(v_n, dtype=np.float64)
        label_n = np.ascontiguousarray(label_n, dtype=np.int32)
        if v_n.shape != (self.N_local,) or label_n.shape != (self.N_local,):
            raise ValueError("v_n and label_n must have N_local entries")
        self._check(self._lib.mbar_ctx_fill_masked_rows(self._ctx, int(row0), int(nrows), _dptr(v_n),
                                                        label_n.ctypes.data_as(C.POINTER(C.c_int32))))

    def set_option(self, key, value):
        self._check(self._lib.mbar_ctx_set_option(self._ctx, key.encode(), int(value)))

    def synchronize(self):
        self._check(self._lib.mbar_ctx_synchronize(self._ctx))

    def to_host(self):
        from .utils import prefault

        out = prefault(np.empty((self.K, self.N_local), dtype=np.float64))
        self._check(self._lib.mbar_ctx_download_u(self._ctx, _dptr(out), self.N_local))
        return out

    def set_Nk(self, N_k):
        """GLOBAL sample counts (any numeric dtype; zeros allowed)."""
        Nk = np.ascontiguousarray(N_k, dtype=np.float64)
        if Nk.shape != (self.K,):
            raise ValueError(f"N_k must have shape ({self.K},)")
        if self._Nk is None or not np.array_equal(Nk, self._Nk):
            self._version += 1
            self._check(self._lib.mbar_ctx_set_Nk(self._ctx, _dptr(Nk)))
            self._Nk = Nk.copy()

    def set_sample_weights(self, c_n):
        """Per-sample multiplicities (``None`` restores 1): every sum over samples becomes ``sum_n c_n (...)``.
        A bootstrap replicate is ``np.bincount(resampled_indices, minlength=N)``."""
        self._wtag = None if c_n is None else object()
        if c_n is None:
            self._check(self._lib.mbar_ctx_set_sample_weights(self._ctx, None))
            return
        c_n = np.ascontiguousarray(c_n, dtype=np.float64)
        if c_n.shape != (self.N_local,):
            raise ValueError(f"sample weights must have shape ({self.N_local},)")
        self._check(self._lib.mbar_ctx_set_sample_weights(self._ctx, _dptr(c_n)))

    def draw_bootstrap_weights(self, seed, replicate, cumN, order=None, n_global0=0, layout_key=None):
        """Per-sample multiplicities = draw counts of bootstrap replicate ``replicate`` of the counter-based stream ``seed``, drawn
        ON THE DEVICE (``mbar_ctx_draw_bootstrap_weights``; :func:`pymbar_amd._lib.bootstrap_draws` gives the same draws on the
        host).  ``cumN`` (K + 1): positions of the states' runs; ``order``: sample index of a position (``None``: the default layout).
        ``layout_key``: any object that identifies (cumN, order) for the caller -- the layout is uploaded when the key changes
        (``mbar_ctx_set_bootstrap_layout``) and replicates with the same key touch no host array of N integers; without a key the
        arrays travel with every call and the library digests them to see whether its device copy still matches."""
        self._wtag = object()
        ip = C.POINTER(C.c_int64)
        if layout_key is not None and layout_key is getattr(self, "_boot_layout_key", None):
            self._check(self._lib.mbar_ctx_draw_bootstrap_weights(self._ctx, C.c_uint64(int(seed)), int(replicate), None, 0, None, int(n_global0)))
            return
        cumN = np.ascontiguousarray(cumN, dtype=np.int64)
        optr = None
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
            optr = order.ctypes.data_as(ip)
        if layout_key is not None:
            self._check(self._lib.mbar_ctx_set_bootstrap_layout(self._ctx, cumN.ctypes.data_as(ip), len(cumN) - 1, optr))
            self._boot_layout_key = layout_key
            self._check(self._lib.mbar_ctx_draw_bootstrap_weights(self._ctx, C.c_uint64(int(seed)), int(replicate), None, 0, None, int(n_global0)))
            return
        self._boot_layout_key = None
        self._check(self._lib.mbar_ctx_draw_bootstrap_weights(self._ctx, C.c_uint64(int(seed)), int(replicate), cumN.ctypes.data_as(ip),
                                                              len(cumN) - 1, optr, int(n_global0)))

    def weights_from_vec(self, power):
        """Per-sample weights ``(A_n - shift)**power`` from the observable ``vec_logshift`` left on the device (no upload, no host
        pass): the weighted sums of a single observable at the resident states.  ``set_sample_weights(None)`` restores 1."""
        self._wtag = object()
        self._check(self._lib.mbar_ctx_weights_from_vec(self._ctx, float(power)))

    # ---- multi-GPU --------------------------------------------------------------------------------
    def comm_init_rccl(self, unique_id, rank, nranks):
        self._version += 1
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.mbar_ctx_comm_init(self._ctx, buf, rank, nranks))
        self.rank, self.nranks, self.allreduce_kind = rank, nranks, "rccl"

    def set_host_allreduce(self, fn, rank, nranks):
        """``fn(array, op)`` must all-reduce the float64 numpy ``array`` in place (op 'sum'|'max')."""
        self._version += 1

        def _cb(ptr, count, op, _user):
            try:
                arr = np.ctypeslib.as_array(ptr, shape=(count,))
                fn(arr, "sum" if op == 0 else "max")
                return 0
            except Exception:  # pragma: no cover
                return 1

        self._cb = _lib.ALLREDUCE_FN(_cb)
        self._check(self._lib.mbar_ctx_set_host_allreduce(self._ctx, self._cb, None, rank, nranks))
        self.rank, self.nranks, self.allreduce_kind = rank, nranks, "host"

    def set_loopback(self, group, rank):
        """Join the in-process transport ``group`` (:class:`LoopbackGroup`) as ``rank``: collectives run on the compute stream
        like RCCL's, between contexts of this process on one device (one caller thread per context)."""
        self._version += 1
        self._check(self._lib.mbar_ctx_set_loopback(self._ctx, group._h, int(rank)))
        self._loop = group  # (keeps the group alive)
        self.rank, self.nranks, self.allreduce_kind = int(rank), group.nranks, "loopback"

    def comm_destroy(self):
        """Detach the cross-rank transport (destroys an RCCL communicator); the matrix is single-rank again."""
        self._version += 1
        self._check(self._lib.mbar_ctx_comm_destroy(self._ctx))
        self._cb = None
        self.rank, self.nranks, self.allreduce_kind = 0, 1, "none"

    def device_synchronize(self):
        """``hipDeviceSynchronize`` on this matrix's GPU (all streams)."""
        rc = self._lib.mbar_device_synchronize(self.device)
        if rc != _lib.MBAR_OK:
            raise _lib.MbarHipError(rc, _lib.last_error(None))

    # ---- L1 ---------------------------------------------------------------------------------------
    def eval(self, f, gram=False, use_offset=False):
        """One fused sweep for 1 or 2 free-energy vectors.  Returns ``(psum, sumlogden, gram)`` with
        ``psum[i, k] = sum_n N_k W_nk(f_i)``, ``sumlogden[i] = sum_n logden_n(f_i)`` and, if asked,
        ``gram = sum_n p p^T`` at ``f[0]`` (all summed over ranks)."""
        f = np.ascontiguousarray(np.atleast_2d(np.asarray(f, dtype=np.float64)))
        nf = f.shape[0]
        if f.shape[1] != self.K or nf not in (1, 2):
            raise ValueError("f must be (K,) or (1|2, K)")
        psum = np.empty((nf, self.K), dtype=np.float64)
        sld = np.empty(nf, dtype=np.float64)
        G = np.empty((self.K, self.K), dtype=np.float64) if gram else None
        flags = (_lib.EVAL_GRAM if gram else 0) | (_lib.EVAL_USE_OFFSET if use_offset else 0)
        self._check(self._lib.mbar_eval(self._ctx, _dptr(f), nf, flags, _dptr(psum), _dptr(sld),
                                        _dptr(G) if gram else None))
        return psum, sld, G

    def set_objective_offset(self, f0):
        if f0 is None:
            self._check(self._lib.mbar_ctx_set_objective_offset(self._ctx, None))
        else:
            f0 = np.ascontiguousarray(f0, dtype=np.float64)
            self._check(self._lib.mbar_ctx_set_objective_offset(self._ctx, _dptr(f0)))

    def lognum(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        out = np.empty(self.K, dtype=np.float64)
        self._check(self._lib.mbar_lognum(self._ctx, _dptr(f), _dptr(out)))
        return out

    def lognum_cached(self, f):
        """``lognum(f)``, kept while neither the matrix, N_k, the multiplicities nor ``f`` change (consecutive methods of the
        class at the same ``f_k``: one sweep of the resident rows instead of one per call)."""
        f = np.ascontiguousarray(f, dtype=np.float64)
        if self._wtag is not None:
            return self.lognum(f)
        key = (self._version, f.tobytes())
        if self._lognum_cache is None or self._lognum_cache[0] != key:
            self._lognum_cache = (key, self.lognum(f))
        return self._lognum_cache[1].copy()

    def gram_w_cached(self, f):
        """``gram_w(f)`` with unit multiplicities, kept like ``lognum_cached`` (the covariance of the free energies, overlap and
        every expectation at the same ``f_k`` start from the same ``W^T W`` of the resident states)."""
        f = np.ascontiguousarray(f, dtype=np.float64)
        if self._wtag is not None:
            return self.gram_w(f)
        key = (self._version, f.tobytes())
        if self._gram_w_cache is None or self._gram_w_cache[0] != key:
            self._gram_w_cache = (key, self.gram_w(f))
        G, ws = self._gram_w_cache[1]
        return G.copy(), ws.copy()

    def extend(self, nrows):
        """An :class:`ExtendedMatrix` of ``K + nrows`` rows whose first K rows ARE this matrix (no copy) -- or ``None`` when the
        library cannot sweep the two as one panel (more than 128 states here, fewer than 129 or more than 256 rows in total,
        several ranks): the caller then assembles an augmented copy."""
        if self.nranks != 1:
            return None
        ctx = C.c_void_p()
        rc = self._lib.mbar_ctx_create_ext(C.byref(ctx), self._ctx, int(nrows))
        if rc != _lib.MBAR_OK:
            return None
        return ExtendedMatrix(self, ctx, int(nrows))

    def logden(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        out = np.empty(self.N_local, dtype=np.float64)
        self._check(self._lib.mbar_logden(self._ctx, _dptr(f), _dptr(out)))
        return out

    def logw_kn(self, f):
        """``log W`` of this shard as a C-ordered (K, N_local) array; its ``.T`` is the reference's
        F-ordered (N, K) ``Log_W_nk``."""
        f = np.ascontiguousarray(f, dtype=np.float64)
        from .utils import prefault

        out = prefault(np.empty((self.K, self.N_local), dtype=np.float64))  # (pages touched by several threads first)
        self._check(self._lib.mbar_logw(self._ctx, _dptr(f), _dptr(out), self.N_local))
        return out

    def w_kn(self, f):
        """The weights ``W`` of this shard as a C-ordered (K, N_local) array (``exp`` taken on the device); its ``.T`` is the
        reference's (N, K) ``W_nk``."""
        from .utils import prefault

        f = np.ascontiguousarray(f, dtype=np.float64)
        out = prefault(np.empty((self.K, self.N_local), dtype=np.float64))
        self._check(self._lib.mbar_w(self._ctx, _dptr(f), _dptr(out), self.N_local))
        return out

    def gram_w(self, f):
        """``(W^T W, sum_n W_nk)`` over all states (MFMA), summed over ranks."""
        f = np.ascontiguousarray(f, dtype=np.float64)
        G = np.empty((self.K, self.K), dtype=np.float64)
        ws = np.empty(self.K, dtype=np.float64)
        self._check(self._lib.mbar_gram_w(self._ctx, _dptr(f), _dptr(G), _dptr(ws)))
        return G, ws

    # ---- solver loops -----------------------------------------------------------------------------
    def solve_adaptive(self, f, tol=1e-12, maxiter=10000, min_sc_iter=2, gamma=1.0, check_convergence=True,
                       history_rows=0):
        f = np.array(f, dtype=np.float64)
        res = _lib.SolveResult()
        hist = np.zeros((max(1, history_rows), 4), dtype=np.float64)
        self._check(self._lib.mbar_solve_adaptive(self._ctx, _dptr(f), tol, int(maxiter), int(min_sc_iter),
                                                  float(gamma), 1 if check_convergence else 0, _dptr(hist),
                                                  int(history_rows), C.byref(res)))
        out = {k: getattr(res, k) for k, _ in _lib.SolveResult._fields_ if not k.endswith("_")}
        out["success"] = bool(res.success)
        out["history"] = hist[: min(history_rows, res.iterations)]
        # per-state sums at the returned f (the solver has them anyway): gradient norm and all-state update without another sweep
        psum = np.empty(self.K, dtype=np.float64)
        out["psum"] = psum if self._lib.mbar_ctx_last_solve_psum(self._ctx, _dptr(psum)) == _lib.MBAR_OK else None
        return f, out

    def solve_sci(self, f, tol=1e-12, maxiter=10000, check_convergence=True):
        f = np.array(f, dtype=np.float64)
        res = _lib.SolveResult()
        self._check(self._lib.mbar_solve_sci(self._ctx, _dptr(f), tol, int(maxiter),
                                             1 if check_convergence else 0, C.byref(res)))
        out = {k: getattr(res, k) for k, _ in _lib.SolveResult._fields_ if not k.endswith("_")}
        out["success"] = bool(res.success)
        return f, out

    # ---- measurement ------------------------------------------------------------------------------
    def timing(self):
        """{class: (total_ms, launches)} of HIP-event timed kernels since the last reset."""
        out = {}
        for name, which in (("lse", _lib.TIMER_LSE), ("gram", _lib.TIMER_GRAM), ("reduce", _lib.TIMER_REDUCE),
                            ("other", _lib.TIMER_OTHER), ("fused", _lib.TIMER_FUSED), ("newton", _lib.TIMER_NEWTON),
                            ("comm", _lib.TIMER_COMM)):
            ms = C.c_double(0.0)
            n = C.c_int64(0)
            self._check(self._lib.mbar_ctx_timing(self._ctx, which, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def timing_reset(self):
        self._check(self._lib.mbar_ctx_timing_reset(self._ctx))

    def mfma_f64_peak(self):
        t = C.c_double(0.0)
        self._check(self._lib.mbar_mfma_f64_peak(self._ctx, C.byref(t)))
        return t.value


class ExtendedMatrix:
    """``[rows of a resident DeviceMatrix | rows appended to it]`` without a copy of the resident rows: the appended rows live in
    an extension context of the library (``mbar_ctx_create_ext``) and the sweeps read both matrices (``mbar_lognum_ext``,
    ``mbar_gram_w_ext``).  Row indices are those of the augmented matrix of pymbar/mbar.py:886-903 -- ``[0, K)`` the resident
    states, ``[K, K + R)`` the appended ones (N_k = 0) -- and the methods are the subset of :class:`DeviceMatrix` that the
    expectation family drives; rows below K are read-only."""

    def __init__(self, base, ctx, nrows):
        self.base = base
        self._lib = base._lib
        self._ctx = ctx
        self.Kb = base.K
        self.K = base.K + nrows
        self.N_local = base.N_local
        self.nranks = 1

    def _check(self, rc):
        if rc != _lib.MBAR_OK:
            raise _lib.MbarHipError(rc, _lib.last_error(self._ctx))

    def close(self):
        if self._ctx:
            self._lib.mbar_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _ext_row(self, row, n=1):
        if row < self.Kb or row + n > self.K:
            raise ValueError("only the appended rows can be written")
        return int(row - self.Kb)

    def _src(self, row0, n):
        """(context, row) of a source run: the resident matrix below K, the extension from K on (a run does not straddle)."""
        if row0 + n <= self.Kb:
            return self.base._ctx, int(row0)
        if row0 >= self.Kb:
            return self._ctx, int(row0 - self.Kb)
        raise ValueError("a source run straddles the resident and the appended rows")

    def upload_rows(self, row0, rows):
        rows = np.ascontiguousarray(np.atleast_2d(rows), dtype=np.float64)
        if rows.shape[1] != self.N_local:
            raise ValueError("rows must have N_local columns")
        self._check(self._lib.mbar_ctx_upload_rows(self._ctx, self._ext_row(row0, rows.shape[0]), rows.shape[0], _dptr(rows), rows.shape[1]))

    def copy_rows_from(self, src, dst_row0=0, src_row0=0, nrows=None):
        nrows = src.K - src_row0 if nrows is None else nrows
        if src is self.base and dst_row0 == 0 and src_row0 == 0 and nrows == self.Kb:
            return  # (the resident rows ARE the first K rows)
        self._check(self._lib.mbar_ctx_copy_rows(self._ctx, self._ext_row(dst_row0, nrows), src._ctx, int(src_row0), int(nrows)))

    def rows_sub(self, dst_row0, src_row0, nrows, v_n=None):
        sctx, srow = self._src(src_row0, nrows)
        vp = None
        if v_n is not None:
            v_n = np.ascontiguousarray(v_n, dtype=np.float64)
            if v_n.shape != (self.N_local,):
                raise ValueError("v_n must have N_local entries")
            vp = _dptr(v_n)
        self._check(self._lib.mbar_ctx_rows_sub_from(self._ctx, self._ext_row(dst_row0, nrows), sctx, srow, int(nrows), vp))

    def rows_rsub(self, dst_row0, src_row0, nrows):
        sctx, srow = self._src(src_row0, nrows)
        self._check(self._lib.mbar_ctx_rows_rsub_from(self._ctx, self._ext_row(dst_row0, nrows), sctx, srow, int(nrows)))

    def rows_obs_from_base(self, dst_row0, state_row0, obs_row0, nrows):
        """Appended rows ``dst`` = resident rows ``state`` - log(resident rows ``obs`` - shift); ``shift`` per row is returned."""
        shift = np.empty(int(nrows), dtype=np.float64)
        # (the minima of resident rows are a property of the resident matrix: kept on it, keyed by its version and the row run)
        key = (self.base._version, int(obs_row0), int(nrows))
        cache = getattr(self.base, "_rowmin_cache", None)
        mins = cache[1] if cache is not None and cache[0] == key else None
        out = np.empty(int(nrows), dtype=np.float64) if mins is None else None
        self._check(self._lib.mbar_ctx_rows_obs_from(self._ctx, self._ext_row(dst_row0, nrows), self.base._ctx, int(state_row0),
                                                     int(obs_row0), int(nrows), _dptr(shift), None if mins is None else _dptr(mins),
                                                     None if out is None else _dptr(out)))
        if mins is None:
            self.base._rowmin_cache = (key, out)
        return shift

    def rows_logshift(self, row0, nrows):
        shift = np.empty(int(nrows), dtype=np.float64)
        self._check(self._lib.mbar_ctx_rows_logshift(self._ctx, self._ext_row(row0, nrows), int(nrows), _dptr(shift)))
        return shift

    def vec_logshift(self, A_n):
        A_n = np.ascontiguousarray(A_n, dtype=np.float64)
        if A_n.shape != (self.N_local,):
            raise ValueError("A_n must have N_local entries")
        shift = np.empty(1, dtype=np.float64)
        self._check(self._lib.mbar_ctx_vec_logshift(self._ctx, _dptr(A_n), _dptr(shift)))
        return float(shift[0])

    def set_Nk(self, N_k):
        N_k = np.asarray(N_k, dtype=np.float64)
        if (N_k.shape != (self.K,) or np.any(N_k[self.Kb:] != 0) or self.base._Nk is None
                or not np.array_equal(N_k[:self.Kb], self.base._Nk)):
            raise ValueError("the appended rows are unsampled states of the resident matrix's mixture")

    def lognum(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        fb = np.ascontiguousarray(f[:self.Kb])
        out = np.empty(self.K, dtype=np.float64)
        out[:self.Kb] = self.base.lognum_cached(fb)
        ext = np.empty(self.K - self.Kb, dtype=np.float64)
        self._check(self._lib.mbar_lognum_ext(self._ctx, self.base._ctx, _dptr(fb), _dptr(ext)))
        out[self.Kb:] = ext
        return out

    def gram_w(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        fb, fe = np.ascontiguousarray(f[:self.Kb]), np.ascontiguousarray(f[self.Kb:])
        G = np.empty((self.K, self.K), dtype=np.float64)
        ws = np.empty(self.K, dtype=np.float64)
        # (a few appended rows: only their entries are swept for, the resident states' own W^T W is the one the class keeps)
        Gb = np.ascontiguousarray(self.base.gram_w_cached(fb)[0]) if self.K - self.Kb <= 16 else None
        self._check(self._lib.mbar_gram_w_ext(self._ctx, self.base._ctx, _dptr(fb), _dptr(fe), None if Gb is None else _dptr(Gb),
                                              _dptr(G), _dptr(ws)))
        return G, ws


class LoopbackGroup:
    """``mbar_loopback``: stream-ordered all-reduce between the contexts of several threads of this process on one GPU."""

    def __init__(self, nranks):
        self._lib = _lib.load_library()
        self.nranks = int(nranks)
        self._h = C.c_void_p()
        _lib.check(self._lib.mbar_loopback_create(C.byref(self._h), self.nranks))

    def close(self):
        if self._h:
            self._lib.mbar_loopback_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def device_info(device=0):
    lib = _lib.load_library()
    name = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_int64(0)
    rc = lib.mbar_device_info(device, name, 256, C.byref(cu), C.byref(mem))
    if rc != _lib.MBAR_OK:
        raise _lib.BackendUnavailable(_lib.last_error(None))
    return dict(name=name.value.decode(), compute_units=cu.value, total_mem_bytes=mem.value)
