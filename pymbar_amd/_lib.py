"""ctypes binding of libmbar_hip.so (include/mbar_hip.h).  Thin: no numpy logic lives here.

The product path has exactly one compute backend -- the gfx950 library.  If it cannot be loaded,
or no MI355X is visible, every entry point raises :class:`BackendUnavailable`; there is no CPU
fallback (the numpy oracle under ``oracle/`` is test infrastructure and is never imported here).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MBAR_HIP_LIBRARY") or os.path.join(HERE, "csrc", "libmbar_hip.so")  # (override: A/B builds)

MBAR_OK = 0
EVAL_GRAM = 1
EVAL_USE_OFFSET = 2
TIMER_LSE, TIMER_GRAM, TIMER_REDUCE, TIMER_OTHER, TIMER_FUSED, TIMER_NEWTON, TIMER_COMM = 0, 1, 2, 3, 4, 5, 6


class BackendUnavailable(RuntimeError):
    """libmbar_hip.so is missing / not loadable, or there is no gfx950 device."""


class MbarHipError(RuntimeError):
    """A libmbar_hip call returned an error code."""

    def __init__(self, code, message):
        super().__init__(f"libmbar_hip error {code}: {message}")
        self.code = code


class SolveResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int64),
        ("nr_iter", C.c_int64),
        ("sci_iter", C.c_int64),
        ("success", C.c_int32),
        ("gram_sweeps", C.c_int32),
        ("max_delta", C.c_double),
        ("gnorm", C.c_double),
        ("wall_ms", C.c_double),
        ("warm_starts", C.c_int32),
        ("builds", C.c_int32),
        ("light_sweeps", C.c_int32),
        ("reserved_", C.c_int32),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_ctx = C.c_void_p

# name -> (restype, argtypes); this table is also what tests/test_capi_symbols.py checks against
# the declarations in include/mbar_hip.h
SIGNATURES = {
    "mbar_version": (C.c_int, []),
    "mbar_last_error": (C.c_char_p, [_ctx]),
    "mbar_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mbar_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), _ip]),
    "mbar_ctx_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_int64, C.c_int64]),
    "mbar_ctx_destroy": (None, [_ctx]),
    "mbar_ctx_synchronize": (C.c_int, [_ctx]),
    "mbar_device_synchronize": (C.c_int, [C.c_int]),
    "mbar_cache_trim": (C.c_int, []),
    "mbar_host_digest": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_uint64)]),
    "mbar_host_newton_direction": (C.c_int, [_dp, _dp, C.c_int, C.c_int, _dp]),
    "mbar_ctx_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_int64]),
    "mbar_ctx_upload_u": (C.c_int, [_ctx, _dp, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "mbar_ctx_download_u": (C.c_int, [_ctx, _dp, C.c_int64]),
    "mbar_ctx_upload_rows": (C.c_int, [_ctx, C.c_int64, C.c_int64, _dp, C.c_int64]),
    "mbar_ctx_copy_rows": (C.c_int, [_ctx, C.c_int64, _ctx, C.c_int64, C.c_int64]),
    "mbar_ctx_row_sub": (C.c_int, [_ctx, C.c_int64, _dp]),
    "mbar_ctx_rows_sub": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_int64, _dp]),
    "mbar_ctx_rows_rsub": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_int64]),
    "mbar_ctx_rows_logshift": (C.c_int, [_ctx, C.c_int64, C.c_int64, _dp]),
    "mbar_ctx_vec_logshift": (C.c_int, [_ctx, _dp, _dp]),
    "mbar_ctx_fill_masked_rows": (C.c_int, [_ctx, C.c_int64, C.c_int64, _dp, C.POINTER(C.c_int32)]),
    "mbar_ctx_generate_harmonic": (C.c_int, [_ctx, C.c_uint64, _dp, _dp, _ip, C.c_int64]),
    "mbar_ctx_set_Nk": (C.c_int, [_ctx, _dp]),
    "mbar_ctx_set_sample_weights": (C.c_int, [_ctx, _dp]),
    "mbar_ctx_weights_from_vec": (C.c_int, [_ctx, C.c_double]),
    "mbar_ctx_draw_bootstrap_weights": (C.c_int, [_ctx, C.c_uint64, C.c_int64, _ip, C.c_int64, _ip, C.c_int64]),
    "mbar_ctx_set_bootstrap_layout": (C.c_int, [_ctx, _ip, C.c_int64, _ip]),
    "mbar_bootstrap_draws": (C.c_int, [C.c_uint64, C.c_int64, _ip, C.c_int64, _ip, _ip]),
    "mbar_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mbar_ctx_comm_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "mbar_ctx_set_host_allreduce": (C.c_int, [_ctx, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int]),
    "mbar_ctx_comm_destroy": (C.c_int, [_ctx]),
    "mbar_loopback_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "mbar_loopback_destroy": (None, [C.c_void_p]),
    "mbar_ctx_set_loopback": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "mbar_eval": (C.c_int, [_ctx, _dp, C.c_int, C.c_uint, _dp, _dp, _dp]),
    "mbar_ctx_set_objective_offset": (C.c_int, [_ctx, _dp]),
    "mbar_lognum": (C.c_int, [_ctx, _dp, _dp]),
    "mbar_logden": (C.c_int, [_ctx, _dp, _dp]),
    "mbar_logw": (C.c_int, [_ctx, _dp, _dp, C.c_int64]),
    "mbar_w": (C.c_int, [_ctx, _dp, _dp, C.c_int64]),
    "mbar_gram_w": (C.c_int, [_ctx, _dp, _dp, _dp]),
    "mbar_ctx_create_ext": (C.c_int, [C.POINTER(_ctx), _ctx, C.c_int64]),
    "mbar_ctx_rows_sub_from": (C.c_int, [_ctx, C.c_int64, _ctx, C.c_int64, C.c_int64, _dp]),
    "mbar_ctx_rows_rsub_from": (C.c_int, [_ctx, C.c_int64, _ctx, C.c_int64, C.c_int64]),
    "mbar_ctx_rows_obs_from": (C.c_int, [_ctx, C.c_int64, _ctx, C.c_int64, C.c_int64, C.c_int64, _dp, _dp, _dp]),
    "mbar_lognum_ext": (C.c_int, [_ctx, _ctx, _dp, _dp]),
    "mbar_gram_w_ext": (C.c_int, [_ctx, _ctx, _dp, _dp, _dp, _dp, _dp]),
    "mbar_solve_adaptive": (C.c_int, [_ctx, _dp, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                      _dp, C.c_int64, C.POINTER(SolveResult)]),
    "mbar_ctx_last_solve_psum": (C.c_int, [_ctx, _dp]),
    "mbar_solve_sci": (C.c_int, [_ctx, _dp, C.c_double, C.c_int64, C.c_int, C.POINTER(SolveResult)]),
    "mbar_ctx_timing": (C.c_int, [_ctx, C.c_int, _dp, _ip]),
    "mbar_ctx_timing_reset": (C.c_int, [_ctx]),
    "mbar_mfma_f64_peak": (C.c_int, [_ctx, _dp]),
}

_lib = None


def load_library():
    """Load libmbar_hip.so and attach signatures.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pymbar_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the ROCm install
        raise BackendUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error(ctx=None):
    msg = load_library().mbar_last_error(ctx)
    return msg.decode("utf-8", "replace") if msg else ""


def check(code, ctx=None):
    if code != MBAR_OK:
        raise MbarHipError(code, last_error(ctx))


def host_digest(a, threads=0):
    """16-byte digest of every byte of a C-contiguous numpy array (``mbar_host_digest``; host only, no GPU needed)."""
    out = (C.c_uint64 * 2)()
    check(load_library().mbar_host_digest(C.c_void_p(a.ctypes.data), a.nbytes, int(threads), out))
    return bytes(out)


def bootstrap_draws(seed, replicate, cumN, order=None):
    """The draws of replicate ``replicate`` of the counter-based bootstrap stream ``seed`` as the reference's ``bootstrap_rints`` row
    (``mbar_bootstrap_draws``; host only, no GPU needed): ``rints[sample of slot j] = sample drawn``, state by state."""
    import numpy as np

    cumN = np.ascontiguousarray(cumN, dtype=np.int64)
    total = int(cumN[-1])
    out = np.zeros(total, dtype=np.int64)
    ip = C.POINTER(C.c_int64)
    optr = None
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.int64)
        optr = order.ctypes.data_as(ip)
    check(load_library().mbar_bootstrap_draws(C.c_uint64(int(seed)), int(replicate), cumN.ctypes.data_as(ip), len(cumN) - 1, optr,
                                              out.ctypes.data_as(ip)))
    return out


def host_newton_direction(H, g, threads=0):
    """``x = H^+ g - (H^+ g)[0]`` as the host-driven adaptive loop computes it (``mbar_host_newton_direction``; host only).
    ``threads > 0`` forces the blocked, threaded Cholesky factorisation with that team size (tests)."""
    import numpy as np

    H = np.ascontiguousarray(H, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    m = g.shape[0]
    if H.shape != (m, m):
        raise ValueError("H must be (m, m) for g of length m")
    x = np.empty(m, dtype=np.float64)
    check(load_library().mbar_host_newton_direction(H.ctypes.data_as(_dp), g.ctypes.data_as(_dp), m, int(threads), x.ctypes.data_as(_dp)))
    return x


def trim_device_cache():
    """Hand every device / pinned block this library has parked for re-use back to the driver (``mbar_cache_trim``; the bound
    of the cache is ``MBAR_CACHE_MB``, see include/mbar_hip.h)."""
    check(load_library().mbar_cache_trim())


def device_count():
    lib = load_library()
    n = C.c_int(0)
    rc = lib.mbar_device_count(C.byref(n))
    return n.value if rc == MBAR_OK else 0


def require_device():
    """Raise BackendUnavailable unless a gfx950 GPU can be used."""
    if device_count() < 1:
        raise BackendUnavailable(
            "no HIP device visible: pymbar_amd computes only on an MI355X (gfx950) through "
            "libmbar_hip.so and has no CPU fallback")
