"""Reference answers AT THE SIZES THE METRIC IS QUOTED ON (tests/golden/scale_*.npz, tests/golden/make_golden_scale.py).

The fixtures hold what the UNMODIFIED reference (pymbar numpy backend) computes for BASELINE.json config 2 at full size
(K=32, N=1e6), for the headline state count at K=128, N=1e6 (its whole class journey: MBAR(), Delta_f, dDelta_f) and for config
3 ITSELF (K=128, N=1e7: the adaptive solve from zeros, ~16 min and ~50 GB on the build container).  Outputs only: the
matrices are regenerated here from the seed by the host generator (legacy ``np.random.seed`` stream, identical across numpy
versions -- pymbar/testsystems/harmonic_oscillators.py:154-188), uploaded, and solved with the DEFAULT options of the library
(resident probability matrix, fused sweep, hipGraph batches), on one context and on 2 / 8 logical ranks.

Tolerances (BASELINE.json north_star: Deltaf_ij within 1e-8 relative, fp64): ``Delta_f`` rtol 1e-8, ``dDelta_f`` rtol 1e-7,
iteration / Newton-Raphson / self-consistent counts and the per-iteration choice identical to the reference's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from pymbar_amd import mbar_solvers as ms  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.distributed import shard_bounds  # noqa: E402
from tests.test_gpu_loopback import run_ranks  # noqa: E402

DELTA_F_RTOL, DDELTA_F_RTOL = 1e-8, 1e-7


def _delta(f):
    return f[None, :] - f[:, None]


def _assert_delta_f(f, f_ref, what):
    # Delta_f_ij = f_j - f_i within 1e-8 relative; entries that are zero by construction (the diagonal) get 1e-12 absolute
    np.testing.assert_allclose(_delta(f - f[0]), _delta(f_ref - f_ref[0]), rtol=DELTA_F_RTOL, atol=1e-12, err_msg=what)


def _assert_counts(res, g, suffix=""):
    assert res["success"] and bool(g["adaptive_success"])
    assert res["iterations"] == int(g["adaptive_iters" + suffix])
    choices = g["adaptive_choices" + suffix]
    # The LAST iteration of a converged solve compares two gradient norms at round-off level (both candidates ARE the fixed
    # point): its choice is noise in the reference itself, so the Newton-Raphson / self-consistent split may differ by that one.
    n = len(choices) - 1
    assert np.array_equal(res["history"][:n, 0].astype(np.int8), choices[:n])
    assert res["nr_iter"] + res["sci_iter"] == res["iterations"]
    assert abs(res["nr_iter"] - int(g["adaptive_nr" + suffix])) <= 1 and abs(res["sci_iter"] - int(g["adaptive_sci" + suffix])) <= 1


def _mbar_journey(u_kn, N_k, g):
    import pymbar_amd

    m = pymbar_amd.MBAR(u_kn, N_k, copy=False)
    try:
        r = m.compute_free_energy_differences(uncertainty_method="svd-ew")
        _assert_delta_f(m.f_k, g["f_k"], "MBAR().f_k")
        np.testing.assert_allclose(r["Delta_f"], g["Delta_f"], rtol=DELTA_F_RTOL, atol=1e-12)
        np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_svd_ew"], rtol=DDELTA_F_RTOL, atol=1e-12)
    finally:
        m.close()


def _logical_ranks(u_kn, N_k, nranks, g, shards=None):
    """The device-resident loop across ``nranks`` logical ranks (config-4-shaped column sharding, ONE all-reduce per iteration
    on the stream -- what runs under RCCL across GPUs): same counts as the reference, same Delta_f, ranks bit-identical."""
    from pymbar_amd.device import DeviceMatrix, LoopbackGroup

    K, N = u_kn.shape
    bounds = shards if shards is not None else [shard_bounds(N, r, nranks) for r in range(nranks)]
    with LoopbackGroup(nranks) as grp:
        def worker(r):
            n0, n1 = bounds[r]
            with DeviceMatrix.from_host(u_kn, columns=(n0, n1)) as dm:
                dm.set_loopback(grp, r)
                dm.set_Nk(N_k)
                out = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
                dm.comm_destroy()
                return out

        ranks = run_ranks(nranks, worker)
    for f, res in ranks[1:]:
        assert np.array_equal(f, ranks[0][0]) and res["iterations"] == ranks[0][1]["iterations"]
    f, res = ranks[0]
    _assert_counts(res, g)
    _assert_delta_f(f, g["f_adaptive"], f"{nranks} logical ranks")


def test_config2_full_size_matches_the_reference(golden):
    """BASELINE.json config 2 as quoted: K=32, N=1e6.  Pure self-consistent iteration (92 iterations in the reference), the
    adaptive solve with min_sc_iter 0 and 2, and the class journey."""
    from pymbar_amd.device import DeviceMatrix

    g = golden("scale_config2_K32_N1e6.npz")
    x_n, u_kn, N_k, s_n, _, _ = ts.config2(seed=int(g["seed"]))
    assert np.array_equal(N_k, g["N_k"])
    K = u_kn.shape[0]
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        f_sci, r_sci = dm.solve_sci(np.zeros(K), tol=1e-12)
        assert r_sci["success"] and r_sci["iterations"] == int(g["sci_iters"])
        _assert_delta_f(f_sci, g["f_sci"], "pure SCI")
        # the loop variants: tile streams wave-major / workgroup-major, every sweep ascending / odd iterations descending
        # (cache re-use between consecutive sweeps): the same sums regrouped (round-off only), the same 92 iterations
        for balanced, pingpong in ((0, 0), (1, 1), (0, 1)):
            dm.set_option("small_balanced", balanced)
            dm.set_option("sci_pingpong", pingpong)
            fv, rv = dm.solve_sci(np.zeros(K), tol=1e-12)
            assert rv["success"] and rv["iterations"] == int(g["sci_iters"]), (balanced, pingpong, rv)
            _assert_delta_f(fv, g["f_sci"], f"pure SCI, balanced={balanced}, pingpong={pingpong}")
        dm.set_option("small_balanced", 1)
        dm.set_option("sci_pingpong", 1)
        f, res = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
        _assert_counts(res, g)
        _assert_delta_f(f, g["f_adaptive"], "adaptive")
        f2, res2 = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=2, history_rows=64)
        _assert_counts(res2, g, "_msc2")
        _assert_delta_f(f2, g["f_adaptive_msc2"], "adaptive, min_sc_iter=2")
    _mbar_journey(u_kn, N_k, g)
    _logical_ranks(u_kn, N_k, 2, g)


def test_128_states_1e6_samples_match_the_reference(golden):
    """The headline state count at the largest N the reference's whole class journey is comfortable with (K=128, N=1e6):
    adaptive from zeros on one context, on 2 and on 8 logical ranks (config-4-shaped sharding; one more run with ragged shards:
    an EMPTY one, one of 17 columns, one of a single column, nothing tile-aligned), and MBAR() -> Delta_f / dDelta_f."""
    from pymbar_amd.device import DeviceMatrix

    g = golden("scale_K128_N1e6.npz")
    O_k, K_k, N_k = ts.config3_params(128, 1_000_000)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=int(g["seed"]))
    assert np.array_equal(N_k, g["N_k"])
    K, N = u_kn.shape
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        f, res = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
        _assert_counts(res, g)
        _assert_delta_f(f, g["f_adaptive"], "adaptive")
        assert res["builds"] == 1 and res["gram_sweeps"] == 0  # (the default path: P mode, fused, no rejected speculation)
    # the module-level function on the HOST array (upload + solve: what the unchanged reference class calls)
    f_host, r_host = ms.solve_mbar_once(u_kn, N_k, np.zeros(K), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
    _assert_delta_f(f_host, g["f_adaptive"], "solve_mbar_once(host array)")
    _mbar_journey(u_kn, N_k, g)
    _logical_ranks(u_kn, N_k, 2, g)
    _logical_ranks(u_kn, N_k, 8, g)
    assert N == 999_936  # (128 x 7812: the generator rounds N / K down)
    cuts = [0, 100_003, 100_003, 333_329, 600_000, 600_017, 777_777, N - 1, N]  # an empty shard, shards of 17 and 1 columns,
    ragged = list(zip(cuts[:-1], cuts[1:]))                                       # nothing tile-aligned
    _logical_ranks(u_kn, N_k, 8, g, shards=ragged)


def test_config3_itself_matches_the_reference(golden):
    """BASELINE.json config 3 as quoted: K=128, N=1e7 (10.24 GB regenerated on the host from the seed, ~30 s), solved with the
    default options; the reference's own f_k for this matrix is the fixture."""
    from pymbar_amd.device import DeviceMatrix

    g = golden("scale_config3_K128_N1e7.npz")
    N = int(g["N_k"].sum())
    O_k, K_k, N_k = ts.config3_params(128, N)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=int(g["seed"]))
    del x_n, s_n
    K = u_kn.shape[0]
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        f, res = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
        _assert_counts(res, g)
        _assert_delta_f(f, g["f_adaptive"], "config 3, one context")
        assert res["builds"] == 1 and res["gram_sweeps"] == 0
        # (the last iteration found both candidates inside the stop test before its sweep: that sweep ran without the Gram matrix)
        assert res["light_sweeps"] == 1
        # the same answer from the classic sweeps on u (no resident probability matrix) and from a warm start
        dm.set_option("pmode", 0)
        f0, res0 = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
        _assert_counts(res0, g)
        _assert_delta_f(f0, g["f_adaptive"], "config 3, classic sweeps")
        assert res0["light_sweeps"] == 0
        dm.set_option("pmode", 1)
    _logical_ranks(u_kn, N_k, 8, g)


def test_config4_shape_on_one_device():
    """BASELINE.json config 4 (K=128, N=1e8: 102.4 GB of u + 102.4 GB of resident probabilities) on ONE 288 GB device, in the
    shape the metric runs it in: EIGHT logical ranks of 1.25e7 columns each (the real 12.8 GB-per-GPU shard; ``mbar_loopback`` =
    ONE all-reduce per iteration on the compute streams, the code RCCL drives), generated in HBM by the shard-invariant
    generator.  The reference cannot hold this matrix, but the per-iteration sums of pymbar/mbar_solvers.py:581-594 are
    shard-additive, so the checks are: ranks bit-identical; the SAME matrix solved in one context (64-bit lane offsets: a
    different kernel family) gives the same iteration count, the same choices and ``Delta_f`` to 1e-12; the solution is the
    analytic one to sampling error; and the per-state sums / Gram matrix of one 1e5-column block agree with the CPU oracle."""
    from oracle import mbar_oracle as oracle
    from pymbar_amd.device import DeviceMatrix, LoopbackGroup, device_info

    K, N, nranks = 128, 100_000_000, 8
    if device_info()["total_mem_bytes"] < 230 * (1 << 30):
        pytest.skip("config 4 on one device needs 205 GB + work space")
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    assert int(N_k.sum()) == N
    bounds = [shard_bounds(N, r, nranks) for r in range(nranks)]
    assert all(b - a == 12_500_000 for a, b in bounds)
    with LoopbackGroup(nranks) as grp:
        def worker(r):
            n0, n1 = bounds[r]
            with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0, n_global0=n0, N_local=n1 - n0) as dm:
                dm.set_loopback(grp, r)
                dm.set_Nk(N_k)
                out = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
                dm.comm_destroy()
                return out

        ranks = run_ranks(nranks, worker, timeout=900.0)
    f8, r8 = ranks[0]
    for f, res in ranks[1:]:
        assert np.array_equal(f, f8) and res["iterations"] == r8["iterations"] and np.array_equal(res["history"], r8["history"])
    assert r8["success"] and r8["iterations"] <= 8
    assert r8["builds"] == 1 and r8["gram_sweeps"] == 0  # (default path on every rank: P mode, fused sweep)
    assert np.max(np.abs(f8 - ts.harmonic_free_energies(K_k))) < 5e-3
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
        dm.set_Nk(N_k)
        f1, r1 = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
        assert r1["success"] and r1["iterations"] == r8["iterations"]
        n = r1["iterations"] - 1  # (the last choice compares two round-off-level norms)
        assert np.array_equal(r1["history"][:n, 0], r8["history"][:n, 0])
        np.testing.assert_allclose(_delta(f1), _delta(f8), rtol=0, atol=1e-12)
        psum, sld, gram = dm.eval(f1, gram=True)
        assert abs(psum[0].sum() - N) < 1e-6 * N ** 0.5
        assert np.max(np.abs(psum[0] - N_k)) < 1e-5           # gradient vanishes (|g| relative to N_k = 781250: 1e-11)
        assert np.array_equal(gram, gram.T)
    # one block of 1e5 columns from the middle of rank 3's shard, same bits through the shard-invariant generator
    n_blk0 = bounds[3][0] + 4_000_000
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0, n_global0=n_blk0, N_local=100_000) as sub:
        u_sub = sub.to_host()
        sub.set_Nk(N_k)
        ps, sl, gs = sub.eval(f8, gram=True)
    part = oracle.shard_partials(u_sub, N_k, f8, want_gram=True)
    np.testing.assert_allclose(ps[0], part["psum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(gs, part["gram"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sl[0], part["sumlogden"], rtol=1e-13)


def test_600_states_150000_samples_against_the_oracle_loop():
    """Above 256 states at a size where round-off could tell the sweeps on the resident probability matrix from the sweeps on u
    (round 5 rebuilt this whole path: `k_gram_rect`, `host_pmode`, the host Cholesky team): the host-driven adaptive solve with
    host_pmode 2 (both sweeps on P, Gram in 256-state panels + 128 x 256 rectangles), 1 (128-state panels) and 0 (both sweeps on
    u) against the CPU oracle's loop (oracle.adaptive = mbar_solvers.py:575-640 restated; ~30 s) on the same 600 x 150 000
    matrix: iteration count, per-iteration choices, gradient norms of every iteration, free energies."""
    from oracle import mbar_oracle as oracle
    from pymbar_amd.device import DeviceMatrix

    K, N = 600, 150_000
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    _, u_kn, N_k, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=K)
    hist = []
    r_or = oracle.adaptive(np.ascontiguousarray(u_kn), N_k.astype(float), np.zeros(K), tol=1e-10, min_sc_iter=0, history=hist)
    assert r_or["success"]
    gn = np.array([[h["gnorm_sci"], h["gnorm_nr"]] for h in hist])
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for mode in (2, 1, 0):
            dm.set_option("host_pmode", mode)
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-10, maxiter=100, min_sc_iter=0, history_rows=100)
            assert ra["success"] and ra["iterations"] == r_or["iterations"], (mode, ra["iterations"], r_or["iterations"])
            np.testing.assert_allclose(fa, r_or["x"], rtol=0, atol=1e-12 * np.max(np.abs(r_or["x"])), err_msg=f"host_pmode {mode}")
            _assert_delta_f(fa, r_or["x"], f"host_pmode {mode}")
            n = ra["iterations"]
            # gradient norms of every iteration: relative, with the round-off floor of the sums (N_k ~ 250: ~1e-12 absolute)
            np.testing.assert_allclose(ra["history"][:n, 1:3], gn[:n], rtol=1e-2, atol=1e-10, err_msg=f"host_pmode {mode}")
            choices = np.array([1 if h["choice"] == "nr" else 0 for h in hist[:n - 1]])  # (the last one compares round-off)
            assert np.array_equal(ra["history"][:n - 1, 0].astype(int), choices), mode


def test_device_drawn_bootstrap_replicates_at_128_states_1e6_samples():
    """``bootstrap_rng="device"`` at the headline state count: every replicate's multiplicities drawn ON THE DEVICE equal the draw
    counts of the host face of the same counter-based stream (``mbar_bootstrap_draws``) -- the per-state sums of a sweep are then
    bit-identical -- for the default layout and for a permuted ``x_kindices``-like order (layout uploaded once, keyed), and the
    class solves exactly those replicates: ``f_k_boots`` of the device stream = the solve with the host-drawn counts as sample
    weights (1e-10); bit-identical between two objects under the same ``rseed``."""
    import pymbar_amd
    from pymbar_amd import _lib
    from pymbar_amd.device import DeviceMatrix

    K, N = 128, 1_000_000
    O_k, K_k, N_k = ts.config3_params(K, N)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
    N = u_kn.shape[1]
    cum = np.concatenate(([0], np.cumsum(N_k))).astype(np.int64)
    f = ts.harmonic_free_energies(K_k)
    perm = np.random.default_rng(2).permutation(N)
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for order, key in ((None, object()), (perm, object())):
            for rep in (0, 5):
                dm.draw_bootstrap_weights(777, rep, cum, order, layout_key=key)   # (second replicate: no array crosses the boundary)
                ps, _, _ = dm.eval(f)
                rints = _lib.bootstrap_draws(777, rep, cum, order)
                counts = np.bincount(rints, minlength=N)
                # every state keeps its sample count (resampling WITHIN states, mbar.py:425-433)
                owner = np.searchsorted(cum, np.arange(N) if order is None else np.argsort(perm), side="right") - 1
                assert np.array_equal(np.bincount(owner, weights=counts, minlength=K).astype(np.int64), N_k)
                dm.set_sample_weights(counts)
                ps_ref, _, _ = dm.eval(f)
                assert np.array_equal(ps[0], ps_ref[0]), (rep, order is None)
        dm.set_sample_weights(None)
    a = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=3, rseed=5, bootstrap_rng="device", copy=False)
    b = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=3, rseed=5, bootstrap_rng="device", copy=False)
    try:
        assert a.bootstrap_rng_used == "device" and a._bootstrap_rints is None
        assert np.array_equal(a.f_k_boots, b.f_k_boots)
        seed, cumN, order = a._bootstrap_stream
        with DeviceMatrix.from_host(u_kn) as dm:
            for i in range(3):
                dm.set_sample_weights(np.bincount(_lib.bootstrap_draws(seed, i, cumN, order), minlength=N))
                fr = ms.solve_mbar_for_all_states(dm, N_k, a.f_k.copy(), a.states_with_samples, a._bootstrap_protocol)
                # (the object's matrix starts each replicate from the resident probability matrix of its main solve, this one
                # builds its own: the same replicate solved from two anchors -- equal to the solver's tolerance, not bit for bit)
                np.testing.assert_allclose(fr, a.f_k_boots[i], rtol=0, atol=1e-10, err_msg=f"replicate {i}")
        # the spread of the replicates is the bootstrap error estimate: the right order of magnitude against the analytic covariance
        r = a.compute_free_energy_differences(uncertainty_method="svd-ew")
        sd = a.f_k_boots.std(0, ddof=1)
        assert np.all(sd[1:] > 0) and np.median(sd[1:] / r["dDelta_f"][0, 1:]) < 5.0
    finally:
        a.close()
        b.close()
