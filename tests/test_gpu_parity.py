"""Parity of the gfx950 path (through the C ABI, include/mbar_hip.h) with the CPU oracle and with the
reference's golden fixtures.  Everything here needs a real MI355X: run with ``-m gpu``.

Tolerances (BASELINE.json: Deltaf_ij within 1e-8 relative, fp64): L1 quantities are compared at
1e-10 relative or tighter -- they differ from the oracle only by summation order and by the ulp-level
difference between the device exp/log and libm; solutions at 1e-9 absolute."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mbar_oracle as oracle  # noqa: E402
from pymbar_amd import mbar_solvers as ms  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402


@pytest.fixture(scope="module")
def DM():
    from pymbar_amd.device import DeviceMatrix

    return DeviceMatrix


def random_problem(K, N, seed, unsampled=(), spread=1.0):
    rng = np.random.RandomState(seed)
    O_k = np.linspace(0.0, 3.0, K) * spread
    K_k = np.linspace(1.0, 2.5, K)
    N_k = rng.multinomial(N, np.ones(K) / K) if K > 1 else np.array([N])
    for k in unsampled:
        N_k[0] += N_k[k]
        N_k[k] = 0
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=seed)
    f = ts.harmonic_free_energies(K_k) + 0.2 * rng.standard_normal(K)
    f[0] = 0.0
    return u_kn, N_k, f


def check_l1(dm, u_kn, N_k, f, tag=""):
    Nf = N_k.astype(float)
    dm.set_Nk(N_k)
    psum, sld, gram = dm.eval(f, gram=True)
    part = oracle.shard_partials(u_kn, N_k, f, want_gram=True)
    scale = max(1.0, Nf.max())
    np.testing.assert_allclose(psum[0], part["psum"], rtol=1e-11, atol=1e-11 * scale, err_msg=tag + " psum")
    np.testing.assert_allclose(sld[0], part["sumlogden"], rtol=1e-12, atol=1e-9, err_msg=tag + " sumlogden")
    np.testing.assert_allclose(gram, part["gram"], rtol=1e-10, atol=1e-12 * scale, err_msg=tag + " gram")
    assert np.array_equal(gram, gram.T)
    np.testing.assert_allclose(ms.mbar_gradient(dm, N_k, f), oracle.mbar_gradient(u_kn, N_k, f), rtol=1e-9,
                               atol=1e-10 * scale, err_msg=tag + " gradient")
    np.testing.assert_allclose(ms.mbar_hessian(dm, N_k, f), oracle.mbar_hessian(u_kn, N_k, f), rtol=1e-9,
                               atol=1e-11 * scale, err_msg=tag + " hessian")
    np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, f), oracle.self_consistent_update(u_kn, N_k, f),
                               rtol=1e-12, atol=1e-11, err_msg=tag + " sci")
    np.testing.assert_allclose(dm.logden(f), oracle.log_denominator(u_kn, N_k, f), rtol=1e-13, atol=1e-12,
                               err_msg=tag + " logden")
    # two candidates in one sweep == two single sweeps
    f2 = f + 0.05 * np.cos(np.arange(len(f)))
    f2[0] = 0
    ps2, sl2, _ = dm.eval(np.stack([f, f2]))
    np.testing.assert_allclose(ps2[0], psum[0], rtol=1e-13, atol=1e-13 * scale)
    np.testing.assert_allclose(ps2[1], oracle.shard_partials(u_kn, N_k, f2)["psum"], rtol=1e-11, atol=1e-11 * scale,
                               err_msg=tag + " psum(second f)")


@pytest.mark.parametrize("K,N", [(1, 40), (2, 7), (3, 16), (5, 5000), (16, 1000), (17, 999), (32, 4096), (40, 2001),
                                 (64, 3000), (100, 1500), (112, 800), (128, 2048), (128, 33)])
def test_l1_parity_fast_path(DM, K, N):
    u_kn, N_k, f = random_problem(K, N, seed=K * 1000 + N)
    with DM.from_host(u_kn) as dm:
        check_l1(dm, u_kn, N_k, f, tag=f"K={K} N={N}")
        np.testing.assert_array_equal(dm.to_host(), u_kn)


@pytest.mark.parametrize("K,N", [(96, 1500), (100, 777), (112, 800), (128, 2048), (128, 100000), (192, 1200), (256, 600)])
@pytest.mark.parametrize("small_wide", [1, 0])
def test_l1_parity_wide_panels(DM, K, N, small_wide):
    """Wide panels (6 .. 16 blocks of 16 states): evaluation sweep with one tile stream per wave (and, for 129 .. 256 states, the
    general sweep in place of the single-buffer one: ``wide_k_kernel`` 0), full 128-state Gram panel on one wave per SIMD with
    pinned accumulator classes; a short self-consistent loop on top.  (The paired-wave / early-refill / operand-exchange / register-
    staged variants of rounds 1-3 were measured never-best and removed in round 4.)"""
    u_kn, N_k, f = random_problem(K, N, seed=3 * K + N)
    N_k = np.maximum(N_k, 1)  # every state sampled (N_k only acts as a weight vector here)
    with DM.from_host(u_kn) as dm:
        dm.set_option("wide_k_kernel", small_wide)
        check_l1(dm, u_kn, N_k, f, tag=f"K={K} N={N} wide_k_kernel={small_wide}")
        fs, rs = ms.solve_mbar_once(dm, N_k, np.zeros(K), method="self-consistent-iteration", tol=1e-10,
                                    options=dict(maxiter=3))
        f_ref = np.zeros(K)
        for _ in range(3):
            f_ref = oracle.self_consistent_update(u_kn, N_k, f_ref)
            f_ref -= f_ref[0]
        # states without samples are not touched by the device SCI loop
        sampled = N_k > 0
        np.testing.assert_allclose(fs[sampled] - fs[sampled][0], f_ref[sampled] - f_ref[sampled][0], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("K,N", [(129, 1000), (160, 700), (192, 1200), (200, 513), (256, 600)])
def test_l1_parity_paneled_gram(DM, K, N):
    """128 < K <= 256: 2-wave evaluation kernel + 64-state Gram panels (diagonal and off-diagonal launches)."""
    u_kn, N_k, f = random_problem(K, N, seed=K + N)
    with DM.from_host(u_kn) as dm:
        check_l1(dm, u_kn, N_k, f, tag=f"K={K} N={N}")


@pytest.mark.parametrize("K,N,force", [(300, 900, 0), (321, 400, 0), (512, 700, 0), (257, 5003, 0), (300, 900, 1), (600, 300, 0),
                                       (40, 2001, 1), (128, 640, 1), (513, 333, 0), (700, 1000, 0), (768, 250, 0), (769, 130, 0),
                                       (1000, 2100, 0), (1024, 97, 0), (1025, 150, 0), (1100, 100, 1)])
def test_l1_parity_generic_path(DM, K, N, force):
    """K > 256: the one-read evaluation kernel whose eight waves split the rows of a tile (257 .. 1024 states; from 513 on with one
    tile buffer per wave), the layout-agnostic kernels beyond (and as the forced fallback, also for small K); paneled Gram."""
    u_kn, N_k, f = random_problem(K, N, seed=7 * K + N)
    with DM.from_host(u_kn) as dm:
        dm.set_option("force_generic", force)
        check_l1(dm, u_kn, N_k, f, tag=f"K={K} N={N} generic")


def test_unsampled_states_and_logw(DM):
    u_kn, N_k, f = random_problem(24, 3000, seed=5, unsampled=(3, 17, 23))
    with DM.from_host(u_kn) as dm:
        check_l1(dm, u_kn, N_k, f, tag="unsampled")
        logW = ms.mbar_log_W_nk(dm, N_k, f)
        assert logW.shape == (3000, 24) and logW.flags.f_contiguous
        np.testing.assert_allclose(logW, oracle.mbar_log_W_nk(u_kn, N_k, f), rtol=1e-13, atol=1e-12)
        Wd = ms.mbar_W_nk(dm, N_k, f)  # (the exponential taken on the device: mbar_w)
        assert Wd.shape == (3000, 24) and Wd.flags.f_contiguous
        np.testing.assert_allclose(Wd, oracle.mbar_W_nk(u_kn, N_k, f), rtol=1e-12, atol=1e-300)
        G, wsum = dm.gram_w(f)
        W = oracle.mbar_W_nk(u_kn, N_k, f)
        np.testing.assert_allclose(G, W.T @ W, rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(wsum, W.sum(0), rtol=1e-11)
        # masked subset (mbar_solvers.py:253-257)
        sws = np.where(N_k > 0)[0]
        np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, f, states_with_samples=sws),
                                   oracle.self_consistent_update(u_kn[sws], N_k[sws], f[sws]), rtol=1e-12, atol=1e-11)


def test_invariances_and_extreme_energies(DM):
    """Per-sample shifts leave g, H, W unchanged (mbar_solvers.py:727-734); huge energy gaps neither overflow
    nor produce NaN (the reference relies on logsumexp for this)."""
    u_kn, N_k, f = random_problem(20, 2000, seed=11)
    shift = 1.0e4 * np.sin(np.arange(2000))
    big = u_kn.copy()
    big[7] += 3000.0  # a state nobody overlaps with
    big[:, ::5] += 800.0
    with DM.from_host(u_kn) as a, DM.from_host(u_kn + shift) as b, DM.from_host(big) as c:
        for dm in (a, b, c):
            dm.set_Nk(N_k)
        pa, sa, ga = a.eval(f, gram=True)
        pb, sb, gb = b.eval(f, gram=True)
        np.testing.assert_allclose(pb, pa, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(gb, ga, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sb[0] - sa[0], -shift.sum(), rtol=1e-10)
        pc, sc, gc = c.eval(f, gram=True)
        assert np.all(np.isfinite(pc)) and np.all(np.isfinite(gc)) and np.isfinite(sc[0])
        part = oracle.shard_partials(big, N_k, f, want_gram=True)
        np.testing.assert_allclose(pc[0], part["psum"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(ms.self_consistent_update(c, N_k, f), oracle.self_consistent_update(big, N_k, f),
                                   rtol=1e-11, atol=1e-10)
        # far-off f: still finite and equal to the oracle
        f_bad = f + 50.0 * np.cos(np.arange(20))
        f_bad[0] = 0
        pbad, _, _ = a.eval(f_bad)
        np.testing.assert_allclose(pbad[0], oracle.shard_partials(u_kn, N_k, f_bad)["psum"], rtol=1e-10, atol=1e-9)


def test_sample_weights_equal_explicit_resampling(DM):
    """Bootstrap replicates as per-sample multiplicities on the resident matrix == the reference's gathered copy
    u_kn[:, rints] (mbar.py:417-449): every sweep, the all-state update, the covariance input and a whole solve."""
    u_kn, N_k, f = random_problem(24, 3000, seed=41, unsampled=(5,))
    rng = np.random.RandomState(2)
    rints = np.concatenate([np.sort(rng.randint(a, b, size=b - a)) if b > a else np.zeros(0, int)
                            for a, b in zip(np.cumsum(N_k) - N_k, np.cumsum(N_k))]).astype(int)
    u_res = np.ascontiguousarray(u_kn[:, rints])
    counts = np.bincount(rints, minlength=3000)
    assert counts.sum() == 3000 and counts.max() > 1 and (counts == 0).any()
    for force in (0, 1):
        with DM.from_host(u_kn) as dm:
            dm.set_option("force_generic", force)
            dm.set_Nk(N_k)
            dm.set_sample_weights(counts)
            ps, sl, G = dm.eval(f, gram=True)
            part = oracle.shard_partials(u_res, N_k, f, want_gram=True)
            np.testing.assert_allclose(ps[0], part["psum"], rtol=1e-11, atol=1e-10)
            np.testing.assert_allclose(sl[0], part["sumlogden"], rtol=1e-12)
            np.testing.assert_allclose(G, part["gram"], rtol=1e-10, atol=1e-11)
            np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, f), oracle.self_consistent_update(u_res, N_k, f),
                                       rtol=1e-12, atol=1e-11)
            GW, ws = dm.gram_w(f)
            W = oracle.mbar_W_nk(u_res, N_k, f)
            np.testing.assert_allclose(GW, W.T @ W, rtol=1e-10, atol=1e-14)
            np.testing.assert_allclose(ws, W.sum(0), rtol=1e-10)
            sws = np.where(N_k > 0)[0]
            fw = ms.solve_mbar_for_all_states(dm, N_k, np.zeros(24), sws, ms.BOOTSTRAP_SOLVER_PROTOCOL)
            fr, _ = oracle.solve_mbar_for_all_states(u_res, N_k, np.zeros(24), sws, tol=1e-12, min_sc_iter=0)
            np.testing.assert_allclose(fw, fr, rtol=1e-9, atol=1e-10)
            dm.set_sample_weights(None)  # back to the plain data
            ps1, _, _ = dm.eval(f)
            np.testing.assert_allclose(ps1[0], oracle.shard_partials(u_kn, N_k, f)["psum"], rtol=1e-11, atol=1e-10)


def test_mbar_bootstrap_on_gpu():
    """n_bootstraps through the MBAR class: deterministic under rseed (reference tests/test_mbar.py:533-545) and
    equal to solving each gathered replicate with the oracle."""
    import pymbar_amd

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0.0, 1.0, 2.0, 3.0], [1.0, 2.0, 3.0, 4.0], [300, 0, 250, 200], seed=4)
    a = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=4, rseed=11)
    b = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=4, rseed=11)
    assert np.array_equal(a.f_k_boots, b.f_k_boots)
    sws = np.where(N_k > 0)[0]
    for i in range(4):
        fr, _ = oracle.solve_mbar_for_all_states(u_kn[:, a.bootstrap_rints[i]], N_k, a.f_k.copy(), sws, tol=1e-12, min_sc_iter=0)
        np.testing.assert_allclose(a.f_k_boots[i], fr, rtol=1e-8, atol=1e-9)
    r = a.compute_free_energy_differences(uncertainty_method="bootstrap")
    assert r["dDelta_f"].shape == (4, 4) and np.all(np.isfinite(r["dDelta_f"]))
    a.close()
    b.close()


def test_bootstrap_replicates_drawn_on_the_device(DM):
    """mbar_ctx_draw_bootstrap_weights: the multiplicities the device draws are the draw counts of mbar_bootstrap_draws (the
    host face of the same counter-based stream) -- for the default layout, for an order vector, on a shard (n_global0) -- and the
    class with bootstrap_rng="device" solves those replicates (oracle on the gathered matrix), deterministic under rseed, also
    through the bootstrap branch of the expectations."""
    import pymbar_amd
    from pymbar_amd import _lib

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([0.0, 1.0, 2.0, 3.0], [1.0, 2.0, 3.0, 4.0], [3000, 0, 2500, 2001], seed=4)
    N, K = u_kn.shape[1], len(N_k)
    cum = np.concatenate(([0], np.cumsum(N_k))).astype(np.int64)
    f = np.array([0.0, 0.3, 0.5, 0.9])
    perm = np.random.default_rng(1).permutation(N)
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for order in (None, perm):
            for rep in (0, 1, 7):
                dm.draw_bootstrap_weights(12345, rep, cum, order)
                rints = _lib.bootstrap_draws(12345, rep, cum, order)
                ps, _, _ = dm.eval(f)
                dm.set_sample_weights(np.bincount(rints, minlength=N))
                ps_ref, _, _ = dm.eval(f)
                assert np.array_equal(ps[0], ps_ref[0]), (rep, order is None)   # the same integer multiplicities: the same bits
        dm.set_sample_weights(None)
    n0, n1 = 2048, 6016   # a shard of the same replicate
    with DM.from_host(u_kn, columns=(n0, n1)) as sh:
        sh.set_Nk(N_k)
        sh.draw_bootstrap_weights(12345, 3, cum, None, n_global0=n0)
        ps, _, _ = sh.eval(f)
        sh.set_sample_weights(np.bincount(_lib.bootstrap_draws(12345, 3, cum), minlength=N)[n0:n1])
        ps_ref, _, _ = sh.eval(f)
        assert np.array_equal(ps[0], ps_ref[0])
    a = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=4, rseed=11, bootstrap_rng="device")
    b = pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=4, rseed=11, bootstrap_rng="device")
    assert a._bootstrap_stream is not None and a._bootstrap_rints is None
    assert np.array_equal(a.f_k_boots, b.f_k_boots)
    A = np.cos(x_n) + 2.0
    ea = a.compute_expectations(A, uncertainty_method="bootstrap")     # (replicates drawn on the augmented matrix's context)
    eb = b.compute_expectations(A, uncertainty_method="bootstrap")
    assert np.array_equal(ea["sigma"], eb["sigma"]) and np.all(np.isfinite(ea["sigma"])) and np.all(ea["sigma"] > 0)
    assert a._bootstrap_rints is None
    sws = np.where(N_k > 0)[0]
    rints = a.bootstrap_rints
    for i in range(4):
        fr, _ = oracle.solve_mbar_for_all_states(u_kn[:, rints[i]], N_k, a.f_k.copy(), sws, tol=1e-12, min_sc_iter=0)
        np.testing.assert_allclose(a.f_k_boots[i], fr, rtol=1e-8, atol=1e-9)
    a.close()
    b.close()


def test_two_candidates_far_apart(DM):
    """The fused two-candidate sweep derives the second candidate from the first one's exponentials through
    exp(a'_k - a_k); candidates hundreds of kT apart take the two-sweep fallback.  Both must match the oracle."""
    u_kn, N_k, f = random_problem(128, 1500, seed=77)
    N_k = np.maximum(N_k, 1)
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for scale in (0.5, 40.0, 250.0, 700.0):
            f2 = f + scale * np.sin(np.arange(128))
            f2[0] = 0
            ps, sl, _ = dm.eval(np.stack([f, f2]))
            for i, ff in enumerate((f, f2)):
                part = oracle.shard_partials(u_kn, N_k, ff)
                np.testing.assert_allclose(ps[i], part["psum"], rtol=1e-10, atol=1e-9, err_msg=f"scale={scale} cand={i}")
                np.testing.assert_allclose(sl[i], part["sumlogden"], rtol=1e-12, err_msg=f"scale={scale} cand={i}")
            np.testing.assert_allclose(dm.logden(f2), oracle.log_denominator(u_kn, N_k, f2), rtol=1e-13, atol=1e-11)


def test_nan_and_infinite_energies(DM):
    """+inf energies are legal (zero weight, logsumexp semantics); a NaN or -inf entry poisons every sum, as it
    does in the reference where the per-state log-sum-exp runs over all samples."""
    u_kn, N_k, f = random_problem(10, 900, seed=31)
    u_inf = u_kn.copy()
    u_inf[3, ::7] = np.inf
    u_inf[0, 5] = np.inf
    with DM.from_host(u_inf) as dm:
        check_l1(dm, u_inf, N_k, f, tag="+inf entries")
    with np.errstate(all="ignore"):
        for bad in (np.nan, -np.inf):
            u_bad = u_kn.copy()
            u_bad[4, 123] = bad
            # (what scipy's logsumexp does with a NaN term depends on its version; here the whole matrix is poisoned)
            with DM.from_host(u_bad) as dm:
                assert not np.any(np.isfinite(ms.mbar_gradient(dm, N_k, f)))
                assert not np.any(np.isfinite(ms.self_consistent_update(dm, N_k, f)))
                assert not np.any(np.isfinite(ms.mbar_hessian(dm, N_k, f)))
                fa, res = ms.solve_mbar_once(dm, N_k, np.zeros(10), method="adaptive", options=dict(maxiter=5))
                assert not np.any(np.isfinite(fa[1:]))
        with DM.from_host(u_kn) as dm:
            f_nan = f.copy()
            f_nan[2] = np.nan
            assert not np.any(np.isfinite(ms.mbar_gradient(dm, N_k, f_nan)))
            np.testing.assert_allclose(ms.mbar_gradient(dm, N_k, f), oracle.mbar_gradient(u_kn, N_k, f), rtol=1e-9, atol=1e-9)


def test_full_panel_gram_with_and_without_the_exponential_clamp(DM):
    """The 128-state Gram panel runs its exponentials without the clamp when the matrix holds no +inf entry (finite
    stand-ins replace the -inf of unsampled / padded states and of padded or zero-weight samples) and with it
    otherwise.  Both instantiations against the oracle: ragged N, unsampled states, huge finite energies, +inf
    entries, per-sample multiplicities including zeros; the W-mode Gram (all states) as well."""
    K, N = 128, 3001
    u_kn, N_k, f = random_problem(K, N, seed=77, unsampled=(5, 64, 127))
    u_big = u_kn.copy()
    u_big[9, ::11] = 1e300          # "forbidden" configurations written as a huge finite energy
    u_big[17, 3::13] = 7.5e5        # beyond the int32 range of the table index (saturating convert)
    u_inf = u_big.copy()
    u_inf[40, ::5] = np.inf
    Nf = N_k.astype(float)
    for tag, u in (("finite (unclamped)", u_big), ("+inf (clamped)", u_inf)):
        with DM.from_host(u) as dm:
            check_l1(dm, u, N_k, f, tag=tag)
            G, wsum = dm.gram_w(f)
            W = oracle.mbar_W_nk(u, Nf, f)
            np.testing.assert_allclose(G, W.T @ W, rtol=1e-10, atol=1e-15, err_msg=tag + " gram_w")
            # multiplicities (with zeros) == the explicitly resampled matrix
            rng = np.random.default_rng(3)
            c_n = rng.integers(0, 4, size=N).astype(float)
            dm.set_sample_weights(c_n)
            try:
                ps, _, Gc = dm.eval(f, gram=True)
            finally:
                dm.set_sample_weights(None)
            ue = np.repeat(u, c_n.astype(int), axis=1)
            part = oracle.shard_partials(ue, N_k, f, want_gram=True)
            np.testing.assert_allclose(ps[0], part["psum"], rtol=1e-11, atol=1e-9, err_msg=tag + " weighted psum")
            np.testing.assert_allclose(Gc, part["gram"], rtol=1e-10, atol=1e-10, err_msg=tag + " weighted gram")
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-12, maxiter=100, min_sc_iter=0)
            assert ra["success"]
            g = oracle.mbar_gradient(u, Nf, fa)
            assert np.linalg.norm(g) < 1e-6 * N


@pytest.mark.parametrize("K,N,unsampled", [(5, 3000, ()), (40, 20000, (7, 23)), (64, 9000, ()), (128, 30011, (5,)),
                                           (200, 12000, (11,)), (256, 10000, ())])
def test_adaptive_loop_variants_agree(DM, K, N, unsampled):
    """The adaptive iteration in its four forms -- host-driven loop, device-resident loop with per-sweep exponentials,
    with the resident probability matrix (two sweeps), and with the fused sweep (speculated Gram matrix of the Newton
    candidate) -- and with either K x K Newton solve (blocked LDL^T on the matrix cores, register Gauss-Jordan) gives the same free energies, iteration counts and per-iteration gradient norms, also when the
    self-consistent candidate is forced for the first iterations (every speculation of those iterations is rejected),
    with per-sample multiplicities, with a damped Newton step, and for a fixed number of iterations past convergence."""
    u_kn, N_k, f = random_problem(K, N, seed=K + 1, unsampled=unsampled)
    tol = 1e-12 if K <= 128 else 1e-10  # (~50 samples per state above 128 states: 1e-12 is the round-off floor of f there)
    sws = np.where(N_k > 0)[0]
    modes = {"host": dict(device_loop=0), "classic": dict(device_loop=1, pmode=0, fused=0),
             "pmode": dict(device_loop=1, pmode=1, fused=0), "fused": dict(device_loop=1, pmode=1, fused=1),
             "fused eager": dict(device_loop=1, pmode=1, fused=1, graph=0),
             "fused graphs of 2": dict(device_loop=1, pmode=1, fused=1, adapt_batch=2),   # (every batch a replay, also the first)
             "fused, own Newton launch": dict(device_loop=1, pmode=1, fused=1, merge_select=0),
             # the K x K Newton solve of rounds 2-5 (register Gauss-Jordan) instead of the blocked LDL^T on the matrix cores
             "fused, Gauss-Jordan Newton solve": dict(device_loop=1, pmode=1, fused=1, newton_ldlt=0),
             "classic, Gauss-Jordan, own Newton launch": dict(device_loop=1, pmode=0, fused=0, merge_select=0, newton_ldlt=0),
             # the last iteration without its Gram matrix whatever the size (default: from 1e8 matrix entries on)
             "fused, light last sweep": dict(device_loop=1, pmode=1, fused=1, light_last=2),
             "fused, light last sweep, eager": dict(device_loop=1, pmode=1, fused=1, light_last=2, graph=0, merge_select=0)}
    light_seen = 0
    rng = np.random.default_rng(K)
    # a bootstrap replicate: draw counts of a resampling WITHIN each state (sum_n c_n over a state's samples = N_k, or the
    # weighted equations have no solution)
    c_n = np.zeros(u_kn.shape[1])
    start = 0
    for n_k in N_k:
        if n_k > 0:
            c_n[start:start + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
        start += n_k
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for case in (dict(min_sc_iter=0), dict(min_sc_iter=3), dict(min_sc_iter=0, gamma=0.5), dict(min_sc_iter=0, weights=True),
                     dict(min_sc_iter=0, fixed=12),
                     dict(min_sc_iter=0, fixed=30)):  # (30: batches of 8, 8, 8, 6 between two looks at the control words -- the full ones replay the captured hipGraph)
            out = {}
            for name, opts in modes.items():
                for k, v in {"graph": 1, "adapt_batch": 8, "merge_select": 1, "light_last": 0, "newton_ldlt": 1, **opts}.items():
                    dm.set_option(k, v)
                dm.set_sample_weights(c_n if case.get("weights") else None)
                try:
                    fa, ra = dm.solve_adaptive(np.zeros(K), tol=tol, maxiter=case.get("fixed", 200), min_sc_iter=case["min_sc_iter"],
                                               gamma=case.get("gamma", 1.0), check_convergence="fixed" not in case, history_rows=200)
                finally:
                    dm.set_sample_weights(None)
                out[name] = (fa, ra)
            f_ref, r_ref = out["host"]
            assert r_ref["success"] or "fixed" in case
            for name, (fa, ra) in out.items():
                np.testing.assert_allclose(fa[sws], f_ref[sws], rtol=1e-11, atol=1e-11, err_msg=f"{case} {name}")
                assert ra["iterations"] == r_ref["iterations"], (case, name, ra["iterations"], r_ref["iterations"])
                assert ra["success"] == r_ref["success"]
                big = r_ref["history"][:, 1:3] > 1e-6
                np.testing.assert_allclose(ra["history"][:, 1:3][big], r_ref["history"][:, 1:3][big], rtol=1e-6, err_msg=f"{case} {name}")
                if name.startswith("fused") and K <= 128:  # (above 128 states the device loop runs the two-sweep form)
                    # a separate Gram sweep runs only when the accepted candidate is not the speculated one: the fused sweep
                    # speculates on the self-consistent candidate while those steps are forced (sci_iter < min_sc_iter), on the
                    # Newton-Raphson one otherwise -- so only a self-consistent step that WON on its gradient norm costs a sweep
                    forced = np.cumsum(ra["history"][:, 0] == 0) <= case["min_sc_iter"]
                    lost = int(np.sum((ra["history"][:, 0] == 0) & ~forced))
                    assert ra["gram_sweeps"] <= lost, (case, name, ra["gram_sweeps"], lost)
                    # the plain sweep stands in for the fused one in the LAST iteration only (both candidates already met the stop
                    # test), never when the iteration count is fixed, never unless asked for at these sizes
                    assert ra["light_sweeps"] <= (1 if "light" in name and "fixed" not in case else 0), (case, name, ra["light_sweeps"])
                    light_seen += ra["light_sweeps"]
            if not case.get("weights") and "fixed" not in case:
                f_or, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=tol, min_sc_iter=case["min_sc_iter"])
                if case.get("gamma", 1.0) == 1.0:
                    np.testing.assert_allclose(out["fused"][0][sws], (f_or - f_or[sws[0]])[sws], rtol=1e-9, atol=1e-10)
        for k, v in dict(device_loop=1, pmode=1, fused=1, graph=1, adapt_batch=8, merge_select=1, light_last=1, newton_ldlt=1).items():
            dm.set_option(k, v)
    if K == 128:  # (it does happen -- where the grids of the two sweeps are commensurate: the converging cases end on an iteration
        assert light_seen >= 2, light_seen  # both candidates pass)


# (300 / 380 / 440 / 600 / 800 states: the 256-state panels of the P-mode plan + a remainder of 64 / 128 / 192 / 2 x 256 + 128 / 4 x 256 rows;
# 1100: beyond the row-split evaluation kernel -- Gram sweep on the probability matrix, evaluation by the layout-agnostic pair on u)
@pytest.mark.parametrize("K,N", [(160, 6400), (256, 7680), (300, 6000), (380, 5700), (440, 6600), (600, 6000), (800, 8000), (1100, 5500)])
def test_adaptive_solves_above_128_states_match_the_oracle(DM, K, N):
    """Adaptive solves beyond one Gram panel (129-256 states: paneled Gram sweeps; 257-512: row-split evaluation sweep; above:
    layout-agnostic sweeps) against the oracle's loop (mbar_solvers.py:575-640): free energies, iteration counts, the choice
    and both gradient norms of EVERY iteration -- with an unsampled state, with forced self-consistent steps, and for a
    bootstrap replicate (draw counts on the resident matrix against the oracle on the explicitly gathered columns).
    Tolerance 1e-10: with ~20 samples per state the last step of a solve to 1e-12 changes f by 5e-13 ... 1.5e-12 in the
    reference itself (round-off of the gradient times the condition number of the Hessian), so whether iteration 5 or 6 passes
    the test of :636 is noise there; at 1e-10 the count is pinned with two orders of magnitude to spare on either side."""
    tol = 1e-10
    u_kn, N_k, _ = random_problem(K, N, seed=K + 7, unsampled=(K // 3,))
    sws = np.where(N_k > 0)[0]
    Nf = N_k[sws].astype(float)
    rng = np.random.default_rng(K)
    rints = np.zeros(N, dtype=np.int64)
    start = 0
    for n_k in N_k:
        if n_k > 0:
            rints[start:start + n_k] = start + rng.integers(0, n_k, size=n_k)
        start += n_k
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        cases = [dict(min_sc_iter=0), dict(min_sc_iter=2), dict(min_sc_iter=0, boot=True)]
        if K > 256:  # the host-driven loop's sweeps on a resident probability matrix (default: 256-state panels; 1: 128-state) and on u
            cases += [dict(min_sc_iter=0, host_pmode=1), dict(min_sc_iter=0, boot=True, host_pmode=1)]
            cases += [dict(min_sc_iter=0, host_pmode=0), dict(min_sc_iter=0, boot=True, host_pmode=0)]
        for case in cases:
            dm.set_option("host_pmode", case.get("host_pmode", 2))
            u_or = u_kn[:, rints] if case.get("boot") else u_kn
            hist = []
            r_or = oracle.adaptive(np.ascontiguousarray(u_or[sws]), Nf, np.zeros(len(sws)), tol=tol, min_sc_iter=case["min_sc_iter"],
                                   history=hist)
            assert r_or["success"]
            dm.set_sample_weights(np.bincount(rints, minlength=N) if case.get("boot") else None)
            try:
                fa, ra = dm.solve_adaptive(np.zeros(K), tol=tol, maxiter=300, min_sc_iter=case["min_sc_iter"], history_rows=300)
            finally:
                dm.set_sample_weights(None)
            assert ra["success"] and ra["iterations"] == r_or["iterations"], (case, ra["iterations"], r_or["iterations"])
            np.testing.assert_allclose(fa[sws], r_or["x"], rtol=1e-9, atol=1e-9, err_msg=str(case))
            gn = np.array([[h["gnorm_sci"], h["gnorm_nr"]] for h in hist])
            np.testing.assert_allclose(ra["history"][:, 1:3], gn, rtol=1e-9, atol=1e-8, err_msg=str(case))
            # (the choice between two gradient norms at round-off level is noise: compared while they are not)
            clear = np.abs(gn[:, 0] - gn[:, 1]) > 1e-7
            ch_or = np.array([0.0 if h["choice"] == "sci" else 1.0 for h in hist])
            forced = np.arange(len(hist)) < case["min_sc_iter"]
            assert np.array_equal(ra["history"][clear | forced, 0], ch_or[clear | forced]), case


@pytest.mark.parametrize("K", [40, 200])
def test_disconnected_states_take_the_pseudo_inverse_branch(DM, K):
    """Two groups of states with NO overlap (+inf energies on each other's samples): the gauge-fixed Newton system is exactly
    singular, the Cholesky / Gauss-Jordan elimination breaks down and the minimum-norm branch (numpy.linalg.lstsq semantics,
    mbar_solvers.py:582) takes over -- on the host, also for more unknowns than the device solve holds.  The free energy
    between the groups is undetermined (the reference itself wanders there); WITHIN each group the answer is that of the
    group solved alone, and the gradient vanishes."""
    N = 40 * K
    u_kn, N_k, _ = random_problem(K, N, seed=K + 11)
    s_n = np.repeat(np.arange(K), N_k)
    half = K // 2
    u = u_kn.copy()
    u[half:, s_n < half] = np.inf
    u[:half, s_n >= half] = np.inf
    with DM.from_host(u) as dm:
        dm.set_Nk(N_k)
        with np.errstate(all="ignore"):
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-12, maxiter=60, min_sc_iter=0)
        assert np.all(np.isfinite(fa))
        psum, _, _ = dm.eval(fa)
        assert np.linalg.norm(psum[0] - N_k) < 1e-7 * N
    for lo, hi in ((0, half), (half, K)):
        cols = (s_n >= lo) & (s_n < hi)
        r_or = oracle.adaptive(np.ascontiguousarray(u_kn[lo:hi][:, cols]), N_k[lo:hi].astype(float), np.zeros(hi - lo), tol=1e-12,
                               min_sc_iter=0)
        assert r_or["success"]
        np.testing.assert_allclose(fa[lo:hi] - fa[lo], r_or["x"], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("K,step", [(40, 30.0), (128, 9.0)])
def test_device_loop_reanchors_when_a_step_leaves_its_window(DM, K, step):
    """Free energies spread over ~1200 kT with a start at f = 0: the first candidates lie far outside the 250 kT window
    around the anchor of the resident probability matrix, so k_newton hands the solve back, the host runs one classic
    iteration and the device loop re-anchors -- possibly several times.  Same answer and iteration count as the
    host-driven loop and as the oracle."""
    u_kn, N_k, f = random_problem(K, 6000, seed=5)
    u_kn = u_kn + step * np.arange(K)[:, None]          # f_k shifts by step * k
    sws = np.arange(K)
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        out = {}
        for name, opts in (("host", dict(device_loop=0)), ("fused", dict(device_loop=1, pmode=1, fused=1)),
                           ("pmode", dict(device_loop=1, pmode=1, fused=0)), ("classic", dict(device_loop=1, pmode=0, fused=0))):
            for k, v in opts.items():
                dm.set_option(k, v)
            out[name] = dm.solve_adaptive(np.zeros(K), tol=1e-12, maxiter=300, min_sc_iter=0)
        f_ref, r_ref = out["host"]
        assert r_ref["success"]
        assert abs(f_ref[-1] - step * (K - 1)) < 10.0
        for name, (fa, ra) in out.items():
            assert ra["success"] and ra["iterations"] == r_ref["iterations"], (name, ra["iterations"], r_ref["iterations"])
            np.testing.assert_allclose(fa, f_ref, rtol=1e-12, atol=1e-9, err_msg=name)
        f_or, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=1e-12, min_sc_iter=0)
        np.testing.assert_allclose(out["fused"][0], f_or - f_or[0], rtol=1e-10, atol=1e-8)
        for k, v in dict(device_loop=1, pmode=1, fused=1).items():
            dm.set_option(k, v)


def test_objective_offset_matches_preconditioned_objective(DM):
    u_kn, N_k, f = random_problem(12, 1500, seed=21)
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        dm.set_objective_offset(f)
        f2 = f + 0.01 * np.sin(np.arange(12))
        _, sld, _ = dm.eval(f2, use_offset=True)
        pre = oracle.precondition_u_kn(u_kn, N_k, f)
        want = oracle.mbar_objective(pre, N_k, f2)
        got = sld[0] + np.dot(N_k, f) - np.dot(N_k, f2)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
        dm.set_objective_offset(None)
        np.testing.assert_allclose(ms.precondition_u_kn(dm, N_k, f), pre, rtol=1e-12, atol=1e-10)


# ---- golden fixtures produced by the reference -------------------------------------------------
def test_golden_config1_l1_and_adaptive(DM, golden):
    g = golden("config1_ho_K5_N5000.npz")
    u_kn, N_k, f = g["u_kn"], g["N_k"], g["f_eval"]
    with DM.from_host(u_kn) as dm:
        np.testing.assert_allclose(ms.mbar_gradient(dm, N_k, f), g["gradient"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, f), g["sci"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ms.mbar_objective(dm, N_k, f), g["objective"], rtol=1e-12)
        np.testing.assert_allclose(ms.mbar_hessian(dm, N_k, f), g["hessian"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(np.exp(ms.mbar_log_W_nk(dm, N_k, f)).sum(0), g["logW_colsum"], rtol=1e-11)
        fa, res = ms.solve_mbar_once(dm, N_k, np.zeros(5), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        assert res["success"] and res["iterations"] == int(g["adaptive_iters"])
        np.testing.assert_allclose(fa, g["f_adaptive"], rtol=1e-10, atol=1e-11)
        # reference test_mbar_solvers.py:34-41 properties at the solution
        np.testing.assert_allclose(ms.mbar_gradient(dm, N_k, fa), 0, atol=1e-8)
        np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, fa), fa, atol=1e-10)
        W = ms.mbar_W_nk(dm, N_k, fa)
        np.testing.assert_allclose(W.sum(0), 1, atol=1e-10)
        np.testing.assert_allclose(W @ N_k, 1, atol=1e-10)


@pytest.mark.parametrize("method", ["adaptive", "hybr", "lm", "L-BFGS-B", "BFGS", "Newton-CG", "trust-ncg", "dogleg",
                                    "trust-exact", "trust-krylov", "self-consistent-iteration"])
def test_golden_config1_every_method(DM, golden, method):
    g = golden("config1_ho_K5_N5000.npz")
    with DM.from_host(g["u_kn"]) as dm:
        f, res = ms.solve_mbar_once(dm, g["N_k"], np.zeros(5), method=method, tol=1e-12, options=dict())
    np.testing.assert_allclose(f, g["f_k"], atol=1e-7)


def test_golden_mbar_class_config1_and_unsampled(golden):
    import pymbar_amd

    g = golden("config1_ho_K5_N5000.npz")
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"])
    np.testing.assert_allclose(mbar.f_k, g["f_k"], atol=1e-10)
    for method, tag in (("svd-ew", "svd_ew"), ("svd", "svd"), ("approximate", "approximate")):
        r = mbar.compute_free_energy_differences(uncertainty_method=method, return_theta=True)
        np.testing.assert_allclose(r["Delta_f"], g["Delta_f"], atol=1e-10)
        np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_" + tag], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(r["Theta"], g["Theta_" + tag], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mbar.compute_overlap()["matrix"], g["overlap_matrix"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mbar.compute_effective_sample_number(), g["N_eff"], rtol=1e-8)
    mbar.close()
    g = golden("ho_unsampled_K4_N2300.npz")
    for proto in (None, "robust"):
        mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"], solver_protocol=proto)
        np.testing.assert_allclose(mbar.f_k, g["f_k"], rtol=1e-8, atol=1e-9)
        r = mbar.compute_free_energy_differences()
        np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_svd_ew"], rtol=1e-7, atol=1e-9)
        mbar.close()


def test_golden_oscillators_and_ladder(DM, golden):
    g = golden("osc_K50_N5000.npz")
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(1, 5, 50), np.linspace(1, 3, 50), [100] * 50, seed=7)
    with DM.from_host(u_kn) as dm:
        np.testing.assert_allclose(ms.mbar_gradient(dm, N_k, g["f_eval"]), g["gradient"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ms.mbar_hessian(dm, N_k, g["f_eval"]), g["hessian"], rtol=1e-9, atol=1e-9)
        f, res = ms.solve_mbar_once(dm, N_k, np.zeros(50), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        assert res["iterations"] == int(g["adaptive_iters"])
        np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-9, atol=1e-10)
    g = golden("ladder_K32_N32000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config2(seed=0, K=32, N=32000)
    with DM.from_host(u_kn) as dm:
        f, res = ms.solve_mbar_once(dm, N_k, np.zeros(32), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        assert res["iterations"] == int(g["adaptive_iters"]) and abs(res["nr_iter"] - int(g["adaptive_nr"])) <= 1
        np.testing.assert_allclose(f, g["f_adaptive"], rtol=1e-9, atol=1e-10)
        f2, res2 = ms.solve_mbar_once(dm, N_k, np.zeros(32), method="adaptive", tol=1e-12, options=dict())
        assert res2["iterations"] == int(g["adaptive_iters_msc2"])
        fs, rs = ms.solve_mbar_once(dm, N_k, np.zeros(32), method="self-consistent-iteration", tol=1e-12)
        assert rs["success"] and abs(rs["iterations"] - int(g["sci_iters"])) <= 1
        np.testing.assert_allclose(fs, g["f_sci"], rtol=1e-9, atol=1e-10)


def test_golden_config5_alchemical_shape(golden):
    """BASELINE.json config 5: K=40, N=95000, two unsampled states; Delta_f within 1e-8 relative of the
    reference, dDelta_f (svd-ew covariance) likewise."""
    import pymbar_amd

    g = golden("config5_alch_K40_N95000.npz")
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    mbar = pymbar_amd.MBAR(u_kn, N_k)
    r = mbar.compute_free_energy_differences()
    rel = np.abs(r["Delta_f"] - g["Delta_f"]) / np.maximum(np.abs(g["Delta_f"]), 1e-3)
    assert rel.max() < 1e-8, rel.max()
    np.testing.assert_allclose(r["dDelta_f"], g["dDelta_f_svd_ew"], rtol=1e-7, atol=1e-10)
    mbar2 = pymbar_amd.MBAR(u_kn, N_k, solver_protocol=(dict(method="adaptive", options=dict(min_sc_iter=0)),))
    np.testing.assert_allclose(mbar2.f_k, g["f_k_adaptive_protocol"], rtol=1e-8, atol=1e-9)
    mbar.close()
    mbar2.close()


# ---- device generator, sharding, full size ---------------------------------------------------------
def test_device_generator_is_shard_invariant_and_matches_oracle(DM):
    O_k, K_k, N_k = ts.config3_params(K=32, N=64000)
    with DM.harmonic(O_k, K_k, N_k, seed=3) as whole:
        u = whole.to_host()
        assert np.all(np.isfinite(u)) and u.min() >= 0
        n_half = 32000 + 16
        with DM.harmonic(O_k, K_k, N_k, seed=3, n_global0=0, N_local=n_half) as a, \
                DM.harmonic(O_k, K_k, N_k, seed=3, n_global0=n_half, N_local=64000 - n_half) as b:
            np.testing.assert_array_equal(np.concatenate([a.to_host(), b.to_host()], axis=1), u)
            f = ts.harmonic_free_energies(K_k) + 0.01
            f[0] = 0
            for dm in (whole, a, b):
                dm.set_Nk(N_k)
            pw, sw, gw = whole.eval(f, gram=True)
            pa, sa, ga = a.eval(f, gram=True)
            pb, sb, gb = b.eval(f, gram=True)
            np.testing.assert_allclose(pa + pb, pw, rtol=1e-12)
            np.testing.assert_allclose(ga + gb, gw, rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(sa + sb, sw, rtol=1e-13)
        # statistics of the generated ladder: MBAR on it recovers the analytical free energies
        f_sol, res = ms.solve_mbar_once(whole, N_k, np.zeros(32), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        assert res["success"]
        assert np.max(np.abs(f_sol - ts.harmonic_free_energies(K_k))) < 0.05
        np.testing.assert_allclose(f_sol, oracle.solve_mbar_once_adaptive(u, N_k, np.zeros(32), min_sc_iter=0)[0],
                                   rtol=1e-9, atol=1e-10)


def test_config2_sci_on_device(DM):
    """BASELINE.json config 2: K=32, N=1e6, pure self-consistent iteration, checked against a short oracle run
    on the downloaded matrix (same bits) and against the fixed-point property."""
    O_k, K_k, N_k = ts.config3_params(K=32, N=1_000_000)
    with DM.harmonic(O_k, K_k, N_k, seed=0) as dm:
        f, res = ms.solve_mbar_once(dm, N_k, np.zeros(32), method="self-consistent-iteration", tol=1e-12)
        assert res["success"] and 50 < res["iterations"] < 200
        np.testing.assert_allclose(ms.self_consistent_update(dm, N_k, f) - ms.self_consistent_update(dm, N_k, f)[0], f,
                                   atol=1e-10)
        fa, ra = ms.solve_mbar_once(dm, N_k, np.zeros(32), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        np.testing.assert_allclose(f, fa, atol=1e-9)
        assert np.max(np.abs(fa - ts.harmonic_free_energies(K_k))) < 0.02
        u_sub = dm.to_host()[:, :50000]
    # oracle on the first 50k columns vs a device context holding the same columns
    Nk_sub = np.bincount(np.searchsorted(np.cumsum(N_k), np.arange(50000), side="right"), minlength=32)
    with DM.from_host(u_sub) as sub:
        check_l1(sub, u_sub, Nk_sub + 1, f, tag="config2 sub-block")


def test_config3_full_size_properties(DM):
    """BASELINE.json config 3 (K=128, N=1e7) at full size: size-independent properties of the solution."""
    K, N = 128, 10_000_000
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    with DM.harmonic(O_k, K_k, N_k, seed=0) as dm:
        f, res = ms.solve_mbar_once(dm, N_k, np.zeros(K), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
        assert res["success"] and res["iterations"] <= 12
        dm.set_Nk(N_k)
        psum, sld, gram = dm.eval(f, gram=True)
        g = psum[0] - N_k
        H = np.diag(psum[0]) - gram
        assert abs(psum[0].sum() - N) < 1e-6 * N ** 0.5          # sum_k p_nk = 1 for every sample
        assert np.max(np.abs(g)) < 1e-6                           # gradient vanishes at the solution
        assert np.max(np.abs(H @ np.ones(K))) < 1e-6              # null vector 1 (H rows sum to 0)
        assert np.array_equal(gram, gram.T)
        ev = np.linalg.eigvalsh(H)
        assert ev[0] > -1e-6 and ev[1] > 0                        # PSD with a single zero mode
        fs = ms.self_consistent_update(dm, N_k, f)
        np.testing.assert_allclose(fs - fs[0], f, atol=1e-10)     # SCI fixed point (test_mbar_solvers.py:39-41)
        assert np.max(np.abs(f - ts.harmonic_free_energies(K_k))) < 0.02
    # oracle on a 100k-column block of the same bits (a full download is 10 GB; regenerate the first block
    # through the shard-invariant generator instead)
    with DM.harmonic(O_k, K_k, N_k, seed=0, n_global0=0, N_local=100_000) as sub:
        u_sub = sub.to_host()
        sub.set_Nk(N_k)
        ps, sl, gs = sub.eval(f, gram=True)
        part = oracle.shard_partials(u_sub, N_k, f, want_gram=True)
        np.testing.assert_allclose(ps[0], part["psum"], rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(gs, part["gram"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(sl[0], part["sumlogden"], rtol=1e-13)


@pytest.mark.parametrize("K", [32, 128])
def test_wide_row_pitch_matches_sum_of_shards(DM, K):
    """From 7.6e7 samples per rank the tile DMA needs 64-bit lane offsets (separate kernel instantiations).  The
    whole matrix must give the sums of its two column shards (each below the threshold, i.e. on the 32-bit kernels),
    and the solver must run on it.  K = 128 at 7.8e7 samples is a 80 GB matrix: the largest case in the suite."""
    from pymbar_amd.device import device_info

    N = 78_000_000
    if device_info()["total_mem_bytes"] < (8 * K * N * 3) // 2 + (8 << 30):
        pytest.skip("not enough device memory for the wide-pitch case")
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    f = ts.harmonic_free_energies(K_k) + 0.01 * np.cos(np.arange(K))
    f[0] = 0
    f2 = np.stack([f, f + 0.003 * np.sin(np.arange(K))])
    with DM.harmonic(O_k, K_k, N_k, seed=5) as whole:
        whole.set_Nk(N_k)
        pw, sw, gw = whole.eval(f2, gram=True)
        lnw = whole.lognum(f)
        fs, res = ms.solve_mbar_once(whole, N_k, f.copy(), method="adaptive", tol=1e-10, options=dict(min_sc_iter=0, maxiter=3))
        assert np.all(np.isfinite(fs))
    parts = []
    n_half = N // 2 + 16
    for n0, nloc in ((0, n_half), (n_half, N - n_half)):
        with DM.harmonic(O_k, K_k, N_k, seed=5, n_global0=n0, N_local=nloc) as sh:
            sh.set_Nk(N_k)
            parts.append(sh.eval(f2, gram=True) + (sh.lognum(f),))
    np.testing.assert_allclose(parts[0][0] + parts[1][0], pw, rtol=1e-12)
    np.testing.assert_allclose(parts[0][1] + parts[1][1], sw, rtol=1e-13)
    np.testing.assert_allclose(parts[0][2] + parts[1][2], gw, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(np.logaddexp(parts[0][3], parts[1][3]), lnw, rtol=1e-12, atol=1e-12)
    assert abs(pw[0].sum() - N) < 1e-6 * N ** 0.5


@pytest.mark.parametrize("K,N", [(128, 300_000), (256, 100_000), (32, 500_000)])
def test_reduced_outputs_are_bit_reproducible(DM, K, N):
    """Fixed-order reductions, no atomics: every launch returns the same bits (tools/soak_determinism.py runs the long
    version).  Also the canary for matrix-core hazards around the hand-pinned MFMA blocks: they show up as noise."""
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    with DM.harmonic(O_k, K_k, N_k, seed=1) as dm:
        dm.set_Nk(N_k)
        rng = np.random.default_rng(K)
        f = ts.harmonic_free_energies(K_k) + 0.05 * rng.normal(size=K)
        f[0] = 0.0
        f2 = np.stack([f, f + 0.01 * rng.normal(size=K)])
        first = None
        for _ in range(25):
            psum, sld, G = dm.eval(f2, gram=True)
            cur = (psum.tobytes(), sld.tobytes(), G.tobytes(), dm.lognum(f).tobytes())
            first = first or cur
            assert cur == first
        # solves: the first one builds the resident probability matrix, the later ones start warm on it (one fused sweep instead
        # of the build: other bits than a cold start, the same bits among themselves); with the cache off every solve is cold
        a, ra = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        b, rb = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        c, rc = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        assert np.array_equal(b, c) and ra["iterations"] == rb["iterations"] == rc["iterations"]
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
        dm.set_option("pcache", 0)
        d, rd = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        e, re_ = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        assert np.array_equal(a, d) and np.array_equal(d, e)
        if K <= 128:  # (above 128 states the loop has no resident probability matrix)
            assert rd["builds"] == 1 == re_["builds"] and rb["warm_starts"] == 1


def test_rccl_communicator_single_rank(DM):
    """RCCL path end to end on one GPU: dlopen librccl, ncclGetUniqueId, ncclCommInitRank(nranks=1), and every
    reduced output going through ncclAllReduce on the compute stream.  (More ranks need more GPUs; the decomposition
    itself is covered on CPU by tests/test_distributed_gloo.py.)"""
    import ctypes as C

    from pymbar_amd import _lib

    u_kn, N_k, f = random_problem(40, 5000, seed=9)
    with DM.from_host(u_kn) as plain, DM.from_host(u_kn) as comm:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load_library().mbar_comm_unique_id(buf))
        comm.comm_init_rccl(bytes(buf.raw), 0, 1)
        assert comm.allreduce_kind == "rccl"
        for dm in (plain, comm):
            dm.set_Nk(N_k)
        pa, sa, ga = plain.eval(f, gram=True)
        pb, sb, gb = comm.eval(f, gram=True)
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ga, gb)
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(plain.lognum(f), comm.lognum(f))
        fa, ra = plain.solve_adaptive(np.zeros(40), min_sc_iter=0)
        fb, rb = comm.solve_adaptive(np.zeros(40), min_sc_iter=0)
        assert ra["iterations"] == rb["iterations"] and rb["success"]
        np.testing.assert_array_equal(fa, fb)
        fs, rs = comm.solve_sci(np.zeros(40))
        np.testing.assert_allclose(fs, fa, atol=1e-9)
        # the host-callback transport gives the same numbers
        with DM.from_host(u_kn) as host:
            host.set_host_allreduce(lambda arr, op: None, 0, 1)
            host.set_Nk(N_k)
            ph, sh, gh = host.eval(f, gram=True)
            np.testing.assert_array_equal(pa, ph)


def test_mfma_peak_probe_is_sane(DM):
    with DM.from_host(np.zeros((4, 64))) as dm:
        t = dm.mfma_f64_peak()
    assert 60.0 < t < 90.0, t  # 64 cycles per instruction per SIMD at ~2.4 GHz = 78.6 TFLOP/s


def test_resident_probability_matrix_is_reused_across_solves(DM):
    """The resident probability matrix outlives the solve that built it: a second solve on the same matrix from a start within
    the window of its anchor begins with ONE fused sweep (no build) -- a bootstrap replicate, the next protocol stage, a
    restart -- and gives the result of a cold solve; anything that changes the matrix or the sampled-state set, and a start
    outside the window, build again."""
    K, N = 48, 30000
    u_kn, N_k, f = random_problem(K, N, seed=77, unsampled=(9,))
    sws = np.where(N_k > 0)[0]
    rng = np.random.default_rng(3)
    c_n = np.zeros(N)
    start = 0
    for n_k in N_k:
        if n_k > 0:
            c_n[start:start + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
        start += n_k
    with DM.from_host(u_kn) as dm, DM.from_host(u_kn) as cold:
        cold.set_option("pcache", 0)
        for d in (dm, cold):
            d.set_Nk(N_k)
        f1, r1 = dm.solve_adaptive(np.zeros(K), min_sc_iter=0)
        assert r1["builds"] == 1 and r1["warm_starts"] == 0
        f_start = f1 + 0.3 * np.cos(np.arange(K))
        f_start[sws[0]] = 0.0
        for fs, weights, msc in ((np.zeros(K), None, 0), (f_start, None, 2), (f1, c_n, 0), (f_start, c_n, 0)):
            for d in (dm, cold):
                d.set_sample_weights(weights)
            fa, ra = dm.solve_adaptive(fs, min_sc_iter=msc, history_rows=50)
            fb, rb = cold.solve_adaptive(fs, min_sc_iter=msc, history_rows=50)
            assert ra["warm_starts"] == 1 and ra["builds"] == 0 and rb["builds"] == 1 and rb["warm_starts"] == 0
            assert ra["success"] and ra["iterations"] == rb["iterations"]
            np.testing.assert_allclose(fa[sws], fb[sws], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(ra["history"][:, 1:3], rb["history"][:, 1:3], rtol=1e-7, atol=1e-9)
        for d in (dm, cold):
            d.set_sample_weights(None)
        # a start 300 kT away from the anchor: rebuilt there
        far = f1.copy()
        far[sws[1:]] += 300.0
        _, rf = dm.solve_adaptive(far, min_sc_iter=0, maxiter=3, check_convergence=False)
        assert rf["builds"] >= 1 and rf["warm_starts"] == 0
        # a changed matrix, and a changed set of sampled states, invalidate it
        dm.upload_rows(3, u_kn[3] + 0.5)
        _, rm = dm.solve_adaptive(f1, min_sc_iter=0)
        assert rm["builds"] == 1 and rm["warm_starts"] == 0
        N2 = N_k.copy()
        N2[9], N2[0] = 7, N2[0] - 7
        dm.set_Nk(N2)
        _, rn = dm.solve_adaptive(np.zeros(K), min_sc_iter=0, maxiter=4, check_convergence=False)
        assert rn["builds"] == 1 and rn["warm_starts"] == 0


@pytest.mark.parametrize("K,N,unsampled", [(5, 3000, ()), (48, 20000, (9,)), (128, 30000, ()), (200, 16000, (11,)), (300, 12000, ())])
def test_solve_hands_back_the_per_state_sums_at_its_result(DM, K, N, unsampled):
    """mbar_ctx_last_solve_psum: the sums the adaptive loop ends with are those of the f it returns (every loop form: the fused
    device-resident one, the split one above 128 states, the host-driven one above 256; cold, warm, weighted) -- what the host
    side uses for the gradient norm (mbar_solvers.py:939) and the final all-state update (:1012) instead of another sweep."""
    u_kn, N_k, _ = random_problem(K, N, seed=K + 1, unsampled=unsampled)
    rng = np.random.default_rng(K)
    c_n, first = np.zeros(N), 0
    for n_k in N_k:  # draw counts of one bootstrap replicate (they sum to N_k within each state's block)
        if n_k > 0:
            c_n[first:first + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
        first += n_k
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for weights, start in ((None, None), (None, "warm"), (c_n, "warm")):
            dm.set_sample_weights(weights)
            f0 = np.zeros(K) if start is None else fa + 0.1 * np.sin(np.arange(K)) * (N_k > 0)  # noqa: F821
            fa, ra = dm.solve_adaptive(f0, tol=1e-10, min_sc_iter=0)
            assert ra["success"] and ra["psum"] is not None
            psum = dm.eval(fa)[0][0]
            np.testing.assert_allclose(ra["psum"], psum, rtol=1e-11, atol=1e-11 * float(N_k.max()))
        dm.set_sample_weights(None)
        # nothing stale: a change of the matrix withdraws them
        dm.upload_rows(0, u_kn[0] + 0.25)
        from pymbar_amd.device import _dptr

        assert dm._lib.mbar_ctx_last_solve_psum(dm._ctx, _dptr(np.empty(K))) != 0


@pytest.mark.parametrize("K,N,unsampled", [(129, 3000, ()), (144, 5000, (3,)), (160, 7001, ()), (161, 2000, ()), (200, 6000, (11,)), (224, 9000, ()),
                                           (225, 3000, ())])
def test_padding_blocks_left_out_of_the_one_read_sweeps(DM, K, N, unsampled):
    """Up to 160 (224) states the last two 16-state blocks of the 192- (256-) row panel are padding: k_gram_quad / k_fused_quad
    neither stage nor multiply them ("quad_trim").  Every block is still accumulated by ONE wave over the same tile sequence,
    so the Gram matrices agree bit for bit with the whole-panel kernels; so do the solves."""
    u_kn, N_k, f = random_problem(K, N, seed=K, unsampled=unsampled)
    sws = np.where(N_k > 0)[0]
    out = []
    for trim in (0, 1):
        with DM.from_host(u_kn) as dm:
            dm.set_option("quad_trim", trim)
            dm.set_Nk(N_k)
            check_l1(dm, u_kn, N_k, f, tag=f"K={K} trim={trim}")
            ev = dm.eval(f, gram=True)
            gw = dm.gram_w(f)
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-10, min_sc_iter=0, history_rows=100)
            f2, r2 = dm.solve_adaptive(fa + 0.05 * np.sin(np.arange(K)) * (N_k > 0), tol=1e-10, min_sc_iter=2)  # warm, on the resident P
            out.append((ev, gw, fa, ra, f2, r2))
    (e0, g0, f0, r0, w0, rw0), (e1, g1, f1, r1, w1, rw1) = out
    assert np.array_equal(e0[2], e1[2]) and np.array_equal(e0[0], e1[0]) and np.array_equal(g0[0], g1[0])
    assert r0["success"] and r1["success"] and r0["iterations"] == r1["iterations"] and rw0["iterations"] == rw1["iterations"]
    assert np.array_equal(r0["history"], r1["history"]) and np.array_equal(f0, f1) and np.array_equal(w0, w1)
    f_or, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=1e-10, min_sc_iter=0)
    np.testing.assert_allclose(f1[sws] - f1[sws[0]], (f_or - f_or[sws[0]])[sws], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("K,N,unsampled", [(5, 20000, ()), (8, 9000, (2,)), (20, 30000, ()), (24, 30000, (0, 23)), (48, 20000, tuple(range(8, 16)) + tuple(range(40, 48))),
                                           (40, 50000, (39,)), (100, 20000, tuple(range(96, 100))), (32, 20000, tuple(range(16, 32)))])
def test_rows_that_cannot_contribute_are_not_streamed(DM, K, N, unsampled):
    """Aligned groups of 8 rows (2 in the few-state kernel) whose exponent constants are all -inf -- padding rows of the device
    matrix, runs of states without samples -- are requested from the first tile's columns instead of the current tile's
    (RowIdentity::cols: an L2 hit instead of HBM traffic).  Their terms are exactly zero either way, so nothing may change:
    every reduced quantity against the oracle, with real (finite, different) energies in the unsampled rows, and the outputs
    that DO depend on those rows (log numerators of unsampled states, log weights, W^T W) as well."""
    u_kn, N_k, f = random_problem(K, N, seed=11 * K + 1, unsampled=unsampled)
    N = u_kn.shape[1]
    sws = np.where(N_k > 0)[0]
    rng = np.random.default_rng(K)
    c_n = rng.integers(0, 3, size=N).astype(float)
    with DM.from_host(u_kn) as dm:
        check_l1(dm, u_kn, N_k, f, tag=f"K={K}")
        np.testing.assert_allclose(-dm.lognum(f), oracle.self_consistent_update(u_kn, N_k, f), rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(ms.mbar_log_W_nk(dm, N_k, f), oracle.mbar_log_W_nk(u_kn, N_k, f), rtol=1e-12, atol=1e-11)
        W = oracle.mbar_W_nk(u_kn, N_k, f)
        np.testing.assert_allclose(dm.gram_w(f)[0], W.T @ W, rtol=1e-9, atol=1e-14)
        dm.set_sample_weights(c_n)  # (weighted sums: the oracle sees repeated columns)
        u_rep = np.ascontiguousarray(np.repeat(u_kn, c_n.astype(int), axis=1))
        part = oracle.shard_partials(u_rep, N_k, f, want_gram=True)
        psum, _, gram = dm.eval(f, gram=True)
        np.testing.assert_allclose(psum[0], part["psum"], rtol=1e-10, atol=1e-11 * N_k.max())
        np.testing.assert_allclose(gram, part["gram"], rtol=1e-9, atol=1e-12 * N_k.max())
        dm.set_sample_weights(None)
        for opts in (dict(), dict(pmode=0), dict(fused=0), dict(device_loop=0)):
            for k, v in opts.items():
                dm.set_option(k, v)
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-10, min_sc_iter=0)
            f_or, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=1e-10, min_sc_iter=0)
            assert ra["success"]
            np.testing.assert_allclose(fa[sws] - fa[sws[0]], (f_or - f_or[sws[0]])[sws], rtol=1e-8, atol=1e-8, err_msg=str(opts))
            for k in opts:
                dm.set_option(k, 1)
        fs, rs = dm.solve_sci(np.zeros(K), tol=1e-9, maxiter=5000)
        np.testing.assert_allclose(fs[sws] - fs[sws[0]], (f_or - f_or[sws[0]])[sws], atol=1e-6)


def test_module_functions_see_in_place_edits_of_a_host_matrix():
    """The reference's module-level functions are pure functions of the array they are handed (mbar_solvers.py:260-292).  The
    drop-in keeps a device copy of a host matrix between calls; ONE element edited in place between two calls -- outside any
    sampled spot check -- must give the oracle's NEW answer, and an unchanged array must not be uploaded again."""
    ms.drop_resident_cache()
    x_n, u_kn, N_k, s_n, _, _ = ts.config2(seed=3, K=24, N=240_000)
    f = ts.harmonic_free_energies(np.linspace(1.0, 3.0, 24)) * 0.9
    Nf = N_k.astype(float)
    g0 = ms.mbar_gradient(u_kn, N_k, f)
    up0 = ms._resident_cache.uploads
    H0 = ms.mbar_hessian(u_kn, N_k, f)
    assert ms._resident_cache.uploads == up0  # same bytes: the resident copy serves the second call
    np.testing.assert_allclose(g0, oracle.mbar_gradient(u_kn, Nf, f), rtol=1e-10, atol=1e-8)
    u_kn[3, 12_345] += 1.0
    g1 = ms.mbar_gradient(u_kn, N_k, f)
    assert ms._resident_cache.uploads == up0 + 1
    np.testing.assert_allclose(g1, oracle.mbar_gradient(u_kn, Nf, f), rtol=1e-10, atol=1e-8)
    assert np.max(np.abs(g1 - g0)) > 1e-6  # (the edit is visible in the answer: a stale copy would have failed above)
    u_kn[5, 200_001:200_004] = np.inf  # a masked-out run of samples, the other in-place edit users make
    g2 = ms.mbar_gradient(u_kn, N_k, f)
    np.testing.assert_allclose(g2, oracle.mbar_gradient(u_kn, Nf, f), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(ms.mbar_hessian(u_kn, N_k, f), oracle.mbar_hessian(u_kn, Nf, f), rtol=1e-9, atol=1e-8)
    assert ms._resident_cache.uploads == up0 + 2 and len(ms._resident_cache.entries) == 1  # stale copies are dropped at once
    assert np.max(np.abs(H0 - oracle.mbar_hessian(u_kn, Nf, f))) > 1e-9
    ms.drop_resident_cache()


def test_call_sequence_of_the_unchanged_reference_class_on_host_arrays(golden):
    """What the reference's UNCHANGED ``MBAR`` does with ``pymbar.mbar.mbar_solvers`` re-bound to this module, call for call
    (mbar.py:413 ``solve_mbar_for_all_states(self.u_kn, ...)``, :455 ``mbar_log_W_nk(self.u_kn, ...)``, :701-729 Theta from
    ``exp(Log_W_nk)``, :910 ``mbar_log_W_nk`` again per expectation) -- host arrays in, through the C library, ONE upload --
    against the reference's own results for config 5 (the reference tree itself cannot travel to the GPU box)."""
    import copy

    g = golden("config5_alch_K40_N95000.npz")
    x_n, u_kn, N_k, s_n, _, _ = ts.config5(seed=0)
    K = len(N_k)
    ms.drop_resident_cache()
    up0 = ms._resident_cache.uploads
    u_self = np.array(u_kn, dtype=np.float64)                         # mbar.py:243
    sws = np.where(N_k > 0)[0]                                         # mbar.py:383
    protocol = copy.deepcopy(ms.DEFAULT_SOLVER_PROTOCOL)
    for stage in protocol:                                             # mbar.py:391-406
        stage.setdefault("options", dict())
        stage["options"].setdefault("maxiter", 10000)
        stage["options"].setdefault("verbose", False)
    f_k = ms.solve_mbar_for_all_states(u_self, N_k, np.zeros(K), sws, protocol)   # mbar.py:413
    Log_W_nk = ms.mbar_log_W_nk(u_self, N_k, f_k)                                  # mbar.py:455
    again = ms.mbar_log_W_nk(u_self, N_k, f_k)                                     # mbar.py:910
    assert ms._resident_cache.uploads == up0 + 1
    assert Log_W_nk.shape == (u_kn.shape[1], K) and Log_W_nk.flags.f_contiguous and np.array_equal(Log_W_nk, again)
    np.testing.assert_allclose(f_k, g["f_k"], rtol=1e-8, atol=1e-9)
    W = np.exp(Log_W_nk)                                                            # mbar.py:702
    np.testing.assert_allclose(W.sum(0), 1.0, rtol=1e-10)
    np.testing.assert_allclose(W @ N_k, 1.0, rtol=1e-10)
    Theta = oracle.covariance_theta(W, N_k, "svd-ew")                              # mbar.py:703 (restated in the oracle)
    np.testing.assert_allclose(oracle.error_of_differences(Theta), g["dDelta_f_svd_ew"], rtol=1e-7, atol=1e-10)
    ms.drop_resident_cache()


@pytest.mark.parametrize("K,N,unsampled", [(5, 3000, ()), (16, 70_000, (3,)), (20, 9999, (0, 19)), (32, 250_000, ()), (32, 64, ())])
def test_sci_iteration_in_one_launch_agrees_with_sweep_plus_update(DM, K, N, unsampled):
    """Few states, one rank: the self-consistent iteration as ONE launch per iteration (k_sci_small: the update in the prologue
    of the sweep, records and state double-buffered by parity) against the two-kernel iteration (sweep + single-workgroup
    update) -- fixed iteration counts (odd, even, beyond one hipGraph batch), to convergence, with draw counts, eager and
    graph-replayed -- and against the oracle's loop."""
    u_kn, N_k, f = random_problem(K, N, seed=7 * K + 1, unsampled=unsampled)
    N = u_kn.shape[1]  # (emptying state 0 into itself drops its samples)
    sws = np.where(N_k > 0)[0]
    rng = np.random.default_rng(K)
    c_n = np.zeros(N)
    start = 0
    for n_k in N_k:
        if n_k > 0:
            c_n[start:start + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
        start += n_k
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for weights in (None, c_n):
            dm.set_sample_weights(weights)
            for graph in (1, 0):
                dm.set_option("graph", graph)
                for kw in (dict(maxiter=1, check_convergence=False), dict(maxiter=7, check_convergence=False),
                           dict(maxiter=40, check_convergence=False), dict(tol=1e-11, maxiter=5000)):
                    out = {}
                    for merged in (0, 1):
                        dm.set_option("sci_merged", merged)
                        out[merged] = dm.solve_sci(np.zeros(K), **kw)
                    (f0, r0), (f1, r1) = out[0], out[1]
                    assert r0["iterations"] == r1["iterations"] and r0["success"] == r1["success"], (kw, graph, r0, r1)
                    np.testing.assert_allclose(f1[sws], f0[sws], rtol=1e-12, atol=1e-12, err_msg=str((kw, graph)))
                    assert np.array_equal(f1[N_k == 0], np.zeros(int((N_k == 0).sum())))  # states without samples are left alone
        dm.set_sample_weights(None)
        dm.set_option("graph", 1)
        dm.set_option("sci_merged", 1)
        f_dev, r_dev = dm.solve_sci(np.zeros(K), tol=1e-11, maxiter=5000)
    r_or = oracle.sci_solve(u_kn[sws], N_k[sws], np.zeros(len(sws)), tol=1e-11, maxiter=5000)
    assert r_or["success"] and r_dev["success"] and abs(r_or["iterations"] - r_dev["iterations"]) <= 1
    np.testing.assert_allclose(f_dev[sws] - f_dev[sws][0], r_or["x"] - r_or["x"][0], rtol=1e-9, atol=1e-9)


def test_log_denominators_are_not_recomputed_for_the_same_f(DM):
    """The class methods ask for the log-space numerators, W^T W, log W at the same f_k one after the other; the evaluation sweep
    that leaves the per-sample log-denominators behind runs once per (matrix, N_k, f) -- and again after ANYTHING that could
    change them or the slot they live in: another f, new N_k, a changed matrix row, a solve, other sample weights (which do not
    enter the log-denominators: no new sweep, still the right sums)."""
    u_kn, N_k, f = random_problem(48, 40_000, seed=11, unsampled=(5,))
    Nf = N_k.astype(float)
    with DM.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        dm.set_option("timing", 1)

        def sweeps(fn):
            dm.timing_reset()
            out = fn()
            dm.synchronize()
            return out, dm.timing()["lse"][1]

        W = np.exp(oracle.mbar_log_W_nk(u_kn, Nf, f))
        (ln, n1) = sweeps(lambda: dm.lognum(f))
        (gw, n2) = sweeps(lambda: dm.gram_w(f))
        (lw, n3) = sweeps(lambda: dm.logw_kn(f))
        assert (n1, n2, n3) == (1, 0, 0)
        np.testing.assert_allclose(gw[0], W.T @ W, rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(lw.T, np.log(W), rtol=1e-12, atol=1e-10)
        f2 = f + 0.01
        (_, n4) = sweeps(lambda: dm.gram_w(f2))
        (g2, n5) = sweeps(lambda: dm.gram_w(f))
        assert (n4, n5) == (1, 1)
        np.testing.assert_array_equal(g2[0], gw[0])
        dm.solve_adaptive(np.zeros(48), min_sc_iter=0)          # the solver uses the slot vectors for its own purposes
        (g3, n6) = sweeps(lambda: dm.gram_w(f))
        assert n6 == 1
        np.testing.assert_array_equal(g3[0], gw[0])
        N2 = N_k.copy()
        N2[0] += 7
        dm.set_Nk(N2)
        (g4, n7) = sweeps(lambda: dm.gram_w(f))
        assert n7 == 1
        W2 = np.exp(oracle.mbar_log_W_nk(u_kn, N2.astype(float), f))
        np.testing.assert_allclose(g4[0], W2.T @ W2, rtol=1e-10, atol=1e-300)
        dm.set_Nk(N_k)
        dm.gram_w(f)
        row = u_kn[3] + 0.5
        dm.upload_rows(3, row)                                      # a changed matrix
        (g5, n8) = sweeps(lambda: dm.gram_w(f))
        u3 = u_kn.copy()
        u3[3] = row
        W3 = np.exp(oracle.mbar_log_W_nk(u3, Nf, f))
        assert n8 == 1
        np.testing.assert_allclose(g5[0], W3.T @ W3, rtol=1e-10, atol=1e-300)
        c_n = np.random.default_rng(1).integers(0, 3, size=u_kn.shape[1]).astype(float)
        dm.set_sample_weights(c_n)                                  # weights do not enter the log-denominators
        (g6, n9) = sweeps(lambda: dm.gram_w(f))
        assert n9 == 0
        np.testing.assert_allclose(g6[0], (W3 * c_n[:, None]).T @ W3, rtol=1e-10, atol=1e-300)
