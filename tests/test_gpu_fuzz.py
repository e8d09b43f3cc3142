"""Randomised parity sweep of the gfx950 path against the CPU oracle: state counts on and around every kernel boundary
(16-state blocks, the 128 / 256 / 512 panel limits), sample counts on and around the tile sizes, random sets of states
without samples, random energy scales, bootstrap draw counts, and a random choice among the loop / sweep variants the
library can run a problem through.  Deterministic (case i is seeded by i); ``MBAR_FUZZ_CASES`` sets how many cases run
(default 48; the committed ``profiles/r3_fuzz_parity.txt`` is a run of 400)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mbar_oracle as oracle  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402

EDGE_K = (1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 40, 47, 48, 49, 63, 64, 65, 96, 111, 112, 113, 127, 128, 129, 130, 144, 159, 160, 161,
          176, 191, 192, 193, 208, 240, 255, 256, 257, 258, 300, 383, 384, 385, 511, 512, 513, 600)
EDGE_N = (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096,
          4097, 8191, 8192, 8193)
OPTIONS = (("pmode", (0, 1)), ("fused", (0, 1)), ("graph", (0, 1)), ("device_loop", (0, 1)), ("gram_quad", (0, 1)),
           ("device_loop_wide", (0, 1)), ("wide_pmode", (0, 1)), ("merge_select", (0, 1)), ("small_k_kernel", (0, 1)), ("wide_k_kernel", (0, 1)), ("pcache", (0, 1)),
           ("quad_trim", (0, 1)), ("adapt_batch", (1, 2, 8)))


def draw_case(i):
    rng = np.random.default_rng(1000 + i)
    K = int(rng.choice(EDGE_K)) if rng.random() < 0.7 else int(rng.integers(1, 601))
    cap = max(64, min(60000, 3_000_000 // K))
    N = int(rng.choice(EDGE_N)) if rng.random() < 0.4 else int(rng.integers(max(1, K // 8), cap))
    N = max(1, min(N, cap))
    # sample counts: multinomial, with a random set of states emptied into state `keep` (at least one state keeps samples)
    N_k = rng.multinomial(N, np.ones(K) / K)
    if K > 1 and rng.random() < 0.6:
        empty = rng.choice(K, size=int(rng.integers(1, max(2, K // 3))), replace=False)
        keep = int(rng.integers(0, K))
        for k in empty:
            if k != keep:
                N_k[keep] += N_k[k]
                N_k[k] = 0
    spread = float(rng.choice([0.3, 1.0, 1.0, 2.5]))
    O_k = np.linspace(0.0, 3.0, K) * spread + 0.1 * rng.standard_normal(K)
    K_k = np.exp(rng.uniform(np.log(0.8), np.log(3.0), size=K))
    _, u_kn, N_k, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=i)
    if rng.random() < 0.3:  # per-sample shifts of hundreds of kT leave every reduced quantity but the objective unchanged
        u_kn = u_kn + 300.0 * rng.standard_normal(N)[None, :]
    f = ts.harmonic_free_energies(K_k) + 0.2 * rng.standard_normal(K)
    f -= f[0]
    opts = {name: int(rng.choice(vals)) for name, vals in OPTIONS if rng.random() < 0.35}
    # (added in round 4, drawn from a stream of its own so that the cases of the earlier logs stay what they were: the last
    # iteration without its Gram matrix at every size / never / by the library's size bound)
    opts["light_last"] = int(np.random.default_rng(810_000 + i).choice([0, 1, 2, 2]))
    c_n = None
    if rng.random() < 0.35:
        c_n, first = np.zeros(N), 0
        for n_k in N_k:
            if n_k > 0:
                c_n[first:first + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
            first += n_k
    return dict(K=K, N=N, u_kn=np.ascontiguousarray(u_kn), N_k=N_k, f=f, opts=opts, c_n=c_n, min_sc_iter=int(rng.choice([0, 0, 2, 3])),
                gamma=float(rng.choice([1.0, 1.0, 0.8])))


def check_case(DM, case):
    K, N, u_kn, N_k, f, c_n = case["K"], case["N"], case["u_kn"], case["N_k"], case["f"], case["c_n"]
    Nf = N_k.astype(float)
    scale = max(1.0, float(Nf.max()))
    sws = np.where(N_k > 0)[0]
    # the oracle sees draw counts as repeated columns
    u_or = u_kn if c_n is None else np.ascontiguousarray(np.repeat(u_kn, c_n.astype(int), axis=1))
    with DM.from_host(u_kn) as dm:
        for name, value in case["opts"].items():
            dm.set_option(name, value)
        dm.set_Nk(N_k)
        dm.set_sample_weights(c_n)
        psum, sld, gram = dm.eval(f, gram=True)
        part = oracle.shard_partials(u_or, N_k, f, want_gram=True)
        np.testing.assert_allclose(psum[0], part["psum"], rtol=1e-10, atol=1e-11 * scale, err_msg="psum")
        np.testing.assert_allclose(sld[0], part["sumlogden"], rtol=1e-11, atol=1e-8, err_msg="sumlogden")
        np.testing.assert_allclose(gram, part["gram"], rtol=1e-9, atol=1e-12 * scale, err_msg="gram")
        f2 = f + 0.05 * np.cos(np.arange(K))
        ps2 = dm.eval(np.stack([f, f2]))[0]
        np.testing.assert_allclose(ps2[1], oracle.shard_partials(u_or, N_k, f2)["psum"], rtol=1e-10, atol=1e-11 * scale, err_msg="psum 2")
        np.testing.assert_allclose(ps2[0], psum[0], rtol=1e-12, atol=1e-12 * scale, err_msg="psum 1 of 2")
        np.testing.assert_allclose(-dm.lognum(f), oracle.self_consistent_update(u_or, N_k, f), rtol=1e-11, atol=1e-10, err_msg="lognum")
        if c_n is None:
            np.testing.assert_allclose(dm.logden(f), oracle.log_denominator(u_kn, N_k, f), rtol=1e-12, atol=1e-11, err_msg="logden")
            W = oracle.mbar_W_nk(u_kn, N_k, f)
            G, wsum = dm.gram_w(f)
            np.testing.assert_allclose(G, W.T @ W, rtol=1e-9, atol=1e-13, err_msg="gram_w")
            np.testing.assert_allclose(wsum, W.sum(0), rtol=1e-10, atol=1e-13, err_msg="wsum")
        if len(sws) < 2:
            return "L1 only (one sampled state)"
        # the adaptive solve: same iteration count as the oracle's loop unless a comparison sits at round-off level, and
        # the gradient of the ORACLE at the returned f is as small as at the oracle's own answer
        tol = 1e-10
        f0 = np.zeros(K)
        fa, ra = dm.solve_adaptive(f0, tol=tol, maxiter=500, min_sc_iter=case["min_sc_iter"], gamma=case["gamma"], history_rows=500)
        hist = []
        ro = oracle.adaptive(u_or[sws], Nf[sws], f0[sws], tol=tol, maxiter=500, min_sc_iter=case["min_sc_iter"], gamma=case["gamma"],
                             history=hist)
        noise_floor = False
        if ro["success"] and not ra["success"]:
            # The convergence test is a RELATIVE change of f (mbar_solvers.py:627-631).  A state whose f is ~1e-5 turns the 1e-14
            # round-off jitter of a self-consistent step (sums of thousands of terms) into a relative change of ~5e-10: with few
            # samples per state the loop then sits AT the solution -- gradient at round-off level, f equal to the oracle's to
            # 1e-13 -- with a relative change a few times the tolerance for ever, where the oracle's pairwise numpy sums happen to
            # land below it (case 4013: K=561, N=2048, layout-agnostic kernels: 4.7e-10 against tol 1e-10; oracle 5.2e-11).  That is
            # the noise floor of the criterion, not a wrong answer: accepted iff the change is within 10 x tol and f is the oracle's.
            noise_floor = (ra["max_delta"] < 10.0 * tol
                           and np.max(np.abs((fa[sws] - fa[sws][0]) - ro["x"])) < 1e-11 * max(1.0, np.max(np.abs(ro["x"]))))
        g_dev = oracle.mbar_gradient(u_or[sws], Nf[sws], fa[sws] - fa[sws][0])
        g_or = oracle.mbar_gradient(u_or[sws], Nf[sws], ro["x"])
        # The opposite corner: states that do not overlap IN fp64 (case 13204: two states 7.5 apart, cross weights ~1e-37).  Every
        # sample then belongs to its own state with p = 1 exactly, the sums equal N_k to the last bit and the device loop stops at
        # once with a zero gradient -- while the reference's log-space arithmetic leaves round-off of 1e-11 in BOTH the gradient
        # and the Hessian, takes "Newton steps" that are noise over noise and never meets its own stop test.  Any f solves such a
        # problem; accepted iff the oracle's own gradient at OUR f is at its round-off level too.
        disconnected = (ra["success"] and not ro["success"] and np.abs(g_dev).max() <= 1e-8 * scale
                        and np.abs(g_or).max() <= 1e-8 * scale)
        # ... and its mirror image (cases 17453 / 17455: two sampled states with ONE sample each, no overlap in fp64).  The 1 x 1
        # Newton system is h x = g with h and g both a few ulps: noise over noise, in the reference (lstsq keeps the largest singular
        # value whatever its size, mbar_solvers.py:582) as here.  Where that quotient happens to be 0 the stop test passes (the
        # oracle's arithmetic: 11 / 15 iterations); where it is O(1) the self-consistent candidate wins every choice, f stops
        # changing at once (max_delta = 0 exactly) and `max_diff < sqrt(tol)` (:636) is never met -- with either Newton solve of this
        # library, whichever bits the exponentials leave.  Any f solves such a problem: accepted iff f has stopped moving and the
        # oracle's gradient at OUR f is at its round-off level.
        noise_newton = (ro["success"] and not ra["success"] and ra["max_delta"] == 0.0 and np.abs(g_dev).max() <= 1e-8 * scale
                        and np.abs(g_or).max() <= 1e-8 * scale)
        assert ra["success"] == ro["success"] or noise_floor or disconnected or noise_newton, (ra, ro["iterations"])
        if disconnected:
            return f"{ra['iterations']} iterations; states without overlap in fp64: the reference does not converge ({ro['iterations']})"
        if noise_newton:
            return f"f stationary; states without overlap in fp64: the Newton candidate is noise over noise, the stop test never met ({ro['iterations']})"
        assert np.abs(g_dev).max() <= 10.0 * np.abs(g_or).max() + 1e-8 * scale, (np.abs(g_dev).max(), np.abs(g_or).max())
        if ra.get("psum") is not None:
            # (mbar_gradient = psum - N_k; non-zero where the reference's loop itself runs out of iterations -- case 4089)
            np.testing.assert_allclose((ra["psum"] - Nf)[sws], g_dev, rtol=0, atol=1e-9 * scale + 1e-6 * np.abs(g_dev).max(), err_msg="psum at the result")
        close_call = any(abs(h["gnorm_sci"] - h["gnorm_nr"]) <= 1e-6 * max(h["gnorm_sci"], h["gnorm_nr"]) + 1e-9 * scale for h in hist)
        near_tol = any(0.1 * tol < h["max_delta"] < 10 * tol for h in hist)
        if noise_floor:
            return f"{ra['iterations']} iterations, relative change {ra['max_delta']:.1e} at the round-off floor of the criterion ({ro['iterations']})"
        if not close_call and not near_tol:
            assert ra["iterations"] == ro["iterations"], (ra["iterations"], ro["iterations"], ra["nr_iter"], ro["nr_iter"])
            np.testing.assert_allclose(fa[sws] - fa[sws][0], ro["x"], rtol=1e-7, atol=1e-7, err_msg="f")
        return f"{ra['iterations']} iterations ({ro['iterations']})"


def test_randomised_problems_match_the_oracle():
    from pymbar_amd.device import DeviceMatrix

    n = int(os.environ.get("MBAR_FUZZ_CASES", "48"))
    first = int(os.environ.get("MBAR_FUZZ_FIRST", "0"))
    log = os.environ.get("MBAR_FUZZ_LOG")
    lines, failures = [], []
    for i in range(first, first + n):
        case = draw_case(i)
        tag = (f"case {i}: K={case['K']} N={case['N']} sampled={int((case['N_k'] > 0).sum())} weights={case['c_n'] is not None} "
               f"min_sc_iter={case['min_sc_iter']} gamma={case['gamma']} options={case['opts']}")
        try:
            lines.append(tag + " -> " + check_case(DeviceMatrix, case))
        except Exception as exc:  # noqa: BLE001  (a library error in one case must not hide the others)
            failures.append(tag + "\n" + type(exc).__name__ + ": " + str(exc)[:1500])
            lines.append(tag + " -> FAILED")
    if log:
        with open(log, "w") as fh:
            fh.write("\n".join(lines) + f"\n{len(lines) - len(failures)} of {len(lines)} cases agree with the oracle\n")
    assert not failures, "\n\n".join(failures[:5]) + f"\n({len(failures)} of {n} cases failed)"


def check_case_across_ranks(case, i):
    """The same problem on 2, 3, 4 or 8 logical ranks with RANDOM shard boundaries, empty shards included (in-process stream transport, as in
    tests/test_gpu_loopback.py): ranks bit-identical to each other, and equal to the single-context run up to the
    summation order."""
    from pymbar_amd.device import DeviceMatrix, LoopbackGroup
    from tests.test_gpu_loopback import run_ranks

    K, N, u_kn, N_k, f, c_n = case["K"], case["N"], case["u_kn"], case["N_k"], case["f"], case["c_n"]
    rng = np.random.default_rng(77 + i)
    nranks = int(rng.choice([2, 3, 4, 8]))  # (the target is 8 GPUs: rank counts up to there, not only 2-3)
    # cut points drawn WITH replacement from 0 .. N: repeated cuts are ranks with an EMPTY shard, which must still take part in
    # every collective (N = 1e7 over 8 GPUs never has one, a 100-sample problem over 8 ranks does)
    cuts = np.sort(rng.integers(0, N + 1, size=nranks - 1))
    bounds = [0] + [int(c) for c in cuts] + [N]
    sws = np.where(N_k > 0)[0]
    scale = max(1.0, float(N_k.max()))
    tol = 1e-10

    def run(dm, c_local):
        for name, value in case["opts"].items():
            dm.set_option(name, value)
        dm.set_Nk(N_k)
        dm.set_sample_weights(c_local)
        out = dict(eval=dm.eval(f, gram=True), lognum=dm.lognum(f))
        if len(sws) >= 2:
            out["solve"] = dm.solve_adaptive(np.zeros(K), tol=tol, maxiter=500, min_sc_iter=case["min_sc_iter"], gamma=case["gamma"],
                                             history_rows=500)
        return out

    with DeviceMatrix.from_host(u_kn) as one:
        ref = run(one, c_n)
    with LoopbackGroup(nranks) as grp:
        def worker(r):
            n0, n1 = bounds[r], bounds[r + 1]
            with DeviceMatrix.from_host(u_kn, columns=(n0, n1)) as dm:
                dm.set_loopback(grp, r)
                res = run(dm, None if c_n is None else c_n[n0:n1])
                dm.comm_destroy()
                return res

        ranks = run_ranks(nranks, worker)
    r0 = ranks[0]
    for r in ranks[1:]:
        for a, b in zip(r0["eval"], r["eval"]):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(r0["lognum"], r["lognum"], equal_nan=True)
        if "solve" in r0:
            assert np.array_equal(r0["solve"][0], r["solve"][0], equal_nan=True) and r0["solve"][1]["iterations"] == r["solve"][1]["iterations"]
    for a, b in zip(r0["eval"], ref["eval"]):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11 * scale)
    np.testing.assert_allclose(r0["lognum"], ref["lognum"], rtol=1e-12, atol=1e-11)
    if "solve" not in r0:
        return f"{nranks} ranks {bounds}: L1 only"
    (fa, ra), (fr, rr) = r0["solve"], ref["solve"]
    assert ra["success"] == rr["success"]
    h = rr["history"]
    close_call = np.any(np.abs(h[:, 1] - h[:, 2]) <= 1e-6 * np.maximum(h[:, 1], h[:, 2]) + 1e-9 * scale)
    near_tol = np.any((h[:, 3] > 0.1 * tol) & (h[:, 3] < 10 * tol)) if h.shape[1] > 3 else False
    if not close_call and not near_tol:
        assert ra["iterations"] == rr["iterations"], (ra["iterations"], rr["iterations"])
        np.testing.assert_allclose(fa[sws], fr[sws], rtol=1e-8, atol=1e-8)
    return f"{nranks} ranks {bounds}: {ra['iterations']} iterations ({rr['iterations']})"


def test_randomised_problems_across_logical_ranks():
    n = int(os.environ.get("MBAR_FUZZ_RANK_CASES", "16"))
    first = int(os.environ.get("MBAR_FUZZ_FIRST", "0"))
    log = os.environ.get("MBAR_FUZZ_RANK_LOG")
    lines, failures = [], []
    for i in range(first, first + n):
        case = draw_case(5000 + i)
        tag = (f"case {5000 + i}: K={case['K']} N={case['N']} sampled={int((case['N_k'] > 0).sum())} weights={case['c_n'] is not None} "
               f"min_sc_iter={case['min_sc_iter']} gamma={case['gamma']} options={case['opts']}")
        try:
            lines.append(tag + " -> " + check_case_across_ranks(case, i))
        except Exception as exc:  # noqa: BLE001
            failures.append(tag + "\n" + type(exc).__name__ + ": " + str(exc)[:1500])
            lines.append(tag + " -> FAILED")
    if log:
        with open(log, "w") as fh:
            fh.write("\n".join(lines) + f"\n{len(lines) - len(failures)} of {len(lines)} cases: ranks bit-identical, equal to one context\n")
    assert not failures, "\n\n".join(failures[:5]) + f"\n({len(failures)} of {n} cases failed)"
