"""The multi-rank DEVICE-RESIDENT path on a one-GPU box: several contexts of this process, one caller thread and one column
shard each, joined by the in-process transport (``mbar_loopback``: an all-reduce on the compute streams, like RCCL's).

What runs here is what runs across GPUs under RCCL and nowhere else on one device (RCCL refuses two ranks per device; the host
transport of tests/test_gpu_multirank.py sends the solve to the host-driven loop): the resident probability matrix per shard,
the fused sweep, ONE all-reduce of {per-state sums, Gram records} per iteration on the stream, the K x K Newton solve and
the candidate selection computed redundantly on every rank from bit-identical reduced values, the pause / resume of a
rejected speculation agreed through the control words, the rank-global decision about P mode.

Checked: the ranks agree BIT FOR BIT with each other; with the single-context solve to 1e-11 (the summation order over
shards differs), in iteration counts and per-iteration gradient norms; with the CPU oracle."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mbar_oracle as oracle  # noqa: E402
from pymbar_amd.distributed import shard_bounds  # noqa: E402
from tests.test_gpu_parity import random_problem  # noqa: E402


def run_ranks(nranks, worker, timeout=400.0):
    """worker(rank) in one thread per rank; returns the list of results, re-raises the first failure."""
    out, err = [None] * nranks, [None] * nranks

    def body(r):
        try:
            out[r] = worker(r)
        except BaseException as exc:  # noqa: BLE001
            err[r] = exc

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    errors = [(r, e) for r, e in enumerate(err) if e is not None]
    if errors:
        # the root cause first: a rank that fails leaves its peers waiting in a collective ("a peer did not arrive")
        roots = [(r, e) for r, e in errors if "did not arrive" not in str(e)] or errors
        r, e = roots[0]
        raise AssertionError(f"rank {r} of {nranks}: {type(e).__name__}: {e} (ranks with errors: {[x for x, _ in errors]})") from e
    return out


def bootstrap_counts(N_k, seed):
    rng = np.random.default_rng(seed)
    c_n = np.zeros(int(N_k.sum()))
    start = 0
    for n_k in N_k:
        if n_k > 0:
            c_n[start:start + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
        start += n_k
    return c_n


CASES = (dict(min_sc_iter=0), dict(min_sc_iter=2), dict(min_sc_iter=0, weights=True), dict(min_sc_iter=0, fixed=11),
         dict(min_sc_iter=3, gamma=0.7))


def solve_cases(dm, K, c_n, tol=1e-12):
    out = []
    for case in CASES:
        dm.set_sample_weights(c_n if case.get("weights") else None)
        try:
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=tol, maxiter=case.get("fixed", 200), min_sc_iter=case["min_sc_iter"],
                                       gamma=case.get("gamma", 1.0), check_convergence="fixed" not in case, history_rows=200)
        finally:
            dm.set_sample_weights(None)
        out.append((fa, ra))
    return out


@pytest.mark.parametrize("K,N,unsampled,nranks", [(40, 20000, (7, 23), 2), (128, 30011, (5,), 2), (64, 9000, (), 3), (5, 3000, (), 2),
                                                  (200, 16000, (11,), 2), (128, 30011, (5,), 4), (128, 100003, (), 8), (40, 400, (7,), 8)])
def test_device_resident_loop_across_logical_ranks(K, N, unsampled, nranks):
    from pymbar_amd.device import DeviceMatrix, LoopbackGroup

    u_kn, N_k, f = random_problem(K, N, seed=K + 3, unsampled=unsampled)
    sws = np.where(N_k > 0)[0]
    c_n = bootstrap_counts(N_k, K)
    tol = 1e-12 if K <= 128 else 1e-10  # (few samples per state above 128 states: 1e-12 is the round-off floor of f there)
    with DeviceMatrix.from_host(u_kn) as one:
        one.set_Nk(N_k)
        ref = solve_cases(one, K, c_n, tol)
        ref_eval = one.eval(f, gram=True)
        ref_lognum = one.lognum(f)
        ref_gw = one.gram_w(f)
    with LoopbackGroup(nranks) as grp:
        def worker(r):
            n0, n1 = shard_bounds(N, r, nranks)
            with DeviceMatrix.from_host(u_kn, columns=(n0, n1)) as dm:
                dm.set_loopback(grp, r)
                assert dm.allreduce_kind == "loopback"
                dm.set_Nk(N_k)
                res = dict(solves=solve_cases(dm, K, c_n[n0:n1], tol), eval=dm.eval(f, gram=True), lognum=dm.lognum(f), gw=dm.gram_w(f),
                           sci=dm.solve_sci(np.zeros(K), tol=1e-10, maxiter=3000))
                dm.comm_destroy()
                return res

        ranks = run_ranks(nranks, worker)
    r0 = ranks[0]
    for r in ranks[1:]:  # every rank computed the same thing from the same reduced values
        for (fa, ra), (fb, rb) in zip(r0["solves"], r["solves"]):
            assert np.array_equal(fa, fb) and ra["iterations"] == rb["iterations"] and np.array_equal(ra["history"], rb["history"])
        for a, b in zip(r0["eval"], r["eval"]):
            assert np.array_equal(a, b)
        assert np.array_equal(r0["lognum"], r["lognum"]) and np.array_equal(r0["gw"][0], r["gw"][0])
        assert np.array_equal(r0["sci"][0], r["sci"][0])
    scale = float(N_k.max())
    for a, b in zip(r0["eval"], ref_eval):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(r0["lognum"], ref_lognum, rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(r0["gw"][0], ref_gw[0], rtol=1e-11, atol=1e-300)
    for case, (fa, ra), (fr, rr) in zip(CASES, r0["solves"], ref):
        assert ra["success"] == rr["success"], (case, ra, rr)
        # (a relative change that lands within a factor of a few of the tolerance converges an iteration earlier or later
        # depending on the summation order over the shards: 1.04e-12 against tol 1e-12 at 8 ranks)
        h_all = np.concatenate([ra["history"][:, 3], rr["history"][:, 3]])
        near_tol = bool(np.any((h_all > 0.2 * tol) & (h_all < 5.0 * tol)))
        if near_tol and "fixed" not in case and abs(ra["iterations"] - rr["iterations"]) == 1:
            np.testing.assert_allclose(fa[sws], fr[sws], rtol=1e-10, atol=1e-10, err_msg=str(case))
            continue
        assert ra["iterations"] == rr["iterations"], (case, ra["iterations"], rr["iterations"])
        # (the choice between two gradient norms at round-off level is noise, and the shards sum in another order)
        clear = np.abs(rr["history"][:, 1] - rr["history"][:, 2]) > 1e-7
        assert np.array_equal(ra["history"][clear, 0], rr["history"][clear, 0]), case
        np.testing.assert_allclose(fa[sws], fr[sws], rtol=1e-11, atol=1e-11, err_msg=str(case))
        big = rr["history"][:, 1:3] > 1e-6
        np.testing.assert_allclose(ra["history"][:, 1:3][big], rr["history"][:, 1:3][big], rtol=1e-6, err_msg=str(case))
        assert ra["gram_sweeps"] == rr["gram_sweeps"]
    f_or, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=tol, min_sc_iter=0)
    np.testing.assert_allclose(r0["solves"][0][0][sws], (f_or - f_or[sws[0]])[sws], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r0["sci"][0][sws], r0["solves"][0][0][sws], atol=1e-7)


def test_ranks_agree_on_p_mode_and_on_hand_backs():
    """(a) One rank cannot have its resident probability matrix (stood in for by switching P mode off on that rank only): every
    rank must fall back to the classic sweeps -- ranks in different modes would wait for each other in different all-reduces.
    (b) A start hundreds of kT from the answer: k_newton hands the solve back on every rank at the same iteration, one
    host-driven iteration runs through the same transport, the device loop re-anchors.  (c) NaN in one shard only."""
    from pymbar_amd.device import DeviceMatrix, LoopbackGroup

    K, N, nranks = 40, 12000, 2
    u_kn, N_k, f = random_problem(K, N, seed=17)
    u_far = u_kn + 30.0 * np.arange(K)[:, None]
    u_nan = u_kn.copy()
    u_nan[3, N - 5] = np.nan
    with DeviceMatrix.from_host(u_kn) as one:
        one.set_Nk(N_k)
        f_ref, r_ref = one.solve_adaptive(np.zeros(K), min_sc_iter=0)
    with DeviceMatrix.from_host(u_far) as one:
        one.set_Nk(N_k)
        f_far, r_far = one.solve_adaptive(np.zeros(K), min_sc_iter=0, maxiter=300)
        assert r_far["success"]
    with LoopbackGroup(nranks) as grp:
        def worker(r):
            n0, n1 = shard_bounds(N, r, nranks)
            res = {}
            for name, u in (("plain", u_kn), ("far", u_far), ("nan", u_nan)):
                with DeviceMatrix.from_host(u, columns=(n0, n1)) as dm:
                    dm.set_loopback(grp, r)
                    dm.set_Nk(N_k)
                    if name == "plain" and r == 1:
                        dm.set_option("pmode", 0)
                    res[name] = dm.solve_adaptive(np.zeros(K), min_sc_iter=0, maxiter=5 if name == "nan" else 300)
                    if name == "nan":
                        res["nan_eval"] = dm.eval(f)[0]
                    dm.comm_destroy()
            return res

        ranks = run_ranks(nranks, worker)
    for name in ("plain", "far"):
        assert np.array_equal(ranks[0][name][0], ranks[1][name][0])
    np.testing.assert_allclose(ranks[0]["plain"][0], f_ref, rtol=1e-11, atol=1e-11)
    assert ranks[0]["plain"][1]["iterations"] == r_ref["iterations"]
    assert ranks[0]["far"][1]["success"] and ranks[0]["far"][1]["iterations"] == r_far["iterations"]
    np.testing.assert_allclose(ranks[0]["far"][0], f_far, rtol=1e-11, atol=1e-9)
    for r in ranks:  # the poisoned shard poisons every rank's sums, and nobody hangs
        assert np.all(np.isnan(r["nan_eval"]))
        assert np.all(np.isnan(r["nan"][0][1:]))
