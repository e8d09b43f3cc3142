"""The multi-rank code of libmbar_hip.so on real hardware: N sharded by columns over TWO PROCESSES, each
owning one ``mbar_ctx`` with its shard of the same seeded matrix, every reduced output all-reduced across them.

* ``host`` transport: both processes share the one GPU of the test box and reduce through
  ``mbar_ctx_set_host_allreduce`` (rendezvous and transport: ``pymbar_amd.distributed.HostGroup``, standard-library
  TCP).  This drives ``allreduce_dev`` / ``allreduce_host`` / ``agree_with_rank0``, the (max, sum) merge of
  ``mbar_lognum``, the cross-rank NaN flag and the generator's ``n_global0`` offsets.
* ``rccl`` transport: the same program with ``ncclCommInitRank(nranks=2)``; needs two GPUs and skips cleanly otherwise.

Every result must be identical on both ranks (bit for bit), equal the single-context result of the same library to
round-off (only the summation order differs) and equal the CPU oracle."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PROBLEMS = {
    # name: (O_k, K_k, N_k)
    "K6_unsampled": (np.linspace(0, 2, 6), np.linspace(1, 3, 6), [300, 200, 0, 250, 150, 100]),
    "K40": (np.linspace(0, 1.5, 40), np.geomspace(1.0, 16.0, 40), [500] * 40),
    "K128": (np.linspace(0, 4, 128), np.linspace(1, 3, 128), [250] * 128),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem(name):
    from pymbar_amd import testsystems as ts

    O_k, K_k, N_k = PROBLEMS[name]
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=9)
    return np.asarray(O_k, float), np.asarray(K_k, float), u_kn, N_k


def _compute(dm, N_k, K, out, tag):
    """The program every configuration runs (single context or one rank of two)."""
    dm.set_Nk(N_k)
    f0 = np.zeros(K)
    f_ad, r_ad = dm.solve_adaptive(f0, tol=1e-12, maxiter=200, min_sc_iter=0, history_rows=200)
    f_sci, r_sci = dm.solve_sci(f0, tol=1e-11, maxiter=5000)
    rng = np.random.default_rng(5)
    f2 = np.stack([f_ad, f_ad + 0.01 * rng.normal(size=K)])
    f2[:, 0] = 0.0
    psum, sld, G = dm.eval(f2, gram=True)
    out.update({
        f"{tag}f_ad": f_ad, f"{tag}it_ad": r_ad["iterations"], f"{tag}ok_ad": r_ad["success"], f"{tag}hist": r_ad["history"],
        f"{tag}f_sci": f_sci, f"{tag}it_sci": r_sci["iterations"], f"{tag}ok_sci": r_sci["success"],
        f"{tag}psum": psum, f"{tag}sld": sld, f"{tag}G": G, f"{tag}lognum": dm.lognum(f_ad),
    })
    Gw, wsum = dm.gram_w(f_ad)
    out.update({f"{tag}Gw": Gw, f"{tag}wsum": wsum})


def _worker(rank, world, port, out_dir, transport, name):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    from pymbar_amd import _lib
    from pymbar_amd.device import DeviceMatrix
    from pymbar_amd.distributed import HostGroup, attach_allreduce, shard_bounds

    ndev = _lib.device_count()
    dev = rank % ndev if transport == "rccl" else 0
    group = HostGroup(rank, world, base_port=port, token=f"test-{name}-{transport}")
    O_k, K_k, u_kn, N_k = _problem(name)
    K, N = u_kn.shape
    n0, n1 = shard_bounds(N, rank, world)
    out = {"n0": n0, "n1": n1}
    with DeviceMatrix.from_host(u_kn, device=dev, columns=(n0, n1)) as dm:
        kind = attach_allreduce(dm, group, prefer=transport)
        out["kind"] = kind
        _compute(dm, N_k, K, out, "")
    # the on-device generator: the shard starting at global sample n0 must be the columns [n0, n1) of the whole ladder
    Ng = np.asarray(N_k, dtype=np.int64)
    with DeviceMatrix.harmonic(O_k, K_k, Ng, seed=3, n_global0=n0, N_local=n1 - n0, device=dev) as dg:
        out["gen_shard"] = dg.to_host()
        attach_allreduce(dg, group, prefer=transport)
        dg.set_Nk(Ng)
        out["gen_f"], rg = dg.solve_adaptive(np.zeros(K), tol=1e-12, maxiter=200, min_sc_iter=0)
        out["gen_it"] = rg["iterations"]
    # a NaN in ONE shard: every rank must see poisoned sums, nobody may hang in a collective
    bad = u_kn[:, n0:n1].copy()
    if rank == 1:
        bad[1, 5] = np.nan
    with DeviceMatrix.from_host(bad, device=dev) as db:
        attach_allreduce(db, group, prefer=transport)
        db.set_Nk(N_k)
        ps, sl, _ = db.eval(np.zeros(K))
        out["poison_psum"] = ps
        out["poison_lognum"] = db.lognum(np.zeros(K))
        fb, rb = db.solve_sci(np.zeros(K), maxiter=50)
        out["poison_f"] = fb
    group.barrier()
    group.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)


def _run_two_ranks(tmp_path, transport, name):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), transport, name)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    for p in procs:
        if p.is_alive():  # a hang here is the failure the cross-rank flag / loop-exit agreement exist to prevent
            p.kill()
            pytest.fail("a rank did not finish (collective deadlock?)")
        assert p.exitcode == 0
    return [np.load(tmp_path / f"rank{r}.npz", allow_pickle=False) for r in range(2)]


def _check(r0, r1, name, transport):
    from oracle import mbar_oracle as oracle
    from pymbar_amd.device import DeviceMatrix

    assert str(r0["kind"]) == transport and str(r1["kind"]) == transport
    O_k, K_k, u_kn, N_k = _problem(name)
    K, N = u_kn.shape
    assert int(r0["n0"]) == 0 and int(r0["n1"]) == int(r1["n0"]) and int(r1["n1"]) == N
    # both ranks hold bit-identical replicated results
    for key in r0.files:
        if key in ("n0", "n1", "gen_shard", "kind"):
            continue
        np.testing.assert_array_equal(r0[key], r1[key], err_msg=key)
    # against ONE context holding the whole matrix (same kernels; only the order of the partial sums differs)
    single = {}
    with DeviceMatrix.from_host(u_kn, device=0) as dm:
        _compute(dm, N_k, K, single, "")
    assert bool(r0["ok_ad"]) and bool(r0["ok_sci"])
    assert int(r0["it_ad"]) == single["it_ad"]
    assert abs(int(r0["it_sci"]) - single["it_sci"]) <= 1
    for key, tol in (("f_ad", 1e-12), ("f_sci", 1e-9), ("psum", 1e-12), ("sld", 1e-12), ("G", 1e-12), ("lognum", 1e-12),
                     ("Gw", 1e-12), ("wsum", 1e-12)):
        scale = max(1.0, float(np.max(np.abs(single[key]))))
        np.testing.assert_allclose(r0[key], single[key], rtol=tol, atol=tol * scale, err_msg=key)
    # per-iteration gradient norms of both candidates (host-driven loop here, device-resident loop in the single context);
    # norms at round-off level are noise
    big = single["hist"][:, 1:3] > 1e-5
    np.testing.assert_allclose(r0["hist"][:, 1:3][big], single["hist"][:, 1:3][big], rtol=1e-6)
    np.testing.assert_array_equal(r0["hist"][:-1, 0], single["hist"][:-1, 0])  # same choices (the last one is a round-off tie)
    # against the oracle
    sws = np.where(N_k != 0)[0]
    f_ref, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(K), sws, tol=1e-12, min_sc_iter=0)
    f_ad = r0["f_ad"]
    np.testing.assert_allclose(f_ad[sws], f_ref[sws] - f_ref[sws[0]], rtol=1e-9, atol=1e-10)
    Nf = N_k.astype(float)
    np.testing.assert_allclose(r0["psum"][0] - Nf, oracle.mbar_gradient(u_kn, Nf, f_ad), rtol=1e-9, atol=1e-8)
    H = np.diag(r0["psum"][0]) - r0["G"]
    np.testing.assert_allclose(H, oracle.mbar_hessian(u_kn, Nf, f_ad), rtol=1e-10, atol=1e-9)
    W = oracle.mbar_W_nk(u_kn, Nf, f_ad)
    np.testing.assert_allclose(r0["Gw"], W.T @ W, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(r0["wsum"][sws], 1.0, atol=1e-9)
    np.testing.assert_allclose(-r0["lognum"], oracle.self_consistent_update(u_kn, Nf, f_ad), rtol=1e-11, atol=1e-11)
    # generator shards = columns of the whole generated ladder, and the sharded solve on them = the unsharded one
    with DeviceMatrix.harmonic(O_k, K_k, np.asarray(N_k, dtype=np.int64), seed=3) as dg:
        whole = dg.to_host()
        dg.set_Nk(N_k)
        fg, rg = dg.solve_adaptive(np.zeros(K), tol=1e-12, maxiter=200, min_sc_iter=0)
    np.testing.assert_array_equal(r0["gen_shard"], whole[:, int(r0["n0"]):int(r0["n1"])])
    np.testing.assert_array_equal(r1["gen_shard"], whole[:, int(r1["n0"]):int(r1["n1"])])
    np.testing.assert_allclose(r0["gen_f"], fg, rtol=1e-12, atol=1e-12)
    assert int(r0["gen_it"]) == rg["iterations"]
    # the poisoned run: NaN everywhere on BOTH ranks
    sampled = N_k > 0
    assert np.all(np.isnan(r0["poison_psum"])) and np.all(np.isnan(r0["poison_lognum"])) and np.all(np.isnan(r0["poison_f"][sampled]))


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_two_ranks_one_gpu_host_transport(tmp_path, name):
    r0, r1 = _run_two_ranks(tmp_path, "host", name)
    _check(r0, r1, name, "host")


@pytest.mark.parametrize("name", ["K40", "K128"])
def test_two_ranks_rccl(tmp_path, name):
    from pymbar_amd import _lib

    if _lib.device_count() < 2:
        pytest.skip("ncclCommInitRank with nranks = 2 needs two GPUs (RCCL refuses two ranks on one device)")
    r0, r1 = _run_two_ranks(tmp_path, "rccl", name)
    _check(r0, r1, name, "rccl")
