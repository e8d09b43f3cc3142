"""N-sharded solve across 2 processes (gloo, CPU): every rank holds a column shard, each pass ends with
one all-reduce of the per-state partial sums (K+1 doubles, or K^2+K for a Gram pass) -- the decomposition
the GPU path uses with RCCL.  The sharded result must agree with the unsharded oracle to
~1e-13 (only the summation order differs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from pymbar_amd import mbar_solvers as ms
    from pymbar_amd import testsystems as ts
    from pymbar_amd.distributed import shard_bounds
    from tests.cpu_standin import OracleMatrix

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def allreduce(arr, op):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(0, 2, 6), np.linspace(1, 3, 6), [300, 200, 0, 250, 150, 100], seed=9)
    N = u_kn.shape[1]
    n0, n1 = shard_bounds(N, rank, world)
    h = OracleMatrix(u_kn[:, n0:n1], allreduce=allreduce)
    sws = np.where(N_k != 0)[0]
    f = ms.solve_mbar_for_all_states(h, N_k, np.zeros(6), sws, ms.BOOTSTRAP_SOLVER_PROTOCOL)
    f_hybr = ms.solve_mbar_for_all_states(h, N_k, np.zeros(6), sws, ms.DEFAULT_SOLVER_PROTOCOL)
    H = ms.mbar_hessian(h, N_k, f)
    G, wsum = h.gram_w(f)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), f=f, f_hybr=f_hybr, H=H, G=G, wsum=wsum, n0=n0, n1=n1)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp

    from oracle import mbar_oracle as oracle
    from pymbar_amd import testsystems as ts

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["n1"]) == int(r1["n0"]) and int(r0["n0"]) == 0 and int(r1["n1"]) == 1000
    assert int(r0["n1"]) % 16 == 0
    # both ranks hold identical replicated results
    for key in ("f", "f_hybr", "H", "G", "wsum"):
        np.testing.assert_array_equal(r0[key], r1[key])
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(0, 2, 6), np.linspace(1, 3, 6), [300, 200, 0, 250, 150, 100], seed=9)
    sws = np.where(N_k != 0)[0]
    f_ref, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(6), sws, tol=1e-12, min_sc_iter=0)
    np.testing.assert_allclose(r0["f"], f_ref, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(r0["f_hybr"], f_ref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r0["H"], oracle.mbar_hessian(u_kn, N_k, r0["f"]), rtol=1e-11, atol=1e-10)
    W = oracle.mbar_W_nk(u_kn, N_k, r0["f"])
    np.testing.assert_allclose(r0["G"], W.T @ W, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(r0["wsum"], 1.0, atol=1e-9)


def test_shard_bounds_cover_everything():
    from pymbar_amd.distributed import shard_bounds

    for N in (1, 15, 16, 17, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            cover = 0
            prev = 0
            for r in range(world):
                n0, n1 = shard_bounds(N, r, world)
                assert n0 == prev and n1 >= n0
                prev = n1
                cover += n1 - n0
            assert cover == N and prev == N
