"""The launcher rendezvous of pymbar_amd.distributed (standard-library TCP, no torch): broadcast of the 128-byte
communicator id, all-reduce, barrier -- and the N-sharded solve of tests/test_distributed_gloo.py again with this
transport instead of gloo (world_size 2 and 3, CPU stand-in device)."""
import multiprocessing as mp
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


def _worker(rank, world, port, out_dir, blocker_port):
    sys.path.insert(0, ROOT)
    from pymbar_amd import mbar_solvers as ms
    from pymbar_amd import testsystems as ts
    from pymbar_amd.distributed import HostGroup, attach_allreduce, shard_bounds
    from tests.cpu_standin import OracleMatrix

    # launcher-style environment; MASTER_PORT itself is occupied (as under torch.distributed.run) by `blocker_port`
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(blocker_port),
                      MBAR_RDZV_PORT=str(port))
    group = HostGroup.from_env(timeout=60)
    out = {}
    payload = group.broadcast_bytes(bytes(range(128)) if rank == 0 else None, src=0)
    out["payload_ok"] = payload == bytes(range(128))
    out["none_ok"] = group.broadcast_bytes(None, src=0) is None
    a = np.arange(5, dtype=np.float64) + 10.0 * rank
    out["sum"] = group.allreduce(a.copy(), "sum")
    out["max"] = group.allreduce(a.copy(), "max")
    out["min"] = group.allreduce(a.copy(), "min")
    import pymbar_amd.distributed as dist

    keep = dist._MAX_MSG
    dist._MAX_MSG = 4096  # (every rank alike) an array beyond the message bound travels in pieces
    big = np.arange(3001, dtype=np.float64).reshape(3001, 1) * (rank + 1.0)
    out["big_ok"] = bool(np.array_equal(group.allreduce(big.copy(), "sum"),
                                        np.arange(3001, dtype=np.float64).reshape(3001, 1) * (world * (world + 1) / 2.0)))
    dist._MAX_MSG = keep
    group.barrier()

    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(0, 2, 6), np.linspace(1, 3, 6), [300, 200, 0, 250, 150, 100], seed=9)
    n0, n1 = shard_bounds(u_kn.shape[1], rank, world)
    h = OracleMatrix(u_kn[:, n0:n1])

    class _Handle:  # what attach_allreduce needs of a DeviceMatrix: the stand-in has no RCCL, so "host" must come out
        def comm_init_rccl(self, *_):
            raise RuntimeError("no RCCL on the CPU stand-in")

        def comm_destroy(self):
            out["destroy_called"] = True

        def set_host_allreduce(self, fn, r, n):
            h.allreduce = fn
            out["host_args"] = np.array([r, n])

    # prefer="host": straight to the host transport; prefer="rccl" would ask rank 0's libmbar_hip for a unique id
    out["kind"] = attach_allreduce(_Handle(), group, prefer="host")
    sws = np.where(N_k != 0)[0]
    out["f"] = ms.solve_mbar_for_all_states(h, N_k, np.zeros(6), sws, ms.BOOTSTRAP_SOLVER_PROTOCOL)
    out["H"] = ms.mbar_hessian(h, N_k, out["f"])
    group.barrier()
    group.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)


@pytest.mark.parametrize("world", [2, 3])
def test_hostgroup_collectives_and_sharded_solve(tmp_path, world):
    from oracle import mbar_oracle as oracle
    from pymbar_amd import testsystems as ts

    blocker = socket.socket()  # plays the launcher's store: MASTER_PORT is taken, the group must find its own port
    blocker.bind(("127.0.0.1", 0))
    blocker.listen(1)
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), blocker.getsockname()[1])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    blocker.close()
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    base = np.arange(5, dtype=np.float64)
    for r in rs:
        assert bool(r["payload_ok"]) and bool(r["none_ok"]) and str(r["kind"]) == "host" and bool(r["big_ok"])
        np.testing.assert_array_equal(r["sum"], world * base + 10.0 * sum(range(world)))
        np.testing.assert_array_equal(r["max"], base + 10.0 * (world - 1))
        np.testing.assert_array_equal(r["min"], base)
        np.testing.assert_array_equal(r["f"], rs[0]["f"])  # bit-identical on every rank
        np.testing.assert_array_equal(r["H"], rs[0]["H"])
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(np.linspace(0, 2, 6), np.linspace(1, 3, 6), [300, 200, 0, 250, 150, 100], seed=9)
    sws = np.where(N_k != 0)[0]
    f_ref, _ = oracle.solve_mbar_for_all_states(u_kn, N_k, np.zeros(6), sws, tol=1e-12, min_sc_iter=0)
    np.testing.assert_allclose(rs[0]["f"], f_ref, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(rs[0]["H"], oracle.mbar_hessian(u_kn, N_k, rs[0]["f"]), rtol=1e-11, atol=1e-10)


def test_single_rank_group_is_a_no_op():
    from pymbar_amd.distributed import HostGroup, attach_allreduce

    g = HostGroup(0, 1)
    a = np.ones(3)
    assert g.allreduce(a, "sum") is a and g.broadcast_bytes(b"x") == b"x"
    g.barrier()
    assert attach_allreduce(object(), g) == "none"
    g.close()


def test_no_torch_in_the_product_package():
    """north_star: "no PyTorch".  Nothing under pymbar_amd/ may import torch, not even for the rendezvous."""
    import re

    pkg = os.path.join(ROOT, "pymbar_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+torch\b", src, re.M), os.path.join(dirpath, fn)


def test_listener_address_choice(monkeypatch):
    """Rank 0 listens on loopback for a single node, on the interface MASTER_ADDR names across hosts, and on the wildcard
    address when that name resolves to a loopback alias (Debian's 127.0.1.1 line for the own hostname) or cannot be resolved;
    ``MBAR_RDZV_BIND`` overrides."""
    sys.path.insert(0, ROOT)
    import pymbar_amd.distributed as dist

    monkeypatch.delenv("MBAR_RDZV_BIND", raising=False)
    assert dist._bind_candidates("127.0.0.1") == ["127.0.0.1"] and dist._bind_candidates("localhost") == ["127.0.0.1"]
    monkeypatch.setattr(dist.socket, "gethostbyname", lambda name: {"node7": "10.1.2.3", "debian-host": "127.0.1.1"}[name])
    assert dist._bind_candidates("node7") == ["10.1.2.3", "0.0.0.0"]  # (a VIP that is no local interface: bind fails, wildcard next)
    assert dist._bind_candidates("debian-host") == ["0.0.0.0"]

    def boom(name):
        raise OSError("unresolvable")

    monkeypatch.setattr(dist.socket, "gethostbyname", boom)
    assert dist._bind_candidates("k8s-service") == ["0.0.0.0"]
    monkeypatch.setenv("MBAR_RDZV_BIND", "192.168.0.9")
    assert dist._bind_candidates("node7") == ["192.168.0.9"]
