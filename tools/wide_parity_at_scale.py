#!/usr/bin/env python
"""Above 256 states at a size where round-off could tell the sweeps on the resident probability matrix from the sweeps on u: the
adaptive solve on the device (host-driven loop, option host_pmode 2 / 1 / 0) against the CPU oracle's loop
(oracle.adaptive = mbar_solvers.py:575-640 restated) on the same matrix -- free energies, iteration count, the per-iteration choice
and gradient norms.  Test infrastructure like tests/: the oracle is the checker here, nothing of it is timed or shipped."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mbar_oracle as oracle  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

for K, N in ((384, 200_000), (600, 150_000), (1000, 100_000)):
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    _, u_kn, N_k, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=K)
    hist = []
    t0 = time.perf_counter()
    r_or = oracle.adaptive(np.ascontiguousarray(u_kn), N_k.astype(float), np.zeros(K), tol=1e-10, min_sc_iter=0, history=hist)
    t_or = time.perf_counter() - t0
    gn = np.array([[h["gnorm_sci"], h["gnorm_nr"]] for h in hist])
    with DeviceMatrix.from_host(u_kn) as dm:
        dm.set_Nk(N_k)
        for mode in (2, 1, 0):
            dm.set_option("host_pmode", mode)
            t0 = time.perf_counter()
            fa, ra = dm.solve_adaptive(np.zeros(K), tol=1e-10, maxiter=100, min_sc_iter=0, history_rows=100)
            dt = time.perf_counter() - t0
            it = ra["iterations"]
            same = it == r_or["iterations"]
            df = np.max(np.abs(fa - r_or["x"])) / np.max(np.abs(r_or["x"]))
            n = min(it, len(gn))
            dg = np.max(np.abs(ra["history"][:n, 1:3] - gn[:n]) / np.maximum(gn[:n], 1e-9))
            print(f"K={K} N={N} host_pmode={mode}: {it} iterations (oracle {r_or['iterations']}{'' if same else '  <-- differs'}), "
                  f"max |f - f_oracle| / max |f| = {df:.2e}, gradient norms of every iteration within {dg:.1e} (relative, floor 1e-9), "
                  f"device {1e3 * dt:.1f} ms, oracle {t_or:.1f} s", flush=True)
