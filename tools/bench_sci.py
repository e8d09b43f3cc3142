"""Throughput of the evaluation sweep and of the device-resident SCI loop for several shapes
(BASELINE.json config 2 is K=32, N=1e6).  Prints ms per sweep and algorithmic GB/s (8 K N bytes)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts
from pymbar_amd.device import DeviceMatrix

shapes = [(32, 1_000_000), (32, 8_000_000), (16, 4_000_000), (64, 4_000_000), (96, 4_000_000), (128, 4_000_000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for K, N in shapes:
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
        dm.set_Nk(N_k)
        dm.set_option("small_k_kernel", int(os.environ.get("SMALL_K", "1")))  # 0: the 16-lanes-per-sample kernel for K <= 32
        dm.set_option("sci_merged", int(os.environ.get("SCI_MERGED", "1")))  # 0: sweep + separate update kernel per iteration
        if "SMALL_BALANCED" in os.environ:
            dm.set_option("small_balanced", int(os.environ["SMALL_BALANCED"]))  # 0: wave-major tile streams
        if "SCI_PINGPONG" in os.environ:
            dm.set_option("sci_pingpong", int(os.environ["SCI_PINGPONG"]))  # 1: odd iterations sweep the tiles in descending order
        if "SCI_BATCH" in os.environ:
            dm.set_option("sci_batch", int(os.environ["SCI_BATCH"]))
        f0 = np.zeros(K)
        for timing in (1, 0):
            dm.set_option("timing", timing)
            dm.set_option("graph", 0 if timing else 1)  # timing=1: eager launches with HIP events; timing=0: hipGraph batches
            dm.solve_sci(f0, maxiter=20, check_convergence=False)
            dm.synchronize()
            dm.timing_reset()
            iters = 400
            t0 = time.perf_counter()
            f, res = dm.solve_sci(f0, maxiter=iters, check_convergence=False)
            dm.synchronize()
            dt = time.perf_counter() - t0
            t = dm.timing()
            kern = t["lse"][0] / max(1, t["lse"][1]) if timing else float("nan")
            print(f"K={K:4d} N={N:9d} timing={timing}: {1e3*dt/iters:8.4f} ms/iter ({iters/dt:9.1f} it/s)  "
                  f"wall {8*K*N/(dt/iters)/1e9:8.1f} GB/s   lse kernel {kern:8.4f} ms = {8*K*N/(kern*1e-3)/1e9 if timing else float('nan'):8.1f} GB/s")
        f, res = dm.solve_sci(f0, tol=1e-12)
        print(f"      converged={res['success']} in {res['iterations']} iterations, {res['wall_ms']:.2f} ms")
