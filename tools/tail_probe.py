#!/usr/bin/env python
"""What an adaptive iteration costs outside its sweep, and what a cold solve costs, at config 3 and at its 8-rank shard size --
the numbers the round-6 A/B builds are compared on (run it plain for wall-clock figures, under ``rocprofv3 --kernel-trace`` +
tools/trace_timeline.py for the per-kernel timeline):

    python tools/tail_probe.py [label]

Per shape: ms per forced iteration over 30 (library defaults: hipGraph batches, no event timers), the build sweep and the fused
sweep from a second run with event timers, and the best of 5 cold solves from f = 0 to tol 1e-12."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default build"
for K, N in ((128, 10_000_000), (128, 1_250_000)):
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
        dm.set_Nk(N_k)
        dm.set_option("pcache", 0)
        f0 = np.zeros(K)
        steps = 30
        dm.solve_adaptive(f0, maxiter=steps, min_sc_iter=0, check_convergence=False)
        best = 1e9
        for _ in range(3):
            dm.device_synchronize()
            t0 = time.perf_counter()
            dm.solve_adaptive(f0, maxiter=steps, min_sc_iter=0, check_convergence=False)
            dm.device_synchronize()
            best = min(best, time.perf_counter() - t0)
        dm.set_option("timing", 1)
        dm.set_option("graph", 0)
        dm.timing_reset()
        dm.solve_adaptive(f0, maxiter=steps, min_sc_iter=0, check_convergence=False)
        dm.device_synchronize()
        tm = dm.timing()
        fused = tm["fused"][0] / max(1, tm["fused"][1])
        build = tm["other"][0] / max(1, tm["other"][1])
        dm.set_option("timing", 0)
        dm.set_option("graph", 1)
        cold = 1e9
        for _ in range(5):
            dm.device_synchronize()
            t0 = time.perf_counter()
            f, r = dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0)
            dm.device_synchronize()
            cold = min(cold, time.perf_counter() - t0)
        ms_it = 1e3 * best / steps
        print(f"[{label}] K={K} N={N}: {ms_it:.4f} ms per forced iteration ({steps} in one call, build included), fused sweep {fused:.4f} ms, "
              f"build sweep {build:.4f} ms, outside the sweeps {ms_it - fused - build / steps:.4f} ms; cold solve {1e3 * cold:.3f} ms "
              f"({r['iterations']} iterations, success={r['success']}), max |f - analytic| = {np.max(np.abs(f - ts.harmonic_free_energies(K_k))):.2e}", flush=True)
