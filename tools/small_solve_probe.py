#!/usr/bin/env python
"""Whole cold adaptive solves (f = 0 to tol 1e-12, best of 30) at the sizes pymbar is mostly used at, with the blocked LDL^T Newton solve
(newton_ldlt = 1, default) and the register Gauss-Jordan solve of rounds 2-5 (0), alternating on one resident matrix:

    python tools/small_solve_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts
from pymbar_amd.device import DeviceMatrix
for K, N in ((5, 5000), (16, 20000), (32, 100000), (40, 95000), (64, 200000), (100, 300000)):
    O = np.linspace(0, 3, K); Kk = np.linspace(1, 2.5, K); Nk = np.full(K, N // K); Nk[-1] += N - Nk.sum()
    with DeviceMatrix.harmonic(O, Kk, Nk, seed=1) as dm:
        dm.set_Nk(Nk)
        out = []
        for ldlt in (1, 0, 1, 0):
            dm.set_option("newton_ldlt", ldlt)
            dm.set_option("pcache", 0)
            f0 = np.zeros(K)
            dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0)
            best = 1e9
            for _ in range(30):
                t0 = time.perf_counter(); f, r = dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0); best = min(best, time.perf_counter() - t0)
            out.append(f"ldlt={ldlt}: {1e6*best:.0f} us ({r['iterations']} it)")
        print(f"K={K} N={N}: " + " | ".join(out), flush=True)
