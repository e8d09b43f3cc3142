#!/usr/bin/env python
"""What a user of the class sees around the solver: `pymbar_amd.MBAR` construction in its variants (private copy / copy=False,
bootstrap replicates, initialize="BAR") and the Log_W_nk consumers (free energy differences, overlap, expectations in their forms,
perturbed free energies, entropy / enthalpy, a histogram free energy surface), at config 5, K=64 / N=1e6 and K=128 / N=4e6.

Usage: python tools/bench_class.py [ctor] [calls] [fes]     (default: all)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymbar_amd  # noqa: E402
from pymbar_amd import fes  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402

SHAPES = (("config 5-like K=40 N=1e5", 40, 100_000), ("K=64 N=1e6", 64, 1_000_000), ("K=128 N=4e6", 128, 4_000_000))


def problem(K, N):
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    x_n, u_kn, N_k, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
    return x_n, u_kn, N_k


def timeit(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e3 * (time.perf_counter() - t0) / n


def ctor():
    for name, K, N in SHAPES:
        x_n, u_kn, N_k = problem(K, N)
        out = []
        for label, kw in (("MBAR(u_kn, N_k)", {}), ("copy=False", dict(copy=False)), ("n_bootstraps=10", dict(n_bootstraps=10)),
                          ("n_bootstraps=10, bootstrap_rng='device'", dict(n_bootstraps=10, bootstrap_rng="device")),
                          ('initialize="BAR"', dict(initialize="BAR"))):
            m = pymbar_amd.MBAR(u_kn, N_k, **kw)
            m.close()
            del m  # (the release of the previous object's host copy stays out of the timed region)
            t0 = time.perf_counter()
            m = pymbar_amd.MBAR(u_kn, N_k, **kw)
            dt = time.perf_counter() - t0
            m.close()
            del m
            out.append(f"{label} {1e3 * dt:.1f}")
        print(f"{name}: constructor ms: " + " | ".join(out), flush=True)


def calls():
    for name, K, N in SHAPES:
        x_n, u_kn, N_k = problem(K, N)
        m = pymbar_amd.MBAR(u_kn, N_k, copy=False)
        A2 = np.stack([x_n, x_n ** 2])
        u_new = u_kn[:3] * 1.1
        print(f"{name}: ms per call:",
              "free_energy_differences %.2f" % timeit(lambda: m.compute_free_energy_differences()),
              "| overlap %.2f" % timeit(lambda: m.compute_overlap()),
              "| expectations(A_n) %.2f" % timeit(lambda: m.compute_expectations(x_n)),
              "| expectations(A_n, u_kn = 3 new states) %.2f" % timeit(lambda: m.compute_expectations(x_n, u_kn=u_new)),
              "| multiple_expectations(2 observables) %.2f" % timeit(lambda: m.compute_multiple_expectations(A2, u_kn[0])),
              "| perturbed_free_energies(3) %.2f" % timeit(lambda: m.compute_perturbed_free_energies(u_new)),
              "| entropy_and_enthalpy %.2f" % timeit(lambda: m.compute_entropy_and_enthalpy()), flush=True)
        m.close()


def surface():
    for (name, K, N), nb in zip(SHAPES, (30, 50, 100)):
        x_n, u_kn, N_k = problem(K, N)
        m = pymbar_amd.MBAR(u_kn, N_k, copy=False)
        edges = np.linspace(x_n.min(), x_n.max() + 1e-9, nb + 1)
        t_lab = timeit(lambda: fes.label_samples(x_n, edges), 3)
        lab, _ = fes.label_samples(x_n, edges)
        t_fes = timeit(lambda: fes.histogram_fes(m, u_kn[0], lab), 3)
        print(f"{name}: {nb} bins: label_samples {t_lab:.1f} ms, histogram_fes (analytical uncertainties) {t_fes:.1f} ms", flush=True)
        m.close()


if __name__ == "__main__":
    want = sys.argv[1:] or ["ctor", "calls", "fes"]
    for w in want:
        {"ctor": ctor, "calls": calls, "fes": surface}[w]()
