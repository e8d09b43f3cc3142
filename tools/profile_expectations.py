#!/usr/bin/env python
"""Where compute_perturbed_free_energies / compute_entropy_and_enthalpy spend their time at K=128, N=4e6 (the general
augmented path of the expectation family): wall clock per call and, under ``rocprofv3 --kernel-trace --stats``, the kernels.

    python tools/profile_expectations.py [perturbed|entropy|expect3|expect1|all] [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymbar_amd  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K, N = 128, 4_000_000
O_k, K_k, N_k = ts.config3_params(K=K, N=N)
N_k[-1] += N - N_k.sum()
x_n, u_kn, N_k, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
m = pymbar_amd.MBAR(u_kn, N_k, copy=False)
u_new = u_kn[:3] * 1.1
calls = {
    "perturbed": lambda: m.compute_perturbed_free_energies(u_new),
    "entropy": lambda: m.compute_entropy_and_enthalpy(),
    "expect3": lambda: m.compute_expectations(x_n, u_kn=u_new),
    "expect1": lambda: m.compute_expectations(x_n),
}
for name, fn in calls.items():
    if what not in ("all", name):
        continue
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(1e3 * (time.perf_counter() - t0))
    print(f"{name}: {np.median(t):.2f} ms per call (median of {reps}; min {min(t):.2f})", flush=True)
    if os.environ.get("MBAR_PROFILE_PY"):
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        fn()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
m.close()
