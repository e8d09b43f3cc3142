#!/usr/bin/env python
"""A/B of the build sweep of a cold adaptive solve (k_build_gram: 8 K N bytes in, 8 K N bytes of P out, the Gram matrix at the
anchor) at config 3: run once per library build, alternating, e.g.

    for i in 1 2; do for lib in "" pymbar_amd/csrc/ab/libmbar_hip_pnt.so; do MBAR_HIP_LIBRARY=$lib python tools/ab_build_sweep.py; done; done

Prints the HIP-event time of the build sweep, of the fused sweep, and the wall clock of a cold 5-iteration solve."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

K, N = 128, int(float(os.environ.get("AB_N", "1e7")))
O_k, K_k, N_k = ts.config3_params(K=K, N=N)
dm = DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0)
dm.set_Nk(N_k)
dm.set_option("pcache", 0)
dm.set_option("timing", 1)
dm.set_option("graph", 0)
f0 = np.zeros(K)
dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0)
rows = []
for rep in range(5):
    dm.timing_reset()
    dm.device_synchronize()
    t0 = time.perf_counter()
    f, r = dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0)
    dm.device_synchronize()
    dt = time.perf_counter() - t0
    tm = dm.timing()
    rows.append((tm["other"][0] / max(1, tm["other"][1]), tm["fused"][0] / max(1, tm["fused"][1]), 1e3 * dt, r["iterations"]))
b = np.array(rows)
print(f"{os.environ.get('MBAR_HIP_LIBRARY') or 'default library':48s} K={K} N={N}: build sweep {np.median(b[:, 0]):.3f} ms (min {b[:, 0].min():.3f}), "
      f"fused sweep {np.median(b[:, 1]):.3f} ms, cold solve {np.median(b[:, 2]):.2f} ms / {int(b[0, 3])} iterations", flush=True)
dm.close()
