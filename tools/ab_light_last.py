#!/usr/bin/env python
"""A/B: cold adaptive solves from f = 0 to tol 1e-12 with and without the lighter last sweep (option "light_last"): when both
candidates of the coming iteration already meet the stop test, the fused sweep (candidates + speculated Gram matrix) is replaced
by the plain two-candidate sweep on the resident probability matrix.  Alternating runs on one context, best and median of 7."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402


def main():
    cases = ((128, 10_000_000), (128, 1_000_000), (64, 2_000_000), (32, 4_000_000), (40, 95_000))
    if len(sys.argv) > 1:
        cases = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
    for K, N in cases:
        O_k, K_k, N_k = ts.config3_params(K=K, N=N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            dm.set_option("pcache", 0)  # (cold solves: every one builds the probability matrix)
            out = {0: [], 1: [], 2: []}
            info = {}
            for rep in range(8):
                for mode in (0, 1, 2):
                    dm.set_option("light_last", mode)
                    dm.synchronize()
                    t0 = time.perf_counter()
                    f, r = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
                    dt = time.perf_counter() - t0
                    if rep:
                        out[mode].append(dt)
                    info[mode] = (r["iterations"], r["light_sweeps"], r["gram_sweeps"], r["success"], f.copy())
            assert np.allclose(info[0][4], info[2][4], rtol=0, atol=1e-11)
            line = f"K={K} N={N}:"
            for mode, name in ((0, "off"), (1, "default"), (2, "always")):
                line += (f"  {name}: best {1e3 * min(out[mode]):8.3f} ms median {1e3 * float(np.median(out[mode])):8.3f} ms "
                         f"({info[mode][0]} iterations, {info[mode][1]} light, {info[mode][2]} Gram sweeps)")
            print(line, flush=True)


if __name__ == "__main__":
    main()
