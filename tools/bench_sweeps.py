#!/usr/bin/env python
"""Throughput of the secondary sweeps the API hits (HIP-event time per launch -> algorithmic TB/s = 8 K N / t):
single- and two-candidate evaluation sweeps, the log-space per-state reduction (mbar_lognum) and the W^T W Gram sweep, over
a range of state counts.  One line per (K, N)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402


def main():
    cases = [(16, 8_000_000), (32, 8_000_000), (40, 4_000_000), (48, 4_000_000), (64, 4_000_000), (96, 4_000_000),
             (128, 4_000_000), (192, 2_000_000), (256, 2_000_000)]
    if len(sys.argv) > 1:
        cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    reps = 8
    for K, N in cases:
        O_k, K_k, N_k = ts.config3_params(K=K, N=N)
        N_k[-1] += N - N_k.sum()
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            dm.set_option("timing", 1)  # (HIP-event timers are off by default)
            f = ts.harmonic_free_energies(K_k)
            f2 = np.stack([f, f * 0.99])
            gb = 8.0 * K * N * 1e-9
            out = []
            for name, fn, cls in (("eval nf=1", lambda: dm.eval(f), "lse"), ("eval nf=2", lambda: dm.eval(f2), "lse"),
                                  ("lognum", lambda: dm.lognum(f), "other"), ("gram_w", lambda: dm.gram_w(f), "gram")):
                fn()
                dm.timing_reset()
                for _ in range(reps):
                    fn()
                tm = dm.timing()
                ms, n = tm[cls]
                if name in ("lognum", "gram_w"):  # these also run one evaluation sweep per call (class "lse")
                    pass
                per = ms / reps
                out.append(f"{name} {per:7.3f} ms {gb / per:6.2f} TB/s")
            if N * K <= 2.5e8:  # log W (16 K N bytes of HBM traffic + a K N download): only where the host copy is small
                dm.timing_reset()
                dm.logw_kn(f)
                ms, n = dm.timing()["other"]
                out.append(f"logw kernel {ms:7.3f} ms {2 * gb / ms:6.2f} TB/s")
            print(f"K={K:4d} N={N:9d}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
