#!/usr/bin/env python
"""Per-iteration cost of the adaptive loop outside its two sweeps: device-resident (hipGraph batches / eager launches)
against the host-driven loop, at config 3 (K=128, N=1e7) and at the sizes pymbar users actually run (config 5: K=40,
N~1e5; config 2: K=32, N=1e6).  Prints one line per case."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402


def run(dm, K, label, steps):
    f0 = np.zeros(K)
    for mode, opts in (("device graph classic", dict(device_loop=1, graph=1, timing=1, pmode=0, fused=0)),
                       ("device graph 2sweep", dict(device_loop=1, graph=1, timing=1, pmode=1, fused=0)),
                       ("device graph", dict(device_loop=1, graph=1, timing=1, pmode=1, fused=1)),
                       ("device eager", dict(device_loop=1, graph=0, timing=1, pmode=1, fused=1)),
                       ("device eager 2sweep", dict(device_loop=1, graph=0, timing=1, pmode=1, fused=0)),
                       ("device eager notime", dict(device_loop=1, graph=0, timing=0, pmode=1, fused=1)),
                       ("library defaults", dict(device_loop=1, graph=1, timing=0, pmode=1, fused=1)),
                       ("host loop", dict(device_loop=0, graph=1, timing=1, pmode=1, fused=1))):
        for k, v in opts.items():
            dm.set_option(k, v)
        dm.solve_adaptive(f0, maxiter=steps, min_sc_iter=0, check_convergence=False)  # warm-up (captures the graph)
        dm.timing_reset()
        dm.synchronize()
        t0 = time.perf_counter()
        f, r = dm.solve_adaptive(f0, maxiter=steps, min_sc_iter=0, check_convergence=False)
        dt = time.perf_counter() - t0
        tm = dm.timing()
        kern = sum(tm[k][0] for k in ("gram", "lse", "fused")) / steps if tm["gram"][1] else float("nan")
        detail = (f"gram {tm['gram'][0] / max(1, tm['gram'][1]):.4f} x{tm['gram'][1]} lse {tm['lse'][0] / max(1, tm['lse'][1]):.4f} "
                  f"x{tm['lse'][1]} fused {tm['fused'][0] / max(1, tm['fused'][1]):.4f} x{tm['fused'][1]} gram_sweeps {r.get('gram_sweeps')}")
        t1 = time.perf_counter()
        fc, rc = dm.solve_adaptive(f0, tol=1e-12, maxiter=10000, min_sc_iter=0)
        dtc = time.perf_counter() - t1
        print(f"{label:22s} {mode:19s}: {1e3 * dt / steps:8.4f} ms/iteration over {steps} (timed sweeps {kern:7.4f} ms/it; {detail}), "
              f"solve from f=0: {rc['iterations']} iterations ({rc.get('gram_sweeps')} Gram sweeps) {1e3 * dtc:8.3f} ms, success={rc['success']}", flush=True)
    return fc


def main():
    cases = [("config5 K=40 N=95000", None), ("config2 K=32 N=1e6", (32, 1_000_000)), ("K=64 N=1e6", (64, 1_000_000)),
             ("config3 K=128 N=1e7", (128, 10_000_000))]
    for label, kn in cases:
        if kn is None:
            x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
            K = u_kn.shape[0]
            dm = DeviceMatrix.from_host(u_kn)
        else:
            K, N = kn
            O_k, K_k, N_k = ts.config3_params(K=K, N=N)
            dm = DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0)
        dm.set_Nk(N_k)
        run(dm, K, label, 40 if K < 128 else 20)
        dm.close()


if __name__ == "__main__":
    main()
