"""Soak test: every reduced output must be bit-identical from launch to launch (fixed-order reductions, no atomics) and
the pinned-accumulator Gram kernels must agree with the compiler-scheduled variant.  A matrix-core hazard or a race
would show up here as run-to-run noise.   Usage: python tools/soak_determinism.py [repetitions]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for K, N in ((128, 1_000_003), (128, 10_000_000), (256, 400_000), (192, 300_000), (112, 500_000), (40, 95_001), (32, 2_000_000), (16, 1_000_000)):
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=1) as dm:
        dm.set_Nk(N_k)
        rng = np.random.default_rng(K)
        f = ts.harmonic_free_energies(K_k) + 0.05 * rng.normal(size=K)
        f[0] = 0.0
        f2 = np.stack([f, f + 0.01 * rng.normal(size=K)])
        ref = None
        for r in range(reps):
            psum, sld, G = dm.eval(f2, gram=True)
            ln = dm.lognum(f)
            h = hashlib.sha256(psum.tobytes() + sld.tobytes() + G.tobytes() + ln.tobytes()).hexdigest()
            if ref is None:
                ref = h
            assert h == ref, f"K={K}: launch {r} differs from launch 0"
        # the device-resident loop (resident probability matrix, fused sweep with hand-placed matrix instructions, LDS-DMA three
        # quarters of an iteration ahead): the whole solve bit for bit, with and without bootstrap multiplicities
        sreps = max(4, reps // 4)
        for weighted in (False, True):
            if weighted:
                c_n = np.zeros(N)
                start = 0
                for n_k in N_k:
                    c_n[start:start + n_k] = np.bincount(rng.integers(0, n_k, size=n_k), minlength=n_k)
                    start += n_k
                dm.set_sample_weights(c_n)
            # cold solves (the build sweep every time) and warm ones (the resident probability matrix of the previous solve
            # re-used: other bits than a cold start, the same bits among themselves)
            for pcache in (0, 1):
                dm.set_option("pcache", pcache)
                dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)  # (pcache = 1: the solve that leaves the matrix behind)
                fs, res = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
                for r in range(sreps // 2):
                    fs2, res2 = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0, history_rows=64)
                    assert np.array_equal(fs, fs2) and res["iterations"] == res2["iterations"], f"K={K} weighted={weighted} pcache={pcache}: solve {r} differs"
                    assert np.array_equal(np.asarray(res.get("history", 0)), np.asarray(res2.get("history", 0)))
        dm.set_sample_weights(None)
        print(f"K={K:4d} N={N:8d}: {reps} evaluations bit-identical ({ref[:12]}), 2 x {sreps} solves reproducible ({res['iterations']} iterations weighted)")
print("OK")
