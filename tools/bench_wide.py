#!/usr/bin/env python
"""State counts beyond one Gram panel (K > 128): what the sweeps and the adaptive iteration cost there.

  gram    W^T W Gram sweep (mbar_gram_w) at K = 160 ... 256: 128-state panels + 64 x 128 rectangles (2.5 reads of the matrix)
          against the one-read kernel whose four waves share a tile stream (k_gram_quad), in ms and as a fraction of the fp64
          matrix peak on N K (K + 1) flop;
  loop    one adaptive iteration at K = 192 / 256: host-driven loop (panels / one-read Gram) against the device-resident loop
          (blocked Cholesky Newton solve in device memory) in its two forms: two sweeps on u (one-read Gram + four-waves-per-CU
          evaluation sweep), and ONE fused sweep on the resident probability matrix (k_fused_quad); cold and warm solves;
  split   evaluation sweep at 257 ... 512 states: one and two candidates, row-split kernel against the layout-agnostic path.

  large   one adaptive iteration above 256 states (host-driven loop: paneled Gram sweep, one-read two-candidate evaluation, K x K
          solve on the host): wall clock per iteration, the device timers behind it, the Gram sweep as a fraction of the fp64 matrix peak;

  trim    129 ... 160 / 193 ... 224 states: the one-read sweeps without the two padding blocks of their panel.

Usage: python tools/bench_wide.py [gram] [loop] [split] [trim] [large]   (default: the first three)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

PEAK = 78.6


def ladder(K, N):
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    N_k[-1] += N - N_k.sum()
    return O_k, K_k, N_k


def gram():
    for K, N in ((160, 2_000_000), (192, 2_000_000), (256, 2_000_000), (256, 4_000_000)):
        O_k, K_k, N_k = ladder(K, N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            dm.set_option("timing", 1)
            f = ts.harmonic_free_energies(K_k)
            res = {}
            for q in (0, 1):
                dm.set_option("gram_quad", q)
                G, _ = dm.gram_w(f)
                dm.timing_reset()
                for _ in range(6):
                    dm.gram_w(f)
                res[q] = (G, dm.timing()["gram"][0] / 6)
            flop = N * K * (K + 1.0)
            d = np.max(np.abs(res[0][0] - res[1][0])) / np.max(np.abs(res[0][0]))
            print(f"gram_w K={K} N={N}: panels {res[0][1]:.3f} ms = {flop / res[0][1] * 1e-9 / PEAK:.3f} of the fp64 matrix peak | one read "
                  f"{res[1][1]:.3f} ms = {flop / res[1][1] * 1e-9 / PEAK:.3f} | largest relative difference {d:.1e}", flush=True)


def loop():
    for K, N in ((256, 4_000_000), (192, 4_000_000)):
        O_k, K_k, N_k = ladder(K, N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            for name, opts in (("host-driven loop, panel Gram", dict(device_loop=0, gram_quad=0)),
                               ("host-driven loop, one-read Gram", dict(device_loop=0, gram_quad=1)),
                               ("device-resident, two sweeps on u", dict(device_loop=1, gram_quad=1, wide_pmode=0)),
                               ("device-resident, fused on P", dict(device_loop=1, gram_quad=1, wide_pmode=1))):
                for k, v in {"pcache": 0, **opts}.items():
                    dm.set_option(k, v)
                dm.solve_adaptive(np.zeros(K), maxiter=3, min_sc_iter=0, check_convergence=False)
                dm.synchronize()
                t0 = time.perf_counter()
                dm.solve_adaptive(np.zeros(K), maxiter=20, min_sc_iter=0, check_convergence=False)
                dt = (time.perf_counter() - t0) / 20
                t1 = time.perf_counter()
                fc, rc = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
                dtc = time.perf_counter() - t1
                dm.set_option("pcache", 1)
                dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
                t2 = time.perf_counter()
                fw, rw = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
                dtw = time.perf_counter() - t2
                print(f"adaptive K={K} N={N} {name:34s}: {1e3 * dt:7.3f} ms per iteration (20 in one cold solver call); solve from f=0: "
                      f"{rc['iterations']} iterations {1e3 * dtc:7.2f} ms cold, {1e3 * dtw:7.2f} ms warm ({rw['warm_starts']} warm start), "
                      f"success={rc['success']}", flush=True)


def trim():
    """129 .. 160 and 193 .. 224 states: the one-read sweeps with and without the two padding blocks of the 192- / 256-row panel."""
    for K, N in ((144, 4_000_000), (160, 4_000_000), (208, 4_000_000), (224, 4_000_000)):
        O_k, K_k, N_k = ladder(K, N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            f = ts.harmonic_free_energies(K_k)
            out = []
            for q in (0, 1):
                dm.set_option("quad_trim", q)
                dm.set_option("timing", 1)
                G, _ = dm.gram_w(f)
                dm.timing_reset()
                for _ in range(6):
                    dm.gram_w(f)
                tg = dm.timing()["gram"][0] / 6
                dm.set_option("timing", 0)
                dm.set_option("pcache", 0)
                dm.solve_adaptive(np.zeros(K), maxiter=3, min_sc_iter=0, check_convergence=False)
                dm.synchronize()
                t0 = time.perf_counter()
                dm.solve_adaptive(np.zeros(K), maxiter=20, min_sc_iter=0, check_convergence=False)
                dt = (time.perf_counter() - t0) / 20
                t1 = time.perf_counter()
                fc, rc = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
                dtc = time.perf_counter() - t1
                flop = N * K * (K + 1.0)
                out.append((G, fc, f"{'trimmed' if q else 'whole panel'}: gram_w {tg:.3f} ms = {flop / tg * 1e-9 / PEAK:.3f} of the fp64 matrix peak on "
                            f"N K (K + 1) flop, iteration {1e3 * dt:.3f} ms, solve from f=0 {rc['iterations']} iterations {1e3 * dtc:.2f} ms"))
            same = np.array_equal(out[0][0], out[1][0])
            print(f"K={K} N={N}: " + " | ".join(o[2] for o in out) + f" | Gram bit-identical: {same}, largest |df| {np.max(np.abs(out[0][1] - out[1][1])):.1e}",
                  flush=True)


def split():
    for K, N in ((257, 2_000_000), (300, 1_000_000), (384, 1_000_000), (512, 1_000_000), (512, 4_000_000), (600, 1_000_000),
                 (768, 1_000_000), (1000, 1_000_000), (1024, 2_000_000)):
        O_k, K_k, N_k = ladder(K, N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            f = ts.harmonic_free_energies(K_k)
            f2 = np.stack([f, f * 0.99])
            gb = 8.0 * K * N * 1e-9
            out = []
            for force in (1, 0):
                dm.set_option("force_generic", force)
                for name, fn in (("1 candidate", lambda: dm.eval(f)), ("2 candidates", lambda: dm.eval(f2))):
                    fn()
                    dm.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        fn()
                    dt = (time.perf_counter() - t0) / 10
                    out.append(f"{'layout-agnostic' if force else 'row-split'} {name} {1e3 * dt:7.3f} ms ({gb / dt * 1e-3:4.2f} TB/s of one read)")
            print(f"eval K={K} N={N}: " + " | ".join(out), flush=True)


def large():
    for K, N in ((320, 1_000_000), (384, 1_000_000), (512, 1_000_000), (512, 4_000_000), (768, 1_000_000), (1024, 1_000_000)):
        O_k, K_k, N_k = ladder(K, N)
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
            dm.set_Nk(N_k)
            dm.set_option("timing", 1)
            if "RECT_WAVES" in os.environ:
                dm.set_option("rect_waves", int(os.environ["RECT_WAVES"]))  # 4: one wave per SIMD in the 128 x 256 rectangles
            if "HOST_PMODE" in os.environ:
                dm.set_option("host_pmode", int(os.environ["HOST_PMODE"]))  # 0: Gram sweep on u (exponentials per panel)
            dm.solve_adaptive(np.zeros(K), maxiter=2, min_sc_iter=0, check_convergence=False)
            dm.timing_reset()
            dm.synchronize()
            n = 6
            t0 = time.perf_counter()
            dm.solve_adaptive(np.zeros(K), maxiter=n, min_sc_iter=0, check_convergence=False)
            dt = (time.perf_counter() - t0) / n
            tm = dm.timing()
            per = {k: (tm[k][0] / max(1, tm[k][1]), tm[k][1]) for k in ("gram", "lse", "other")}
            flop = float(N) * K * (K + 1)
            t1 = time.perf_counter()
            fc, rc = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
            dtc = time.perf_counter() - t1
            print(f"adaptive K={K} N={N}: {1e3 * dt:8.3f} ms per iteration; Gram sweep {per['gram'][0]:7.3f} ms x{per['gram'][1]} = "
                  f"{flop / (tm['gram'][0] / n * 1e-3) * 1e-12 / PEAK:5.3f} of the fp64 matrix peak over an iteration's launches, evaluation sweep {per['lse'][0]:7.3f} ms x{per['lse'][1]} "
                  f"({8.0 * K * N / (per['lse'][0] * 1e-3) * 1e-12:4.2f} TB/s of one read), other {per['other'][0]:7.3f} ms x{per['other'][1]}; device time per iteration "
                  f"{sum(tm[k][0] for k in ('gram', 'lse', 'other')) / n:8.3f} ms; solve from f=0: {rc['iterations']} iterations {1e3 * dtc:8.2f} ms success={rc['success']}", flush=True)


if __name__ == "__main__":
    want = sys.argv[1:] or ["gram", "loop", "split"]
    for w in want:
        {"gram": gram, "loop": loop, "split": split, "trim": trim, "large": large}[w]()
