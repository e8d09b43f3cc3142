#!/usr/bin/env python
"""Time the UNMODIFIED reference (pymbar's numpy/scipy solver path, pymbar/mbar_solvers.py:76-87) on this host.

Run it with the reference on the path and the interpreter the survey found fastest (recipe A):

    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tools/time_reference.py --out profiles/r2_reference_cpu_timing.json

The workload is bench.py's: the harmonic ladder O_k = linspace(0, 4, K), K_k = linspace(1, 3, K), equal N_k, K = 128, drawn
by the reference's own HarmonicOscillatorsTestCase (so the matrix is the reference's, not ours).  K = 128, N = 1e7 needs
~45 GB for one gradient call of the reference, so -- as SURVEY.md 8(d) prescribes -- N in {1e6, 2e6, 4e6} is timed and
the per-iteration cost is extrapolated linearly in N (every sweep is O(K N); the iteration count does not depend on N).

Timed per N (one untimed warm-up call each, then the best of --reps): mbar_gradient, mbar_hessian, self_consistent_update,
ONE adaptive iteration assembled from the reference's own functions exactly as mbar_solvers.py:575-607 does (Hessian, lstsq,
SCI update, two gradients), and -- for the smallest N -- the full solve_mbar_once(method="adaptive", tol=1e-12, min_sc_iter=0).
The extrapolation to N = 1e7 uses the LARGEST N timed (the per-sample cost grows with N on this host: the (N, K)
temporaries leave the caches / the memory controller's comfort zone), so it is a lower bound of the reference's cost there.

bench.py quotes the JSON written here as ``cpu_baseline.reference_build_host`` (the reference tree is not available
on the GPU box, so this number is measured on the build container and committed).
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--N", type=int, nargs="+", default=[1_000_000, 2_000_000, 4_000_000])
    ap.add_argument("--full-solve-N", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3, help="timed repetitions per quantity (the best is kept) after one warm-up call")
    ap.add_argument("--iteration-only-N", type=int, nargs="*", default=[],
                    help="sizes at which ONLY one adaptive iteration is timed, once, without a warm-up call (bounded host time "
                         "on a GPU box whose minutes are metered); said so in the row")
    ap.add_argument("--solve-only-N", type=int, nargs="*", default=[],
                    help="sizes at which ONLY the full adaptive solve from f = 0 is run (once): wall clock to converge and "
                         "iteration count of the reference itself at that size -- config 3 is --solve-only-N 10000000 (~50 GB)")
    ap.add_argument("--host-label", default="build container", help="which host this is (goes into the record)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import pymbar
    from pymbar import mbar_solvers as ref
    from pymbar.testsystems import harmonic_oscillators

    ref_tree = os.path.realpath(os.environ.get("MBAR_REFERENCE_TREE", "/root/reference"))
    assert os.path.realpath(pymbar.__file__).startswith(ref_tree), pymbar.__file__
    import scipy

    K = args.K
    O_k = np.linspace(0.0, 4.0, K)
    K_k = np.linspace(1.0, 3.0, K)
    rows = []
    for N in sorted(set(args.N) | set(args.iteration_only_N) | set(args.solve_only_N)):
        N_k = np.full(K, N // K, dtype=np.int64)
        tc = harmonic_oscillators.HarmonicOscillatorsTestCase(O_k, K_k)
        x_n, u_kn, N_k_out, s_n = tc.sample(N_k, mode="u_kn", seed=0)
        u_kn = np.ascontiguousarray(u_kn, dtype=np.float64)
        Nf = N_k.astype(np.float64)
        f = tc.analytical_free_energies()
        f = 0.9 * (f - f[0])

        def best_of(fn, *a):
            """One untimed warm-up call (page faults of the temporaries, BLAS thread start-up), then the best of `reps`."""
            fn(*a)
            best, out = float("inf"), None
            for _ in range(args.reps):
                t0 = time.perf_counter()
                out = fn(*a)
                best = min(best, time.perf_counter() - t0)
            return out, best

        def one_iteration(g):
            # one adaptive iteration, the reference's own statements (mbar_solvers.py:581-607); g at f is carried over
            H = ref.mbar_hessian(u_kn, Nf, f)
            Hinvg = np.linalg.lstsq(H, g, rcond=-1)[0]
            Hinvg -= Hinvg[0]
            f_nr = f - Hinvg
            f_sci = ref.self_consistent_update(u_kn, Nf, f)
            f_sci = f_sci - f_sci[0]
            g_sci = ref.mbar_gradient(u_kn, Nf, f_sci)
            g_nr = ref.mbar_gradient(u_kn, Nf, f_nr)
            return np.dot(g_sci, g_sci) < np.dot(g_nr, g_nr)

        if N in args.solve_only_N and N not in args.N:
            import io
            import logging

            log = io.StringIO()
            handler = logging.StreamHandler(log)
            logging.getLogger("pymbar.mbar_solvers").addHandler(handler)
            logging.getLogger("pymbar.mbar_solvers").setLevel(logging.INFO)
            t0 = time.perf_counter()
            f_out, res = ref.solve_mbar_once(u_kn, Nf, np.zeros(K), method="adaptive", tol=1e-12,
                                             options=dict(min_sc_iter=0, maxiter=10000, verbose=True))
            wall = time.perf_counter() - t0
            logging.getLogger("pymbar.mbar_solvers").removeHandler(handler)
            text = log.getvalue()
            iters = text.count("Newton-Raphson gradient norm is")  # one such line per iteration (mbar_solvers.py:599-603)
            row = dict(N=N, full_solve_s=wall, full_solve_iterations=iters, full_solve_success=bool(res["success"]),
                       adaptive_iteration_s=wall / max(1, iters), timing="ONE cold solve from f = 0, tol 1e-12, min_sc_iter 0; "
                       "adaptive_iteration_s = wall / iterations",
                       f_k=[float(v) for v in (f_out - f_out[0])])
            print(json.dumps({k: v for k, v in row.items() if k != "f_k"}), flush=True)
            rows.append(row)
            del u_kn, x_n
            continue
        if N not in args.N:  # iteration only, one cold run
            g = ref.mbar_gradient(u_kn, Nf, f)
            t0 = time.perf_counter()
            one_iteration(g)
            row = dict(N=N, adaptive_iteration_s=time.perf_counter() - t0, timing="ONE run, no warm-up call")
            print(json.dumps(row), flush=True)
            rows.append(row)
            del u_kn, x_n
            continue
        g, t_grad = best_of(ref.mbar_gradient, u_kn, Nf, f)
        H, t_hess = best_of(ref.mbar_hessian, u_kn, Nf, f)
        fs, t_sci = best_of(ref.self_consistent_update, u_kn, Nf, f)
        _, t_iter = best_of(one_iteration, g)
        row = dict(N=N, mbar_gradient_s=t_grad, mbar_hessian_s=t_hess, self_consistent_update_s=t_sci,
                   adaptive_iteration_s=t_iter, timing="one warm-up call, then best of %d" % args.reps)
        if N == args.full_solve_N:
            t0 = time.perf_counter()
            f_out, res = ref.solve_mbar_once(u_kn, Nf, np.zeros(K), method="adaptive", tol=1e-12,
                                             options=dict(min_sc_iter=0, maxiter=10000, verbose=False))
            row["full_solve_s"] = time.perf_counter() - t0
            row["full_solve_timing"] = "single run, after the warm calls above"
            row["full_solve_success"] = bool(res["success"])
            row["full_solve_max_abs_error_vs_analytic"] = float(np.max(np.abs((f_out - f_out[0]) - (f / 0.9))))
        print(json.dumps(row), flush=True)
        rows.append(row)
        del u_kn, x_n

    per_sample_rows = [r["adaptive_iteration_s"] / r["N"] for r in rows]
    per_sample = per_sample_rows[int(np.argmax([r["N"] for r in rows]))]
    blas = None
    try:
        from threadpoolctl import threadpool_info

        blas = [dict(api=i.get("user_api"), lib=i.get("internal_api"), threads=i.get("num_threads")) for i in threadpool_info()]
    except Exception:
        pass
    out = {
        "what": "UNMODIFIED reference pymbar (numpy backend, scipy logsumexp) timed on: " + args.host_label,
        "host_label": args.host_label,
        "K": K,
        "rows": rows,
        "adaptive_iteration_seconds_per_sample_by_row": per_sample_rows,
        "adaptive_iteration_seconds_per_sample": per_sample,
        "adaptive_iteration_s_extrapolated_N1e7": per_sample * 1e7,
        "value_iter_per_s_extrapolated_N1e7": 1.0 / (per_sample * 1e7),
        "nproc": os.cpu_count(), "threadpools": blas,
        "unit": "iter/s",
        "kind": "reference",
        "cores": os.cpu_count(),
        "host": platform.processor() or platform.machine(), "cpu_model": cpu_model(),
        "cores_this_process_may_use": len(os.sched_getaffinity(0)),
        "python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
        "blas_threads": os.environ.get("OPENBLAS_NUM_THREADS", os.environ.get("OMP_NUM_THREADS", "default (all cores)")),
        "extrapolation": "linear in N from the largest N timed (every sweep is O(K N))",
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
