#!/usr/bin/env python
"""Benchmark of the MBAR solver hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "MBAR solver iterations/sec + wallclock-to-converge, K=128 N=1e7"):
config 3 -- the synthetic harmonic ladder O_k = linspace(0,4,K), K_k = linspace(1,3,K), equal N_k, K = 128,
N = 1e7 samples PER GPU (weak scaling; 8 GPUs = 8e7 samples, config 4's regime), fp64, generated directly in
HBM.  A step is ONE adaptive iteration (pymbar/mbar_solvers.py:575-640): a Gram/Hessian sweep on the fp64
matrix cores at the current f, the K x K Newton solve, and one sweep evaluating the gradients of both
candidates (f_sci, f_nr), with one all-reduce after each sweep when N > 1.  The timed region runs exactly K
iterations with the convergence test disabled (the work per iteration does not depend on f); wall-clock
to converge from f = 0 at tol 1e-12 is measured separately and reported in the same JSON line.

``value`` = iterations/s x n_gpus, i.e. shard-iterations/s: every rank processes its own K x 1e7 shard each
iteration, so at N = 1 this is the plain solver iterations/s of BASELINE.json.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X public spec (fp64 matrix); MI355X_MICROARCH.md lists no fp64 MFMA figure
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def pmc_traffic(kernel_prefix, K, n_loc):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_fetch_write.json,
    newest round first), corrected as MI355X_MICROARCH.md prescribes (gfx950: FETCH_SIZE x 2).  None if no
    profile of this exact workload is committed -- PMC collection needs its own rocprofv3 run."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") != {"K": K, "N_per_gpu": n_loc}:
            continue
        for name, v in d["kernels"].items():
            if kernel_prefix in name and "FETCH_SIZE_KB_mean_per_launch" in v:
                return (2.0 * v["FETCH_SIZE_KB_mean_per_launch"] + v.get("WRITE_SIZE_KB_mean_per_launch", 0.0)) * 1024.0
    return None


def cpu_baseline(dm_factory, K, N_full, n_sample, seed):
    """Time ONE adaptive iteration of the CPU oracle (the numpy/scipy restatement of the reference's numpy
    path: Hessian + SCI update + two gradients + lstsq, mbar_solvers.py:581-594) on the first ``n_sample``
    columns of the same workload and extrapolate linearly in N (every sweep is O(K N))."""
    from oracle import mbar_oracle as oracle
    from pymbar_amd import testsystems as ts

    O_k, K_k, N_k = ts.config3_params(K=K, N=N_full)
    with dm_factory(n_sample) as sub:
        u = sub.to_host()
    Nk = N_k.astype(np.float64)
    f = ts.harmonic_free_energies(K_k) * 0.9
    t0 = time.perf_counter()
    g = oracle.mbar_gradient(u, Nk, f)                       # carried over from the previous iteration
    t_grad = time.perf_counter() - t0
    t0 = time.perf_counter()
    H = oracle.mbar_hessian(u, Nk, f)
    Hinvg = np.linalg.lstsq(H, g, rcond=-1)[0]
    Hinvg -= Hinvg[0]
    f_nr = f - Hinvg
    f_sci = oracle.self_consistent_update(u, Nk, f)
    f_sci = f_sci - f_sci[0]
    g_sci = oracle.mbar_gradient(u, Nk, f_sci)
    g_nr = oracle.mbar_gradient(u, Nk, f_nr)
    _ = np.dot(g_sci, g_sci) < np.dot(g_nr, g_nr)
    t_iter = time.perf_counter() - t0
    it_per_s_full = (1.0 / t_iter) * (n_sample / float(N_full))
    return {
        "value": it_per_s_full,
        "unit": "iter/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": f"one adaptive iteration of oracle/mbar_oracle.py (numpy {np.__version__}, scipy logsumexp, "
                  f"BLAS threads = all cores) on the first {n_sample} of {N_full} columns, K={K}: "
                  f"{t_iter:.2f} s measured (gradient alone {t_grad:.2f} s), scaled linearly by {N_full / n_sample:.0f}x",
        "seconds_per_iteration_extrapolated": t_iter * N_full / n_sample,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--n-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="columns for the CPU baseline (0 = skip)")
    ap.add_argument("--staging", type=int, default=0)
    ap.add_argument("--lse-variant", type=int, default=1)
    ap.add_argument("--gram-variant", type=int, default=2)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): --n-per-gpu samples on EVERY GPU, value = solver it/s x GPUs; "
                         "strong: --n-per-gpu samples in TOTAL, split over the GPUs, value = solver it/s")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix, device_info

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")  # rendezvous + barriers only; the data path is RCCL inside libmbar_hip

    def barrier_sync(dm):
        if dist is not None:
            dist.barrier()
        dm.synchronize()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    K = args.K
    if args.scaling == "weak":
        n_loc = args.n_per_gpu
        N_total = n_loc * world
    else:  # the same problem on more GPUs: contiguous 16-aligned column shards
        from pymbar_amd.distributed import shard_bounds

        N_total = args.n_per_gpu
        n0_strong, n1_strong = shard_bounds(N_total, rank, world)
        n_loc = n1_strong - n0_strong
    O_k, K_k, N_k = ts.config3_params(K=K, N=N_total)
    N_k = N_k.copy()
    N_k[-1] += N_total - int(N_k.sum())  # keep sum(N_k) == N_total when K does not divide it

    from pymbar_amd import _lib

    ndev = max(1, _lib.device_count())
    dev = local_rank % ndev  # one rank per GPU under the launcher; wraps only when ranks outnumber devices (testing)
    info = device_info(dev)
    n_global0 = rank * n_loc if args.scaling == "weak" else n0_strong
    dm = DeviceMatrix.harmonic(O_k, K_k, N_k, seed=args.seed, n_global0=n_global0, N_local=n_loc, device=dev)
    dm.set_option("staging", args.staging)
    dm.set_option("lse_variant", args.lse_variant)
    dm.set_option("gram_variant", args.gram_variant)
    dm.set_Nk(N_k)
    allreduce = "none"
    if world > 1:
        from pymbar_amd.distributed import attach_allreduce

        allreduce = attach_allreduce(dm)

    f0 = np.zeros(K)
    # ---- warm-up (untimed) ----
    if args.warmup > 0:
        dm.solve_adaptive(f0, tol=1e-12, maxiter=args.warmup, min_sc_iter=0, check_convergence=False)
    # ---- timed: exactly `steps` adaptive iterations ----
    dm.timing_reset()
    barrier_sync(dm)
    t0 = time.perf_counter()
    f_end, res = dm.solve_adaptive(f0, tol=1e-12, maxiter=args.steps, min_sc_iter=0, check_convergence=False)
    barrier_sync(dm)
    elapsed = time.perf_counter() - t0
    timing = dm.timing()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- wall-clock to converge from f = 0 (reported, not the headline value) ----
    barrier_sync(dm)
    t0 = time.perf_counter()
    f_conv, conv = dm.solve_adaptive(f0, tol=1e-12, maxiter=10000, min_sc_iter=0, check_convergence=True)
    barrier_sync(dm)
    t_conv = time.perf_counter() - t0
    err_analytic = float(np.max(np.abs(f_conv - ts.harmonic_free_energies(K_k))))

    mfma_peak = dm.mfma_f64_peak() if rank == 0 else None

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        def factory(n):
            return DeviceMatrix.harmonic(O_k, K_k, N_k, seed=args.seed, n_global0=0, N_local=n, device=dev)
        cpu = cpu_baseline(factory, K, n_loc, min(args.cpu_sample, n_loc), args.seed)

    if rank == 0:
        it_per_s = args.steps / elapsed
        gram_ms, gram_n = timing["gram"]
        lse_ms, lse_n = timing["lse"]
        gram_avg = gram_ms / max(1, gram_n)
        lse_avg = lse_ms / max(1, lse_n)
        flops = float(n_loc) * K * (K + 1)           # symmetric Gram: K(K+1)/2 entries x 2 flop x N
        bytes_pass = 8.0 * K * n_loc                 # one read of the shard per sweep
        # the adaptive loop issues 1 single-f sweep (initial gradient) + `steps` two-candidate sweeps
        achieved_tf = flops / (gram_avg * 1e-3) * 1e-12 if gram_avg > 0 else 0.0
        achieved_gbs = bytes_pass / (lse_avg * 1e-3) * 1e-9 if lse_avg > 0 else 0.0
        out = {
            "metric": "mbar_adaptive_iterations_per_sec",
            "value": it_per_s * (world if args.scaling == "weak" else 1),
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"config3: harmonic ladder K={K}, N={n_loc} per GPU (N_total={N_total}), adaptive NR/SCI "
                            f"iteration = MFMA Gram sweep + 2-candidate gradient sweep, fp64, generated in HBM",
                "K": K, "N_per_gpu": n_loc, "N_total": N_total, "parallelism": f"N-sharded x{world}",
                "allreduce": allreduce, "device": info["name"], "solver_iterations_per_sec": it_per_s,
            },
            "roofline": {
                "kernel": ({0: "k_gram_xchg<8>", 1: "k_gram_pair<8>"}.get(args.gram_variant, "k_gram<8,8> one wave per SIMD") + " (fp64 MFMA W^T W)") if K == 128 else "k_gram",
                "bound": "mfma", "achieved": achieved_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic("k_gram<", K, n_loc),
                "avg_launch_ms": gram_avg, "launches": gram_n, "algorithmic_flop_per_launch": flops,
                "measured_mfma_f64_peak_tflops": mfma_peak,
            },
            "roofline_lse": {
                "kernel": "k_lse<8,2> (log-sum-exp + per-state sums, 2 candidates per sweep, one exp per element)",
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("k_lse<8, 2", K, n_loc),
                "avg_launch_ms": lse_avg, "launches": lse_n, "algorithmic_bytes_per_launch": bytes_pass,
            },
            "cpu_baseline": cpu,
            "wallclock_to_converge_s": t_conv,
            "iterations_to_converge": int(conv["iterations"]),
            "converged": bool(conv["success"]),
            "nr_iterations": int(conv["nr_iter"]), "sci_iterations": int(conv["sci_iter"]),
            "max_abs_error_vs_analytic_f": err_analytic,
            "gnorm_at_solution": float(conv["gnorm"]),
        }
        if cpu is not None:
            out["speedup_vs_cpu_baseline"] = it_per_s / cpu["value"]
        print(json.dumps(out))
    dm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
