#!/usr/bin/env python
"""Benchmark of the MBAR solver hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "MBAR solver iterations/sec + wallclock-to-converge, K=128 N=1e7, 1/2/4/8 GPUs"): the
synthetic harmonic ladder O_k = linspace(0,4,K), K_k = linspace(1,3,K), equal N_k, K = 128, fp64, generated directly
in HBM from a counter RNG keyed by (seed, global sample index), so every sharding sees the same data.

    --gpus 1, 2, 4, 8 : config 3 -- the configuration the metric is quoted on -- N = 1e7 samples in TOTAL, column-sharded over the
                        GPUs (strong scaling: `value` at N GPUs against N x `value` at one is the scaling efficiency)
    --gpus 8          : ALSO measures config 4 (N = 1e8 in total = 1.25e7 per GPU) after the headline run and reports it as the
                        `config4` object of the same JSON line (`--config4 0` skips it, `--config4 1` forces it at any count)

A step is ONE adaptive iteration (pymbar/mbar_solvers.py:575-640): the K x K Newton solve from the Hessian at the current f,
ONE fused sweep over the resident probability matrix that evaluates the gradients of both candidates (f_sci, f_nr) AND
accumulates, on the fp64 matrix cores, the Gram matrix of the candidate that is about to be accepted (the next iteration's
Hessian), the choice -- all device-resident -- with ONE ncclAllReduce per iteration when N > 1.  The timed region is ONE solver
call of exactly K iterations with the convergence test disabled, including the solver's build sweep (probability matrix,
gradient and Hessian at the start point).  From f = 0 this workload converges in 5 iterations, so most of the K timed steps
run at the fixed point; they cost the same ONE sweep as the steps before it as long as no speculation is rejected
(``separate_gram_sweeps_in_timed_region`` counts the rejected ones: each costs one more sweep, and none occurs here).  What
the steady-state rate hides is the build sweep, amortised over K steps instead of a real solve's 5: ``per_solve`` reports the
solve from f = 0 to tol 1e-12 -- iterations, wall clock, ms per iteration and the fraction of the fp64 matrix peak over the
WHOLE solve -- for min_sc_iter = 0 (BASELINE's adaptive configuration) and for the reference's default min_sc_iter = 2.
``value`` = solver iterations per second of the whole job (NOT multiplied by the number of GPUs).

Without a launcher (no WORLD_SIZE in the environment) ``--gpus N`` spawns its own N ranks, one per GPU, and fails (exit code 3)
when the box has fewer than N devices.

Ranks rendezvous through ``pymbar_amd.distributed.HostGroup`` (standard-library TCP on MASTER_ADDR / MASTER_PORT + 1);
the data path is RCCL inside libmbar_hip.so.  If RCCL cannot be initialised on every rank the run REFUSES to measure (exit
code 3, the RCCL error on stderr): a number over the host transport is not the metric.  ``--allow-host-allreduce`` lets such a
run go on as a diagnosis of the node -- its line then says ``"valid": false``, ``config.allreduce == "host-fallback"`` with the
RCCL error beside it (``config.rccl_error``), carries no config 4, and the process still exits with code 4.

At ``--gpus 1`` the line also carries ``config4_one_device`` (BASELINE.json config 4's WHOLE matrix, K=128, N=1e8, on the one
288 GB device: 102 GB of u + 102 GB of resident probabilities -- NOT the 8-GPU metric) and ``scaling_model``: the pieces of
one iteration measured on ONE GPU at the shard sizes of 1, 2, 4 and 8 ranks, and what they predict for the strong-scaling
curve -- a model, labelled as such, since no multi-GPU run can be made from here.
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
    """HBM bytes per launch of a kernel from the COMMITTED rocprofv3 PMC passes (profiles/*_pmc_fetch_write.json,
    newest round first), corrected as MI355X_MICROARCH.md prescribes (gfx950: FETCH_SIZE x 2).  Returns
    ``(bytes, source_file)`` or ``(None, None)`` if no profile of this exact workload is committed -- PMC collection needs
    its own rocprofv3 run, so this number is never measured inside the bench run itself."""
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
                b = (2.0 * v["FETCH_SIZE_KB_mean_per_launch"] + v.get("WRITE_SIZE_KB_mean_per_launch", 0.0)) * 1024.0
                return b, os.path.relpath(path, ROOT)
    return None, None


def measured_kernel_clock():
    """Shader clock the dominant kernel sustains (GHz), from the committed PMC pass (profiles/r*_pmc_fused_clock.json: GRBM_GUI_ACTIVE /
    8 XCDs / kernel duration): the matrix peak of the spec sheet assumes 2.4 GHz, the chip runs this kernel power-limited below it."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fused_clock.json")), reverse=True):
        try:
            d = json.load(open(path))
            return float(d["effective_clock_GHz"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def reference_build_host_baseline(K):
    """The UNMODIFIED reference (pymbar numpy path) timed on the build container by tools/time_reference.py; committed
    under profiles/ because /root/reference does not exist on the GPU box."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_timing.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("K") == K:
            d = dict(d)
            d["source"] = os.path.relpath(path, ROOT)
            return d
    return None


def reference_gpu_host_baseline(K, N_full):
    """The UNMODIFIED reference timed on the host of an MI355X box of this pool (tools/reference_on_gpu_box.sh: the reference
    tree travels there as an untracked tarball for that one call -- bench.py itself never reads it); committed under profiles/.
    At N = 1e7 the record holds the reference's own cold solve of config 3 (no extrapolation)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_timing_gpu_host.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("K") != K:
            continue
        rows = {int(r["N"]): r for r in d.get("rows", [])}
        exact = rows.get(int(N_full))
        if exact and "adaptive_iteration_s" in exact:
            sec, how = float(exact["adaptive_iteration_s"]), f"measured at N = {N_full} ({exact.get('timing')})"
        else:
            sec = float(d["adaptive_iteration_seconds_per_sample"]) * N_full
            how = f"scaled linearly to N = {N_full} from the largest N timed"
        rec = {
            "value": 1.0 / sec, "unit": "iter/s", "cores": d.get("cores"), "kind": "reference",
            "host": f"host of an MI355X box of this pool ({d.get('cpu_model')}, {d.get('cores')} logical cores): {os.path.relpath(path, ROOT)}",
            "sample": f"unmodified pymbar (python {d.get('python')}, numpy {d.get('numpy')}, scipy {d.get('scipy')}, BLAS threads "
                      f"{d.get('blas_threads')}; its logsumexp passes are single-threaded numpy), K={K}: {how}",
            "seconds_per_iteration": sec,
        }
        if exact and "full_solve_s" in exact:
            rec["wallclock_to_converge_s"] = float(exact["full_solve_s"])
            rec["iterations_to_converge"] = int(exact.get("full_solve_iterations", 0))
        return rec
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
    port = {
        "value": it_per_s_full,
        "unit": "iter/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "host": "this GPU box",
        "sample": f"one adaptive iteration of oracle/mbar_oracle.py (numpy {np.__version__}, scipy logsumexp, "
                  f"BLAS threads = all cores) on the first {n_sample} of {N_full} columns, K={K}: "
                  f"{t_iter:.2f} s measured (gradient alone {t_grad:.2f} s), scaled linearly by {N_full / n_sample:.2f}x",
        "seconds_per_iteration_extrapolated": t_iter * N_full / n_sample,
    }
    try:  # (threads the port's BLAS calls can use; its logsumexp passes are single-threaded numpy)
        from threadpoolctl import threadpool_info

        blas = [int(t.get("num_threads", 0)) for t in threadpool_info() if t.get("user_api") == "blas"]
        if blas:
            port["cores"] = max(blas)
            port["sample"] += f"; BLAS pool {max(blas)} threads of {os.cpu_count()} logical cores, element-wise passes single-threaded"
    except Exception:
        pass
    # `cpu_baseline` itself is what was timed LIVE on this host's cores: the oracle port.  The REFERENCE's own CPU path (unmodified
    # pymbar, numpy backend) cannot be read by bench.py; its timing on the host of an MI355X box of this pool (round 5) and on the
    # build container (tools/time_reference.py, committed under profiles/) ride along.
    ref_gpu_host = reference_gpu_host_baseline(K, N_full)
    if ref_gpu_host is not None:
        port["reference_on_gpu_host"] = ref_gpu_host
    ref = reference_build_host_baseline(K)
    if ref is None:
        return port
    sec = ref["adaptive_iteration_seconds_per_sample"] * N_full
    rows = ref.get("rows", [])
    biggest = max(rows, key=lambda r: r["N"]) if rows else {}
    port["reference_on_build_container"] = {
        "value": 1.0 / sec,
        "unit": "iter/s",
        "cores": ref.get("cores"),
        "kind": "reference",
        "host": f"build container ({ref.get('cores')} cores, {ref.get('host')}), NOT this GPU box: {ref['source']}",
        "sample": f"unmodified pymbar (python {ref.get('python')}, numpy {ref.get('numpy')}, scipy {ref.get('scipy')}), one adaptive "
                  f"iteration = mbar_hessian + lstsq + self_consistent_update + 2 mbar_gradient (mbar_solvers.py:581-607) on "
                  f"N = {biggest.get('N')} of the same harmonic ladder, K={K}: {biggest.get('adaptive_iteration_s', float('nan')):.1f} s "
                  f"(warm-up + best of 3), scaled linearly to N = {N_full}",
        "seconds_per_iteration_extrapolated": sec,
        "reference_timing": ref,
    }
    return port


def api_end_to_end(K, N_total, seed, dev, O_k, K_k, N_k):
    """What a user of ``MBAR(u_kn, N_k)`` sees: a HOST (K, N) array in, PCIe upload + NaN scan + default solver protocol
    + ``compute_free_energy_differences()`` (covariance sweep on the matrix cores), wall clock."""
    import pymbar_amd
    from pymbar_amd.device import DeviceMatrix

    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=seed, n_global0=0, N_local=N_total, device=dev) as gen:
        u_host = gen.to_host()  # pageable host memory, like any numpy array a user would pass
    t0 = time.perf_counter()
    mbar = pymbar_amd.MBAR(u_host, N_k, device=dev, copy=False)  # (no second 10 GB host copy: the array is not touched again)
    t_ctor = time.perf_counter() - t0
    t1 = time.perf_counter()
    r = mbar.compute_free_energy_differences()
    t_diff = time.perf_counter() - t1
    total = time.perf_counter() - t0
    ok = bool(np.all(np.isfinite(r["dDelta_f"])))
    stats = dict(getattr(mbar, "upload_stats", {}) or {})
    mbar.close()
    del mbar
    # the constructor as most callers use it: with the object's own host copy of the matrix (the reference's semantics)
    t2 = time.perf_counter()
    mbar = pymbar_amd.MBAR(u_host, N_k, device=dev)
    t_ctor_copy = time.perf_counter() - t2
    mbar.close()
    del mbar, u_host
    return {"api_end_to_end_s": total, "constructor_s": t_ctor, "compute_free_energy_differences_s": t_diff,
            "constructor_with_private_host_copy_s": t_ctor_copy, "host_bytes": 8.0 * K * N_total, "finite": ok, **stats}


def config2_object(dev, seed):
    """BASELINE.json config 2 (K=32, N=1e6, fp64): the device-resident self-consistent iteration (mbar_solvers.py:231-242 in a
    loop; the reference's stopping rule :627-631).  An iteration reads the matrix once: 8 K N bytes against HBM."""
    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix

    K, N = 32, 1_000_000
    O_k, K_k, N_k = ts.config3_params(K=K, N=N)
    with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=seed, n_global0=0, N_local=N, device=dev) as dm:
        dm.set_Nk(N_k)
        f0 = np.zeros(K)
        dm.solve_sci(f0, tol=1e-12, maxiter=64, check_convergence=False)  # warm-up: graph capture, first-touch
        iters = 512
        el = float("inf")
        for _ in range(3):  # (best of three: < 0.1 s of GPU time in total; the clock of a box that just ran config 3 is still settling)
            dm.device_synchronize()
            t0 = time.perf_counter()
            dm.solve_sci(f0, tol=1e-12, maxiter=iters, check_convergence=False)
            dm.device_synchronize()
            el = min(el, time.perf_counter() - t0)
        t1 = time.perf_counter()
        f_c, r_c = dm.solve_sci(f0, tol=1e-12)
        t_conv = time.perf_counter() - t1
        t2 = time.perf_counter()
        f_a, r_a = dm.solve_adaptive(f0, tol=1e-12, min_sc_iter=0)
        t_ad = time.perf_counter() - t2
        us = 1e6 * el / iters
        return {
            "workload": f"config2: harmonic ladder K={K}, N={N}, fp64, generated in HBM; pure self-consistent iteration, device-resident",
            "sci_iterations_per_s": iters / el, "us_per_iteration": us, "iterations_timed": iters, "timing": "best of 3 calls",
            "roofline": {"bound": "hbm", "achieved": 8.0 * K * N / (us * 1e-6) * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": 8.0 * K * N / (us * 1e-6) * 1e-9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": 8.0 * K * N,
                         "note": "whole iteration (sweep + update + launch gaps), not the sweep kernel alone"},
            "sci_iterations_to_converge": int(r_c["iterations"]), "sci_converged": bool(r_c["success"]),
            "sci_wallclock_to_converge_ms": 1e3 * t_conv,
            "adaptive_iterations_to_converge": int(r_a["iterations"]), "adaptive_wallclock_to_converge_ms": 1e3 * t_ad,
            "max_abs_difference_sci_vs_adaptive": float(np.max(np.abs(f_c - f_a))),
            "max_abs_error_vs_analytic_f": float(np.max(np.abs(f_a - ts.harmonic_free_energies(K_k)))),
        }


def config5_object(dev):
    """BASELINE.json config 5 (alchemical shape: K=40, N=95 000, two unsampled states): what a user of the class sees, host
    arrays in -- ``MBAR(u_kn, N_k)`` (upload + default protocol) + ``compute_free_energy_differences()`` (covariance sweep on the
    matrix cores + host eigh / pinv), steady state (median of 7 after one warm-up object)."""
    import pymbar_amd
    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix

    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    K, N = u_kn.shape
    ctor, diff = [], []
    for rep in range(8):
        t0 = time.perf_counter()
        m = pymbar_amd.MBAR(u_kn, N_k, device=dev)
        t1 = time.perf_counter()
        r = m.compute_free_energy_differences()
        t2 = time.perf_counter()
        m.close()
        if rep:
            ctor.append(t1 - t0)
            diff.append(t2 - t1)
    with DeviceMatrix.from_host(u_kn, device=dev) as dm:
        dm.set_Nk(N_k)
        dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
        solves = []
        for _ in range(7):
            dm.set_option("pcache", 0)
            t0 = time.perf_counter()
            f_a, r_a = dm.solve_adaptive(np.zeros(K), tol=1e-12, min_sc_iter=0)
            solves.append(time.perf_counter() - t0)
    return {
        "workload": f"config5: alchemical-shaped ladder K={K}, N={N} (states 7 and 23 unsampled), host arrays in",
        "constructor_ms": 1e3 * float(np.median(ctor)), "compute_free_energy_differences_ms": 1e3 * float(np.median(diff)),
        "class_journey_ms": 1e3 * float(np.median(np.add(ctor, diff))),
        "adaptive_solve_resident_us": 1e6 * float(np.median(solves)), "adaptive_iterations": int(r_a["iterations"]),
        "adaptive_us_per_iteration": 1e6 * float(np.median(solves)) / max(1, int(r_a["iterations"])),
        "finite": bool(np.all(np.isfinite(r["dDelta_f"]))),
    }


def config4_object(dev, K, args, rank, world, group):
    """BASELINE.json config 4 (K=128, N=1e8 in total, column-sharded over the ranks): the same adaptive iteration, a few forced steps
    in one cold solver call + one solve to tol 1e-12.  world == 1: the WHOLE matrix on the one device (102.4 GB of u + 102.4 GB of
    resident probabilities; rows of 1e8 samples take the kernels with 64-bit lane offsets) -- reported as `config4_one_device`,
    not the 8-GPU metric."""
    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix
    from pymbar_amd.distributed import attach_allreduce, shard_bounds

    N4 = 100_000_000
    a0, a1 = shard_bounds(N4, rank, world)
    O4, K4, Nk4 = ts.config3_params(K=K, N=N4)
    Nk4 = Nk4.copy()
    Nk4[-1] += N4 - int(Nk4.sum())
    t_gen = time.perf_counter()
    d4 = DeviceMatrix.harmonic(O4, K4, Nk4, seed=args.seed, n_global0=a0, N_local=a1 - a0, device=dev)
    d4.device_synchronize()
    t_gen = time.perf_counter() - t_gen
    try:
        for key, val in (("device_loop", args.device_loop), ("pmode", args.pmode), ("fused", args.fused),
                         ("pcache", 0), ("timing", 1), ("graph", 0)):
            d4.set_option(key, val)
        d4.set_Nk(Nk4)
        kind4 = attach_allreduce(d4, group) if world > 1 else "none"
        if kind4 not in ("none", "rccl"):
            return {"skipped": f"RCCL unavailable for the second communicator (transport would be '{kind4}')"}
        f0 = np.zeros(K)
        steps4 = max(3, min(args.steps, 8))
        d4.solve_adaptive(f0, tol=1e-12, maxiter=1, min_sc_iter=0, check_convergence=False)
        d4.timing_reset()
        if group is not None:
            group.barrier()
        d4.device_synchronize()
        t0 = time.perf_counter()
        _, r4 = d4.solve_adaptive(f0, tol=1e-12, maxiter=steps4, min_sc_iter=0, check_convergence=False)
        if group is not None:
            group.barrier()
        d4.device_synchronize()
        e4 = time.perf_counter() - t0
        tm4 = d4.timing()
        t1 = time.perf_counter()
        f4, c4 = d4.solve_adaptive(f0, tol=1e-12, maxiter=10000, min_sc_iter=0, check_convergence=True)
        d4.device_synchronize()
        tc4 = time.perf_counter() - t1
        if group is not None:
            tt = np.array([e4, tc4])
            group.allreduce(tt, "max")
            e4, tc4 = float(tt[0]), float(tt[1])
        fus4 = tm4.get("fused", (0.0, 0))
        bld4 = tm4.get("other", (0.0, 0))
        flop4 = float(a1 - a0) * K * (K + 1)
        fus_avg = fus4[0] / fus4[1] if fus4[1] else None
        out = {
            "workload": f"config4: harmonic ladder K={K}, N_total={N4} ({a1 - a0} per GPU), same adaptive iteration, {steps4} steps "
                        "in one cold solver call",
            "scaling_note": "ten times the samples of the headline workload: compare per GPU, not with `value`",
            "iterations_per_s": steps4 / e4, "ms_per_step": 1e3 * e4 / steps4, "allreduce": kind4,
            "fused_sweep_ms": fus_avg, "build_sweep_ms": bld4[0] / bld4[1] if bld4[1] else None,
            "fused_sweep_frac_of_matrix_peak": (flop4 / (fus_avg * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS) if fus_avg else None,
            "iterations_to_converge": int(c4["iterations"]), "converged": bool(c4["success"]), "wallclock_to_converge_s": tc4,
            "nr_iterations": int(c4["nr_iter"]), "sci_iterations": int(c4["sci_iter"]), "build_sweeps": int(c4.get("builds", -1)),
            "max_abs_error_vs_analytic_f": float(np.max(np.abs(f4 - ts.harmonic_free_energies(K4)))),
            "generate_in_hbm_s": t_gen,
        }
        if world == 1:
            out["workload"] = (f"config4's WHOLE matrix on ONE device: harmonic ladder K={K}, N={N4}, fp64, generated in HBM "
                               f"({8e-9 * K * N4:.1f} GB of u + as much of resident probabilities), {steps4} adaptive steps in one cold solver call")
            out["scaling_note"] = ("NOT the 8-GPU metric of BASELINE.json config 4: one device sweeps all 1e8 samples (ten times the "
                                   "headline workload per step); per-GPU shard of the 8-GPU run = 1.25e7 samples")
            if fus_avg:
                out["roofline"] = {"kernel": "k_fused<8> with 64-bit lane offsets (row pitch 1e8 samples)", "bound": "mfma",
                                   "achieved": flop4 / (fus_avg * 1e-3) * 1e-12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": flop4 / (fus_avg * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                                   "avg_launch_ms": fus_avg, "launches": fus4[1], "algorithmic_flop_per_launch": flop4,
                                   "hbm_GBps_of_one_read": 8.0 * K * N4 / (fus_avg * 1e-3) * 1e-9}
        return out
    finally:
        d4.close()


def scaling_model_object(dev, K, N_total, seed, steps, warmup):
    """What ONE GPU can say about the strong-scaling curve of the metric (config 3 in total, column-sharded over G ranks): for
    G = 1, 2, 4, 8 a problem of the SHARD's shape (K states, N_total / G samples of the same ladder, well-posed on its own so the
    loop takes its normal path) runs `steps` forced adaptive iterations twice -- timing level 1 for the wall clock and the sweep
    (what `value` is made of), timing level 3 for the split {reduction levels, in-stream all-reduce, Newton solve + selection} --
    with a single-rank RCCL communicator attached, so the all-reduce of the iteration (2K + 2 + 36 x 256 doubles) IS enqueued
    on the stream and its nranks = 1 cost is in the tail.  Not measurable from here: the latency of that all-reduce over xGMI
    between real ranks, and the wait for the slowest rank's sweep -- the model quotes the prediction for several assumed values."""
    import ctypes as C

    from pymbar_amd import _lib
    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix
    from pymbar_amd.distributed import shard_bounds

    rows = {}
    rccl = "rccl (nranks = 1)"
    for G in (1, 2, 4, 8):
        a0, a1 = shard_bounds(N_total, 0, G)
        n = a1 - a0
        O_k, K_k, N_k = ts.config3_params(K=K, N=n)
        N_k = N_k.copy()
        N_k[-1] += n - int(N_k.sum())
        with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=seed, n_global0=0, N_local=n, device=dev) as dm:
            for key, val in (("pcache", 0), ("graph", 0), ("timing", 1)):
                dm.set_option(key, val)
            try:
                buf = C.create_string_buffer(128)
                _lib.check(_lib.load_library().mbar_comm_unique_id(buf))
                dm.comm_init_rccl(bytes(buf.raw), 0, 1)
            except Exception as exc:  # (no librccl: the tail is then measured without the collective's enqueue)
                rccl = f"unavailable ({exc})"
            dm.set_Nk(N_k)
            f0 = np.zeros(K)
            dm.solve_adaptive(f0, tol=1e-12, maxiter=max(1, warmup), min_sc_iter=0, check_convergence=False)
            best = None
            for _ in range(2):
                dm.timing_reset()
                dm.device_synchronize()
                t0 = time.perf_counter()
                _, r = dm.solve_adaptive(f0, tol=1e-12, maxiter=steps, min_sc_iter=0, check_convergence=False)
                dm.device_synchronize()
                el = time.perf_counter() - t0
                tm = dm.timing()
                if best is None or el < best[0]:
                    best = (el, tm, r)
            el, tm, r = best
            dm.set_option("timing", 3)
            dm.timing_reset()
            dm.solve_adaptive(f0, tol=1e-12, maxiter=steps, min_sc_iter=0, check_convergence=False)
            dm.device_synchronize()
            t3 = dm.timing()
            if dm.allreduce_kind == "rccl":
                dm.comm_destroy()
        fus_ms, fus_n = tm.get("fused", (0.0, 0))
        bld_ms, bld_n = tm.get("other", (0.0, 0))
        sweep = fus_ms / max(1, fus_n)
        ms_step = 1e3 * el / steps
        tail = ms_step - sweep - bld_ms / steps

        def per_it(name):
            ms, cnt = t3.get(name, (0.0, 0))
            return ms / steps if cnt else None
        rows[str(G)] = {
            "N_per_rank": n, "sweep_ms": sweep, "build_sweep_ms": bld_ms / max(1, bld_n), "ms_per_step_one_rank": ms_step,
            "tail_ms": tail,
            "tail_split_ms": {"reduce_levels": per_it("reduce"), "all_reduce_nranks_1": per_it("comm"), "newton_and_select": per_it("newton"),
                              "note": "event pairs of timing level 3 (their marker packets lengthen the gaps: the sum exceeds tail_ms)"},
            "sweep_frac_of_matrix_peak": float(n) * K * (K + 1) / (sweep * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS if sweep > 0 else None,
            "separate_gram_sweeps": int(r.get("gram_sweeps", -1)),
        }
    base = rows["1"]["ms_per_step_one_rank"]
    assumed_us = (0.0, 15.0, 30.0, 60.0)
    for G in (1, 2, 4, 8):
        row = rows[str(G)]
        row["predicted"] = {}
        for L in assumed_us if G > 1 else (0.0,):
            ms = row["ms_per_step_one_rank"] + L * 1e-3
            row["predicted"][f"xgmi_allreduce_plus_{L:g}us"] = {"it_per_s": 1e3 / ms, "efficiency": base / (G * ms)}
        row["predicted_it_per_s"] = row["predicted"]["xgmi_allreduce_plus_30us" if G > 1 else "xgmi_allreduce_plus_0us"]["it_per_s"]
        row["predicted_efficiency"] = row["predicted"]["xgmi_allreduce_plus_30us" if G > 1 else "xgmi_allreduce_plus_0us"]["efficiency"]
    return {
        "kind": "model -- no multi-GPU run",
        "what": f"config 3 in total (K={K}, N_total={N_total}) over G ranks; every row is ONE GPU running a problem of the shard's shape "
                f"({steps} forced iterations in one cold solver call, build sweep included, like `value`)",
        "collective": f"ONE in-stream all-reduce per iteration of {2 * K + 2 + (K // 16) * (K // 16 + 1) // 2 * 256} doubles; measured here: {rccl}",
        "unknown": "the latency of that all-reduce (~76 KB) over xGMI between real ranks, and the wait for the slowest rank's sweep "
                   "(boxes of this pool differ by +-3 %); `predicted_*` assume 30 us on top of the single-rank enqueue, `predicted` lists 0 / 15 / 30 / 60 us",
        "ranks": rows,
    }


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as a launcher would) and wait.  Returns the exit code: 3 if the box has fewer than N GPUs, the
    first non-zero code of a rank otherwise (the remaining ranks are then stopped -- they would wait for the dead one)."""
    import socket
    import subprocess

    from pymbar_amd import _lib

    ndev = _lib.device_count()
    if ndev < n and "--oversubscribe" not in sys.argv:
        print(f"bench.py: --gpus {n} but only {ndev} GPU(s) visible; refusing to measure fewer ranks than asked for", file=sys.stderr)
        return 3
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    secret = os.urandom(16).hex()  # (handshake token of the ranks' TCP rendezvous)
    for r in range(n):
        env = dict(os.environ, MBAR_RDZV_SECRET=secret, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0:
                    rc = code
    for p in live:  # (a rank failed: its peers would wait in the rendezvous or in a collective)
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--n-total", type=int, default=0,
                    help="samples in total (default: 1e7 = config 3 for 1, 2, 4 GPUs; 1e8 = config 4 for 8 GPUs)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1_600_000,
                    help="columns for the CPU baseline: ~10-15 s of CPU work on the GPU host (0 = skip)")
    ap.add_argument("--api-e2e", type=int, default=1, help="also time MBAR(u_kn_host, N_k) end to end (1 GPU only; 0 = skip)")
    ap.add_argument("--small-configs", type=int, default=1,
                    help="also measure BASELINE.json configs 2 and 5 (1 GPU only, < 1 s of GPU time together; 0 = skip)")
    ap.add_argument("--device-loop", type=int, default=1, help="0 = host-driven adaptive loop (A/B)")
    ap.add_argument("--pmode", type=int, default=1,
                    help="1 = the device-resident loop sweeps the resident probability matrix P = exp(a0 - u - logden(a0)) "
                         "(no exponentials after the build); 0 = every sweep recomputes its exponentials from u (A/B)")
    ap.add_argument("--fused", type=int, default=1,
                    help="1 = ONE sweep per iteration in P mode: the candidate sweep also accumulates the Gram matrix of the "
                         "Newton-Raphson candidate (the separate Gram sweep runs only when that candidate is rejected); 0 = two sweeps (A/B)")
    ap.add_argument("--allow-host-allreduce", action="store_true",
                    help="when RCCL cannot be initialised on every rank: go on over the host transport instead of refusing (exit code 3); "
                         "the line then says \"valid\": false and the process exits with code 4 -- a diagnosis, never the metric")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing only: let --gpus N spawn more ranks than the box has GPUs (ranks share devices; RCCL refuses that, so "
                         "this needs --allow-host-allreduce and proves nothing but that the multi-rank bench path executes)")
    ap.add_argument("--config4-one-device", type=int, default=-1,
                    help="--gpus 1: also measure config 4's WHOLE matrix (K=128, N=1e8) on the one device as `config4_one_device`: "
                         "-1 = when the device has 230 GB or more (default), 0 = never, 1 = always")
    ap.add_argument("--scaling-model", type=int, default=1,
                    help="--gpus 1: also measure the pieces of an iteration at the shard sizes of 1/2/4/8 ranks and emit `scaling_model` (0 = skip)")
    ap.add_argument("--config4", type=int, default=-1,
                    help="also measure config 4 (K=128, N=1e8 in total, sharded) after the headline run: -1 = only with 8 or more "
                         "GPUs (default), 0 = never, 1 = always (needs 205 GB on a single GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    # The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a version banner from C stdio when a
    # communicator is created, flushed at exit -- i.e. AFTER the line): file descriptor 1 is pointed at stderr for everything but
    # the line itself, which goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from pymbar_amd import _lib
    from pymbar_amd import testsystems as ts
    from pymbar_amd.device import DeviceMatrix, device_info
    from pymbar_amd.distributed import HostGroup, attach_allreduce, shard_bounds

    group = HostGroup.from_env() if world > 1 else None

    K = args.K
    N_total = args.n_total if args.n_total > 0 else 10_000_000  # the metric's workload at EVERY GPU count (strong scaling)
    config_name = "config4" if N_total == 100_000_000 else ("config3" if N_total == 10_000_000 else "custom")
    n0, n1 = shard_bounds(N_total, rank, world)
    n_loc = n1 - n0
    O_k, K_k, N_k = ts.config3_params(K=K, N=N_total)
    N_k = N_k.copy()
    N_k[-1] += N_total - int(N_k.sum())  # keep sum(N_k) == N_total when K does not divide it

    ndev = max(1, _lib.device_count())
    dev = local_rank % ndev  # one rank per GPU under the launcher; wraps only when ranks outnumber devices (testing)
    info = device_info(dev)
    dm = DeviceMatrix.harmonic(O_k, K_k, N_k, seed=args.seed, n_global0=n0, N_local=n_loc, device=dev)
    dm.set_option("device_loop", args.device_loop)
    dm.set_option("pmode", args.pmode)
    dm.set_option("fused", args.fused)
    dm.set_option("pcache", 0)  # every solver call of the timed region builds its own probability matrix (no warm starts)
    # HIP-event pairs around the sweeps (off by default in the library); with several ranks also around the reduction, the
    # all-reduce and the Newton / selection launches, so that a multi-GPU line explains where its iteration goes
    dm.set_option("timing", 3 if world > 1 else 1)
    dm.set_option("graph", 0)  # eager launches: per-kernel HIP-event timers inside the timed region (the GPU queue never
    #                            runs dry at this size: an iteration is ~5 ms of kernels against ~0.1 ms of enqueueing)
    dm.set_Nk(N_k)
    allreduce = "none"
    rccl_error = None
    if world > 1:
        allreduce = attach_allreduce(dm, group)
        if allreduce != "rccl":
            # RCCL could not be initialised on every rank.  Default: refuse (exit code 3) -- a scaling harness that reads `value` or
            # the exit code must not record a host-transport number as the RCCL result.  With --allow-host-allreduce the run goes on
            # over the host transport (TCP all-reduce, host-driven loop) as a diagnosis: "valid": false, "allreduce":
            # "host-fallback" plus the RCCL error, no config 4, exit code 4.
            rccl_error = getattr(dm, "rccl_error", None) or "unknown"
            allreduce = "host-fallback"
            if not args.allow_host_allreduce:
                # (every rank takes this branch: attach_allreduce's outcome is agreed across the ranks)
                if rank == 0:
                    print(f"bench.py: RCCL could not be initialised on every rank ({rccl_error}); refusing to measure the host "
                          "transport as if it were the metric (--allow-host-allreduce runs it as a diagnosis)", file=sys.stderr)
                dm.close()
                group.barrier()
                group.close()
                sys.exit(3)
            if rank == 0:
                print(f"bench.py: RCCL could not be initialised on every rank ({rccl_error}); measuring the host fallback "
                      "(\"valid\": false)", file=sys.stderr)

    def barrier_sync():
        if group is not None:
            group.barrier()
        dm.device_synchronize()

    f0 = np.zeros(K)
    # ---- warm-up (untimed) ----
    if args.warmup > 0:
        dm.solve_adaptive(f0, tol=1e-12, maxiter=args.warmup, min_sc_iter=0, check_convergence=False)
    # ---- timed: exactly `steps` adaptive iterations ----
    dm.timing_reset()
    barrier_sync()
    t0 = time.perf_counter()
    f_end, res = dm.solve_adaptive(f0, tol=1e-12, maxiter=args.steps, min_sc_iter=0, check_convergence=False)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    timing = dm.timing()
    if group is not None:
        t = np.array([elapsed])
        group.allreduce(t, "max")
        elapsed = float(t[0])
    assert res["iterations"] == args.steps, res

    # ---- wall-clock to converge from f = 0 (reported, not the headline value): whole solves, build sweep included ----
    def timed_solve(min_sc_iter, warm=False):
        # cold: the solver call builds its resident probability matrix (what a first solve on a matrix costs); warm: the matrix
        # of the previous call is re-used (bootstrap replicates, protocol stages: one fused sweep instead of the build sweep)
        dm.set_option("pcache", 1 if warm else 0)
        barrier_sync()
        t0 = time.perf_counter()
        f_c, r_c = dm.solve_adaptive(f0, tol=1e-12, maxiter=10000, min_sc_iter=min_sc_iter, check_convergence=True)
        barrier_sync()
        t = time.perf_counter() - t0
        if group is not None:
            tt = np.array([t])
            group.allreduce(tt, "max")
            t = float(tt[0])
        return f_c, r_c, t

    f_conv, conv, t_conv = timed_solve(0)
    _, conv2, t_conv2 = timed_solve(2)   # the reference's default for adaptive() (mbar_solvers.py:545; "robust" protocol)
    dm.set_option("pcache", 1)
    dm.solve_adaptive(f0, tol=1e-12, maxiter=10000, min_sc_iter=0)  # (leaves its probability matrix behind)
    _, conv_w, t_conv_w = timed_solve(0, warm=True)
    err_analytic = float(np.max(np.abs(f_conv - ts.harmonic_free_energies(K_k))))

    mfma_peak = dm.mfma_f64_peak() if rank == 0 else None

    # ---- config 4 (K = 128, N = 1e8 in total, sharded): BASELINE.json's eight-GPU configuration, next to the headline run ----
    config4 = None
    if (args.config4 == 1 or (args.config4 < 0 and world >= 8)) and N_total != 100_000_000 and allreduce != "host-fallback":
        config4 = config4_object(dev, K, args, rank, world, group)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        def factory(n):
            return DeviceMatrix.harmonic(O_k, K_k, N_k, seed=args.seed, n_global0=0, N_local=n, device=dev)
        cpu = cpu_baseline(factory, K, N_total, min(args.cpu_sample, n_loc), args.seed)

    e2e = None
    if rank == 0 and world == 1 and args.api_e2e:
        dm.close()  # the end-to-end run owns its own device copy
        dm = None
        e2e = api_end_to_end(K, N_total, args.seed, dev, O_k, K_k, N_k)

    config2 = config5 = None
    if rank == 0 and world == 1 and args.small_configs:
        if dm is not None:
            dm.close()
            dm = None
        config2 = config2_object(dev, args.seed)
        config5 = config5_object(dev)

    scaling_model = config4_one = None
    if rank == 0 and world == 1 and N_total == 10_000_000 and args.device_loop and args.pmode and args.fused:
        if args.scaling_model:
            if dm is not None:
                dm.close()
                dm = None
            scaling_model = scaling_model_object(dev, K, N_total, args.seed, args.steps, args.warmup)
        want4 = args.config4_one_device
        if want4 == 1 or (want4 < 0 and info["total_mem_bytes"] >= 230 * (1 << 30)):
            if dm is not None:
                dm.close()
                dm = None
            _lib.load_library().mbar_cache_trim()  # (205 GB of 288: the parked blocks of the runs above go back to the driver first)
            try:
                config4_one = config4_object(dev, K, args, 0, 1, None)
            except Exception as exc:  # (a shared or smaller device: the headline line must still be printed)
                config4_one = {"skipped": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        it_per_s = args.steps / elapsed
        ms_step = 1e3 * elapsed / args.steps
        gram_ms, gram_n = timing["gram"]
        lse_ms, lse_n = timing["lse"]
        fus_ms, fus_n = timing.get("fused", (0.0, 0))
        # (the device-resident loop, hence P mode and the fused sweep, needs RCCL or a single rank; with the host transport the
        # host-driven loop and its classic sweeps run)
        pm = bool(args.pmode and args.device_loop and allreduce in ("none", "rccl"))
        fused = bool(pm and args.fused and fus_n > 0)
        flops = float(n_loc) * K * (K + 1)           # symmetric Gram: K(K+1)/2 entries x 2 flop x N (this rank's shard)
        bytes_pass = 8.0 * K * n_loc                 # one read of the shard per sweep
        sweep_flops = 8.0 * K * n_loc                # candidate sweep in P mode: 2 candidates x (dot + accumulation) x 2 flop
        if fused:
            # ONE kernel per iteration: candidates' normalisers / per-state sums + the Gram matrix of the Newton-Raphson
            # candidate.  The separate Gram sweep runs only for rejected speculations (and once at the start); its launches
            # are no-ops otherwise, so its timer average is not a sweep time -- the count of real sweeps comes from the solver.
            dom_avg = fus_ms / fus_n
            dom_flops = flops + sweep_flops
            tr_g, src_g = pmc_traffic("k_fused<8", K, n_loc)
            # `achieved` / `frac` count the MATRIX flop only (N K (K+1): what the 16x16x4 peak is a peak of); the candidate sweep's
            # 8 K N vector flop (VALU FMAs and 4x4x4 matrix instructions of the normalisers) ride in the same kernel and are
            # reported next to it, not added to the numerator
            roof = {
                "kernel": "k_fused<8> (one pass over the resident probability matrix: 2 candidates' normalisers and per-state "
                          "sums + fp64 MFMA Gram matrix of the candidate about to be accepted)",
                "bound": "mfma", "achieved": flops / (dom_avg * 1e-3) * 1e-12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "traffic": tr_g, "traffic_source": f"committed PMC pass {src_g} (not collected in this run)" if src_g else None,
                "avg_launch_ms": dom_avg, "launches": fus_n, "algorithmic_flop_per_launch": flops,
                "vector_flop_per_launch_not_counted": sweep_flops,
                "frac_incl_vector": dom_flops / (dom_avg * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
            }
            roof["frac"] = roof["achieved"] / FP64_MFMA_PEAK_TFLOPS
            hbm_gbs = bytes_pass / (dom_avg * 1e-3) * 1e-9
            roof2 = {
                "kernel": "k_fused<8> seen from HBM (it is not bound by it)",
                "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                "traffic": tr_g, "avg_launch_ms": dom_avg, "launches": fus_n, "algorithmic_bytes_per_launch": bytes_pass,
            }
            # the solver call's build sweep (k_build_gram: P = exp(a0 - u - logden), gradient and Gram matrix at the start
            # point -- the work of "iteration 0") runs once per call inside the timed region: a sweep, reported on its own
            build_ms, build_n = timing.get("other", (0.0, 0))
            outside = ms_step - dom_avg - (gram_ms + build_ms) / args.steps
            extra = {"separate_gram_sweeps_in_timed_region": int(res.get("gram_sweeps", -1)),
                     "separate_gram_sweep_ms_total": gram_ms, "separate_gram_launches_incl_noops": gram_n,
                     "build_sweep_ms_total": build_ms, "build_sweeps_in_timed_region": build_n,
                     "ms_per_step_build_sweep_share": build_ms / args.steps}
        else:
            gram_avg = gram_ms / max(1, gram_n)
            lse_avg = lse_ms / max(1, lse_n)
            achieved_tf = flops / (gram_avg * 1e-3) * 1e-12 if gram_avg > 0 else 0.0
            achieved_gbs = bytes_pass / (lse_avg * 1e-3) * 1e-9 if lse_avg > 0 else 0.0
            tr_g, src_g = pmc_traffic("k_gram<", K, n_loc)
            tr_l, src_l = pmc_traffic("k_psweep<8, 2" if pm else "k_lse<8, 2", K, n_loc)
            roof = {
                "kernel": ("k_gram<8,8> one wave per SIMD" + (", operands P / s" if pm else ", operands by table exp") + " (fp64 MFMA W^T W)") if K == 128 else "k_gram",
                "bound": "mfma", "achieved": achieved_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / FP64_MFMA_PEAK_TFLOPS, "traffic": tr_g,
                "traffic_source": f"committed PMC pass {src_g} (not collected in this run)" if src_g else None,
                "avg_launch_ms": gram_avg, "launches": gram_n, "algorithmic_flop_per_launch": flops,
            }
            roof2 = {
                "kernel": ("k_psweep<8,2> (normalisers + per-state sums of 2 candidates from the resident probability matrix, no exp)"
                           if pm else "k_lse<8,2> (log-sum-exp + per-state sums, 2 candidates per sweep, one exp per element)"),
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": tr_l,
                "traffic_source": f"committed PMC pass {src_l} (not collected in this run)" if src_l else None,
                "avg_launch_ms": lse_avg, "launches": lse_n, "algorithmic_bytes_per_launch": bytes_pass,
            }
            outside = ms_step - gram_avg - lse_avg
            extra = {}
        roof["measured_mfma_f64_peak_tflops"] = mfma_peak
        ghz, ghz_src = measured_kernel_clock()
        if ghz and roof.get("bound") == "mfma":
            # (the share of the gap to the spec-sheet peak that is the power limit, not the kernel: 78.6 TFLOP/s is 2.4 GHz)
            roof["frac_at_measured_clock"] = roof["achieved"] / (FP64_MFMA_PEAK_TFLOPS * ghz / 2.4)
            roof["measured_clock_GHz"] = ghz
            roof["measured_clock_source"] = ghz_src
        if fused:
            sweeps = "ONE fused sweep per iteration over the resident probability matrix (built once per solver call, inside the timed region)"
            what = "fused sweep (2-candidate gradients + MFMA Gram of the Newton-Raphson candidate) + K x K Newton solve"
        elif pm:
            sweeps = "resident probability matrix (built once per solver call, inside the timed region), two sweeps per iteration"
            what = "MFMA Gram sweep + K x K Newton solve + 2-candidate gradient sweep"
        else:
            sweeps = "exponentials recomputed from u in every sweep, two sweeps per iteration"
            what = "MFMA Gram sweep + K x K Newton solve + 2-candidate gradient sweep"
        out = {
            "metric": "mbar_adaptive_iterations_per_sec",
            "value": it_per_s,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{config_name}: harmonic ladder K={K}, N_total={N_total} ({n_loc} per GPU), adaptive NR/SCI "
                            f"iteration = {what}, device-resident, fp64, generated in HBM",
                "K": K, "N_per_gpu": n_loc, "N_total": N_total, "parallelism": f"N-sharded x{world}",
                "allreduce": allreduce, **({"rccl_error": rccl_error} if allreduce == "host-fallback" else {}), "device": info["name"],
                "adaptive_loop": "device-resident" if (args.device_loop and allreduce in ("none", "rccl")) else "host-driven",
                "sweeps": sweeps,
            },
            "roofline": roof,
            "roofline_lse": roof2,
            "ms_per_step_outside_the_sweeps": outside,
            "cpu_baseline": cpu,
            "wallclock_to_converge_s": t_conv,
            "iterations_to_converge": int(conv["iterations"]),
            "converged": bool(conv["success"]),
            "nr_iterations": int(conv["nr_iter"]), "sci_iterations": int(conv["sci_iter"]),
            "separate_gram_sweeps_to_converge": int(conv.get("gram_sweeps", -1)),
            # whole solves from f = 0 to tol 1e-12 (build sweep included): `frac` = one Hessian's matrix flop per iteration
            # over the wall clock of the solve, against the fp64 matrix peak
            "per_solve": {
                name: {"min_sc_iter": msc, "iterations": int(r["iterations"]), "nr_iterations": int(r["nr_iter"]),
                       "sci_iterations": int(r["sci_iter"]), "converged": bool(r["success"]), "wallclock_ms": 1e3 * t,
                       "ms_per_iteration": 1e3 * t / max(1, int(r["iterations"])),
                       "iterations_per_s": int(r["iterations"]) / t,
                       "separate_gram_sweeps": int(r.get("gram_sweeps", -1)),
                       # (the last iteration's sweep when both candidates already met the stop test: no Hessian accumulated, none needed)
                       "sweeps_without_gram_matrix": int(r.get("light_sweeps", -1)),
                       "build_sweeps": int(r.get("builds", -1)), "warm_starts": int(r.get("warm_starts", -1)),
                       "frac": int(r["iterations"]) * flops * world / t * 1e-12 / (FP64_MFMA_PEAK_TFLOPS * world)}
                for name, msc, r, t in (("adaptive_min_sc_iter_0", 0, conv, t_conv), ("adaptive_min_sc_iter_2", 2, conv2, t_conv2),
                                        ("adaptive_min_sc_iter_0_warm_start", 0, conv_w, t_conv_w))},
            "max_abs_error_vs_analytic_f": err_analytic,
            "gnorm_at_solution": float(conv["gnorm"]),
            "api_end_to_end": e2e,
            "config2": config2,
            "config4": config4,
            "config4_one_device": config4_one,
            "config5": config5,
            "scaling_model": scaling_model,
        }
        if allreduce == "host-fallback":
            out["valid"] = False
            out["invalid_because"] = "RCCL could not be initialised on every rank; measured over the host transport (diagnosis only)"
        out.update(extra)
        if world > 1:
            def per_launch(name):
                ms, n = timing.get(name, (0.0, 0))
                return {"ms_per_iteration": ms / args.steps, "launches": n} if n else None
            out["iteration_split_rank0"] = {
                "note": "HIP-event pairs on rank 0's stream around each section of the device-resident iteration (timing level 3); "
                        "the all-reduce interval includes waiting for the slowest rank's sweep",
                "sweep": per_launch("fused") or per_launch("lse"), "separate_gram_sweep": per_launch("gram"),
                "build_sweep_once_per_call": per_launch("other"),
                "reduce": per_launch("reduce"), "all_reduce": per_launch("comm"), "newton_and_select": per_launch("newton"),
                "all_reduce_doubles": int(2 * K + 2 + (K // 16) * (K // 16 + 1) // 2 * 256) if fused else None,
            }
        if cpu is not None:
            out["speedup_vs_cpu_baseline"] = it_per_s / cpu["value"]
            if "reference_on_gpu_host" in cpu:
                rg = cpu["reference_on_gpu_host"]
                out["speedup_vs_reference_on_gpu_host"] = it_per_s / rg["value"]
                if "wallclock_to_converge_s" in rg:
                    out["wallclock_to_converge_speedup_vs_reference_on_gpu_host"] = rg["wallclock_to_converge_s"] / t_conv
            if "reference_on_build_container" in cpu:
                out["speedup_vs_reference_on_build_container"] = it_per_s / cpu["reference_on_build_container"]["value"]
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dm is not None:
        dm.close()
    if group is not None:
        group.barrier()
        group.close()
    if allreduce == "host-fallback":
        sys.exit(4)  # (a harness that reads the exit code does not take this line for the RCCL number)


if __name__ == "__main__":
    main()
