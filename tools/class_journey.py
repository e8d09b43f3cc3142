#!/usr/bin/env python
"""One pass of what a user does with the class at config-3 size (K=128, N=1e7, a 10 GB host matrix), twice in one process:
construction with three bootstrap replicates, free energy differences (asymptotic and bootstrap), overlap, expectations,
perturbed free energies, entropy / enthalpy, bootstrapped expectations.  Wall time per call; the first pass pays the first
allocation of the 20 GB augmented matrix, the second runs out of the device block cache."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import pymbar_amd
from pymbar_amd import testsystems as ts
K, N = 128, 10_000_000
O_k, K_k, N_k = ts.config3_params(K=K, N=N)
t0 = time.perf_counter()
x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
print(f"host generation {time.perf_counter()-t0:.1f} s", flush=True)
def step(name, fn):
    t0 = time.perf_counter(); r = fn(); print(f"{name}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True); return r
for rep in range(2):
    m = step("MBAR(u_kn, N_k, n_bootstraps=3)", lambda: pymbar_amd.MBAR(u_kn, N_k, n_bootstraps=3))
    r = step("compute_free_energy_differences", lambda: m.compute_free_energy_differences())
    step("compute_free_energy_differences(bootstrap)", lambda: m.compute_free_energy_differences(uncertainty_method="bootstrap"))
    step("compute_overlap", lambda: m.compute_overlap())
    e = step("compute_expectations(x_n)", lambda: m.compute_expectations(x_n))
    step("compute_expectations(x_n) again", lambda: m.compute_expectations(x_n))
    step("compute_perturbed_free_energies(2 new states)", lambda: m.compute_perturbed_free_energies(u_kn[:2]))
    step("compute_entropy_and_enthalpy", lambda: m.compute_entropy_and_enthalpy())
    step("compute_expectations(x_n, bootstrap)", lambda: m.compute_expectations(x_n, uncertainty_method="bootstrap"))
    print("max |Delta_f[0] - analytic|", float(np.max(np.abs(r["Delta_f"][0] - (ts.harmonic_free_energies(K_k) - ts.harmonic_free_energies(K_k)[0])))), "  <x> state 0 / 127:", e["mu"][0], e["mu"][-1], flush=True)
    step("close", lambda: m.close()); del m
