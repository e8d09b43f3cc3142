"""PCIe-inclusive figures for the host-buffer boundary (`mbar_ctx_upload_u`): upload rate of a host (K, N) matrix and
the end-to-end time of `solve_mbar_once(u_kn_host, ...)` (upload + NaN scan + adaptive solve), next to the resident
solve.  The headline `bench.py` value starts with the matrix in HBM; this is the number for callers that hand over
numpy arrays.   Usage: python tools/bench_upload.py [K N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pymbar_amd import mbar_solvers as ms, testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
O_k, K_k, N_k = ts.config3_params(K=K, N=N)
with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as gen:   # same data as the bench workload, brought to the host
    u = gen.to_host()
gb = u.nbytes / 1e9
for rep in range(2):
    t0 = time.perf_counter()
    dm = DeviceMatrix.from_host(u)
    dm.synchronize()
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    f, res = ms.solve_mbar_once(dm, N_k, np.zeros(K), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
    t_solve = time.perf_counter() - t0
    dm.close()
    t0 = time.perf_counter()
    f2, res2 = ms.solve_mbar_once(u, N_k, np.zeros(K), method="adaptive", tol=1e-12, options=dict(min_sc_iter=0))
    t_all = time.perf_counter() - t0
    print(f"K={K} N={N} ({gb:.2f} GB): upload {t_up*1e3:.1f} ms = {gb/t_up:.1f} GB/s (pageable host memory); resident solve "
          f"{t_solve*1e3:.1f} ms ({res['iterations']} iterations); host-array solve_mbar_once {t_all*1e3:.1f} ms "
          f"= {res2['iterations']/t_all:.1f} it/s PCIe-inclusive vs {res['iterations']/t_solve:.1f} it/s resident")
    assert np.allclose(f, f2, atol=1e-12)
