#!/usr/bin/env python
"""INTEGRATION.md section 2, literally, on an MI355X box and at scale: the reference's UNCHANGED ``pymbar.MBAR`` class
(its only calls into the solver module: mbar.py:413,437,455,910) with the one binding

    pymbar.mbar.mbar_solvers = pymbar_amd.mbar_solvers

on the REAL device, K = 128, N = 1e6 drawn by the reference's OWN sampler (HarmonicOscillatorsTestCase, seed 0),
compared with the committed answer of the unmodified reference for this very matrix (tests/golden/scale_K128_N1e6.npz,
made by tests/golden/make_golden_scale.py k128 on the build container: 10 min of numpy there).

Needs the reference tree on PYTHONPATH (MBAR_REFERENCE_TREE; tools/reference_on_gpu_box.sh stages it).  Prints one JSON line.
This is a measurement / evidence tool, not part of the test suite or of bench.py (neither may read the reference tree)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import pymbar
    import pymbar.mbar
    from pymbar.testsystems import harmonic_oscillators

    ref_tree = os.path.realpath(os.environ.get("MBAR_REFERENCE_TREE", "/root/reference"))
    assert os.path.realpath(pymbar.__file__).startswith(ref_tree), pymbar.__file__
    import pymbar_amd.mbar_solvers as hip
    from pymbar_amd import _lib, device

    _lib.require_device()
    K, N = 128, 1_000_000
    O_k, K_k = np.linspace(0.0, 4.0, K), np.linspace(1.0, 3.0, K)
    N_k = np.full(K, N // K, dtype=np.int64)
    tc = harmonic_oscillators.HarmonicOscillatorsTestCase(O_k, K_k)
    x_n, u_kn, _, s_n = tc.sample(N_k, mode="u_kn", seed=0)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "scale_K128_N1e6.npz"))
    assert np.array_equal(gold["N_k"], N_k)

    ref_solvers = pymbar.mbar.mbar_solvers
    assert os.path.realpath(ref_solvers.__file__).startswith(ref_tree)
    pymbar.mbar.mbar_solvers = hip  # <- the whole integration
    try:
        t0 = time.perf_counter()
        mbar = pymbar.MBAR(u_kn, N_k)
        t_ctor = time.perf_counter() - t0
        assert type(mbar).__module__ == "pymbar.mbar"
        t0 = time.perf_counter()
        res = mbar.compute_free_energy_differences(uncertainty_method="svd-ew")
        t_diff = time.perf_counter() - t0
    finally:
        pymbar.mbar.mbar_solvers = ref_solvers

    def rel(a, b):
        a, b = np.asarray(a), np.asarray(b)
        return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

    with open("/proc/self/maps") as fh:
        mapped = sorted({ln.split()[-1] for ln in fh if "libmbar_hip" in ln or "/oracle/" in ln})
    out = {
        "what": "unchanged reference pymbar.MBAR with pymbar.mbar.mbar_solvers = pymbar_amd.mbar_solvers on the real device",
        "K": K, "N": N, "device": device.device_info()["name"], "mapped_native_code": mapped,
        "reference_tree": ref_tree, "MBAR_class_module": type(mbar).__module__,
        "oracle_imported": any(m == "oracle" or m.startswith("oracle.") for m in sys.modules),
        "constructor_s": t_ctor, "compute_free_energy_differences_s": t_diff,
        "reference_wall_s_for_the_same_journey_on_the_build_container": float(gold["wall_s"]),
        "f_k_max_abs_dev": float(np.max(np.abs(mbar.f_k - gold["f_k"]))),
        "Delta_f_max_rel_dev": rel(res["Delta_f"], gold["Delta_f"]),
        "dDelta_f_max_rel_dev": rel(res["dDelta_f"], gold["dDelta_f_svd_ew"]),
        "Log_W_nk_shape": list(mbar.Log_W_nk.shape), "Log_W_nk_F_contiguous": bool(mbar.Log_W_nk.flags.f_contiguous),
        "tolerances": {"Delta_f": 1e-8, "dDelta_f": 1e-7},
    }
    out["ok"] = bool(out["Delta_f_max_rel_dev"] < 1e-8 and out["dDelta_f_max_rel_dev"] < 1e-7 and mapped
                     and not out["oracle_imported"])
    print(json.dumps(out))
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
