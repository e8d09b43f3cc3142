"""The north-star boundary, exercised literally: the UNCHANGED ``pymbar.MBAR`` class of the reference tree
(/root/reference/pymbar/mbar.py:413,437,455,910 are its only calls into the solver module) with ONE binding replaced,

    pymbar.mbar.mbar_solvers = pymbar_amd.mbar_solvers

and compared, in the same process, with the reference running on its own numpy solver module.  The device behind the
drop-in is the CPU stand-in (tests/cpu_standin.py) in the build container, where the reference tree is mounted and there
is no GPU (tests/test_reference_boundary.py; PYTHONPATH = tests/refshim : /root/reference : repo root) -- or, with
``MBAR_REFSHIM_DEVICE=hip`` and ``MBAR_REFERENCE_TREE=<staged copy>``, the REAL device on an MI355X box
(tools/reference_on_gpu_box.sh).  Prints one JSON line with the largest deviations; exits non-zero on a mismatch."""
import json
import os
import sys

import numpy as np


def main():
    import pymbar  # the reference
    import pymbar.mbar

    REF = os.path.realpath(os.environ.get("MBAR_REFERENCE_TREE", "/root/reference"))
    ON_HIP = os.environ.get("MBAR_REFSHIM_DEVICE", "standin") == "hip"
    assert os.path.realpath(pymbar.__file__).startswith(REF), pymbar.__file__
    import pymbar_amd.device
    import pymbar_amd.mbar_solvers as amd_solvers
    from pymbar_amd import testsystems as ts
    if ON_HIP:  # an MI355X box with a staged reference tree: the real device behind the drop-in
        from pymbar_amd import _lib

        _lib.require_device()
        OracleMatrix = pymbar_amd.device.DeviceMatrix
    else:
        from tests.cpu_standin import OracleMatrix

        pymbar_amd.device.DeviceMatrix = OracleMatrix  # the drop-in's device -> numpy oracle (no GPU here)
    ref_solvers = pymbar.mbar.mbar_solvers
    assert os.path.realpath(ref_solvers.__file__).startswith(REF)
    assert pymbar.MBAR.__module__ == "pymbar.mbar"

    worst = {}

    def cmp(name, a, b, tol=1e-12):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        scale = max(1.0, float(np.max(np.abs(a)))) if a.size else 1.0
        dev = float(np.max(np.abs(a - b))) / scale if a.size else 0.0
        worst[name] = max(worst.get(name, 0.0), dev)
        if not dev <= tol:
            print(json.dumps({"mismatch": name, "deviation": dev, "tol": tol}))
            sys.exit(1)

    def both(build):
        pymbar.mbar.mbar_solvers = ref_solvers
        a = build()
        pymbar.mbar.mbar_solvers = amd_solvers  # <- the whole integration
        try:
            b = build()
        finally:
            pymbar.mbar.mbar_solvers = ref_solvers
        return a, b

    cases = {
        "config1": ts.config1(seed=0)[1:3],
        "unsampled": ts.harmonic_u_kn([0, 1, 2, 3], [1, 2, 4, 8], [800, 0, 700, 800], seed=3)[1:3],
        "ladder_K12": ts.harmonic_u_kn(np.linspace(0, 3, 12), np.linspace(1, 3, 12), [400] * 12, seed=5)[1:3],
    }
    for cname, (u_kn, N_k) in cases.items():
        K, N = u_kn.shape
        rng = np.random.default_rng(1)
        A_n = rng.normal(size=N) ** 2 + 0.5
        for proto in (None, "robust"):
            ref, new = both(lambda: pymbar.MBAR(u_kn, N_k, solver_protocol=proto))
            tag = f"{cname}/{proto or 'default'}"
            assert type(new) is type(ref) and type(new).__module__ == "pymbar.mbar"
            cmp(f"{tag}/f_k", ref.f_k, new.f_k)
            cmp(f"{tag}/Log_W_nk", ref.Log_W_nk, new.Log_W_nk)
            assert new.Log_W_nk.shape == (N, K)
            for method in (None, "svd", "approximate"):
                ra = ref.compute_free_energy_differences(uncertainty_method=method)
                rb = new.compute_free_energy_differences(uncertainty_method=method)
                cmp(f"{tag}/Delta_f", ra["Delta_f"], rb["Delta_f"])
                cmp(f"{tag}/dDelta_f[{method}]", ra["dDelta_f"], rb["dDelta_f"], 1e-10)
            ea, eb = ref.compute_expectations(A_n), new.compute_expectations(A_n)
            cmp(f"{tag}/expectation", ea["mu"], eb["mu"])
            cmp(f"{tag}/expectation_sigma", ea["sigma"], eb["sigma"], 1e-10)
            pa, pb = ref.compute_perturbed_free_energies(u_kn[:2] * 1.1), new.compute_perturbed_free_energies(u_kn[:2] * 1.1)
            cmp(f"{tag}/perturbed_f", pa["Delta_f"], pb["Delta_f"])
            cmp(f"{tag}/overlap", ref.compute_overlap()["matrix"], new.compute_overlap()["matrix"])
        # bootstrap replicates (mbar.py:417-449, BOOTSTRAP_SOLVER_PROTOCOL) and the bootstrap branch of the expectations
        # (mbar.py:905-912 calls mbar_solvers.mbar_log_W_nk on the resampled matrix)
        ref, new = both(lambda: pymbar.MBAR(u_kn, N_k, n_bootstraps=4, rseed=11))
        cmp(f"{cname}/bootstrap/f_k_boots", ref.f_k_boots, new.f_k_boots, 1e-10)
        ra = ref.compute_free_energy_differences(uncertainty_method="bootstrap")
        rb = new.compute_free_energy_differences(uncertainty_method="bootstrap")
        cmp(f"{cname}/bootstrap/dDelta_f", ra["dDelta_f"], rb["dDelta_f"], 1e-9)
        ea = ref.compute_expectations(A_n, uncertainty_method="bootstrap")
        eb = new.compute_expectations(A_n, uncertainty_method="bootstrap")
        cmp(f"{cname}/bootstrap/expectation_sigma", ea["sigma"], eb["sigma"], 1e-9)
    # Residency behind the literal drop-in: ONE upload per construction of the unchanged pymbar.MBAR (its calls at mbar.py:413 and
    # :455 hand over the same self.u_kn), none for the expectation / perturbed-free-energy calls that follow (:910 passes
    # self.u_kn again), one per bootstrap replicate (a gathered matrix each).
    uploads = []

    class Counting(OracleMatrix):
        @classmethod
        def from_host(cls, u_kn, device=None, columns=None):
            uploads.append(np.shape(u_kn))
            return super().from_host(u_kn, device=device, columns=columns)

    pymbar_amd.device.DeviceMatrix = Counting
    u_kn, N_k = cases["ladder_K12"]
    pymbar.mbar.mbar_solvers = amd_solvers
    try:
        amd_solvers.drop_resident_cache()
        m = pymbar.MBAR(u_kn, N_k)
        n_ctor = len(uploads)
        m.compute_free_energy_differences()
        m.compute_overlap()
        n_after = len(uploads)
        amd_solvers.drop_resident_cache()
        del uploads[:]
        pymbar.MBAR(u_kn, N_k, n_bootstraps=3, rseed=5)
        n_boot = len(uploads)
    finally:
        pymbar.mbar.mbar_solvers = ref_solvers
        pymbar_amd.device.DeviceMatrix = OracleMatrix
    if (n_ctor, n_after, n_boot) != (1, 1, 4):
        print(json.dumps({"mismatch": "uploads per pymbar.MBAR construction", "uploads": [n_ctor, n_after, n_boot]}))
        sys.exit(1)
    with open("/proc/self/maps") as fh:
        mapped = sorted({ln.split()[-1] for ln in fh if "libmbar_hip" in ln})
    print(json.dumps({"ok": True, "device": "hip" if ON_HIP else "cpu stand-in", "mapped_native_code": mapped,
                      "reference_tree": REF, "worst_deviation": max(worst.values()), "checks": len(worst),
                      "uploads_per_construction": n_ctor, "uploads_per_construction_with_3_bootstraps": n_boot,
                      "worst": sorted(worst.items(), key=lambda kv: -kv[1])[:5]}))


if __name__ == "__main__":
    main()
