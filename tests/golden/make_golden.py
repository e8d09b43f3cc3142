"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run (in the build container only; /root/reference does not exist on the GPU box):

    cd /root/repo && PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden.py

(oracle recipe A of SURVEY.md 8(c): Python 3.9, numpy 1.26.4, scipy 1.7.1, numexpr present; the
reference runs its numpy backend -- JAX is absent.)  Inputs come from the seeded generators of
``pymbar_amd/testsystems.py`` (loaded by path so that nothing else of the package is imported);
the script asserts that they are bit-identical to what the reference's own test systems draw for
the same seed.  Outputs are what the reference computes on those inputs.
"""
import importlib.util
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

spec = importlib.util.spec_from_file_location("ts", os.path.join(ROOT, "pymbar_amd", "testsystems.py"))
ts = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ts)

logging.disable(logging.WARNING)
import pymbar  # noqa: E402  (the reference, from PYTHONPATH=/root/reference)
from pymbar import mbar_solvers as ref  # noqa: E402
from pymbar import testsystems as ref_ts  # noqa: E402

assert os.path.realpath(pymbar.__file__).startswith("/root/reference"), pymbar.__file__


def l1_block(u_kn, N_k, f_k):
    """Every L1 function of the reference evaluated at one (generally unconverged) f_k."""
    Nf = 1.0 * np.asarray(N_k)
    obj, grad = ref.mbar_objective_and_gradient(u_kn, Nf, f_k)
    logW = ref.mbar_log_W_nk(u_kn, Nf, f_k)
    return dict(
        f_eval=f_k,
        gradient=ref.mbar_gradient(u_kn, Nf, f_k),
        sci=ref.self_consistent_update(u_kn, Nf, f_k),
        objective=np.float64(ref.mbar_objective(u_kn, Nf, f_k)),
        objective2=np.float64(obj),
        gradient2=grad,
        hessian=ref.mbar_hessian(u_kn, Nf, f_k),
        logW_colsum=np.exp(logW).sum(0),
        logW_sample=logW[:: max(1, logW.shape[0] // 64)].copy(),
        logW_stride=np.int64(max(1, logW.shape[0] // 64)),
    )


def adaptive_block(u_kn, N_k, f0, tol=1e-12, **opts):
    """solve_mbar_once(method="adaptive") of the reference, with its iteration bookkeeping."""
    records = []

    class Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())

    h = Grab()
    logging.disable(logging.NOTSET)
    ref.logger.addHandler(h)
    ref.logger.setLevel(logging.INFO)
    options = dict(verbose=True)
    options.update(opts)
    f, results = ref.solve_mbar_once(u_kn, N_k, f0, method="adaptive", tol=tol, options=options)
    ref.logger.removeHandler(h)
    logging.disable(logging.WARNING)
    nr = sci = iters = -1
    for m in records:
        if m.startswith("Of "):
            parts = m.split()
            iters, nr, sci = int(parts[1]), int(parts[3]), int(parts[8])
    choices = []
    for m in records:
        if m.startswith("Choosing self-consistent"):
            choices.append(0)
        elif m.startswith("Newton-Raphson used"):
            choices.append(1)
    return dict(f_adaptive=np.array(f), adaptive_success=np.bool_(results["success"]),
                adaptive_iters=np.int64(iters), adaptive_nr=np.int64(nr), adaptive_sci=np.int64(sci),
                adaptive_choices=np.array(choices, dtype=np.int8))


def mbar_block(u_kn, N_k, **kw):
    m = pymbar.MBAR(u_kn, N_k, **kw)
    out = dict(f_k=np.array(m.f_k))
    for method in ("svd-ew", "svd", "approximate"):
        r = m.compute_free_energy_differences(uncertainty_method=method, return_theta=True)
        tag = method.replace("-", "_")
        out["Delta_f"] = r["Delta_f"]
        out["dDelta_f_" + tag] = r["dDelta_f"]
        out["Theta_" + tag] = r["Theta"]
    ov = m.compute_overlap()
    out["overlap_matrix"] = ov["matrix"]
    out["overlap_scalar"] = np.float64(np.real(ov["scalar"]))
    out["N_eff"] = m.compute_effective_sample_number()
    return out


def expectations_block(x_n, u_kn, N_k, **kw):
    """Log_W_nk consumers of the reference (mbar.py:732-1681) on one data set."""
    m = pymbar.MBAR(u_kn, N_k, **kw)
    out = dict()
    r = m.compute_expectations(x_n)
    out["exp_x_mu"], out["exp_x_sigma"] = r["mu"], r["sigma"]
    r = m.compute_expectations(x_n ** 2, output="differences")
    out["exp_x2_diff_mu"], out["exp_x2_diff_sigma"] = r["mu"], r["sigma"]
    r = m.compute_expectations(u_kn, state_dependent=True, return_theta=True)
    out["exp_u_sd_mu"], out["exp_u_sd_sigma"], out["exp_u_sd_Theta"] = r["mu"], r["sigma"], r["Theta"]
    u_new = u_kn[:3] * 1.1 + 0.3
    r = m.compute_expectations(x_n, u_kn=u_new)
    out["exp_x_newstates_mu"], out["exp_x_newstates_sigma"] = r["mu"], r["sigma"]
    A_in = np.array([x_n, x_n ** 2, np.cos(x_n)])
    r = m.compute_multiple_expectations(A_in, u_kn[1], compute_covariance=True)
    out["multi_mu"], out["multi_sigma"], out["multi_cov"] = r["mu"], r["sigma"], r["covariances"]
    r = m.compute_perturbed_free_energies(u_new)
    out["pert_Delta_f"], out["pert_dDelta_f"] = r["Delta_f"], r["dDelta_f"]
    r = m.compute_entropy_and_enthalpy()
    for key in ("Delta_f", "dDelta_f", "Delta_u", "dDelta_u", "Delta_s", "dDelta_s"):
        out["se_" + key] = r[key]
    inner = m.compute_expectations_inner(A_in, u_kn[:2], np.array([[0, 0, 1, 1], [0, 1, 2, 1]]), return_theta=True)
    out["inner_observables"], out["inner_f"], out["inner_Theta"], out["inner_Amin"] = (
        inner["observables"], inner["f"], inner["Theta"], inner["Amin"])
    d_ij = np.abs(np.sin(np.arange(36.0).reshape(6, 6))) + 0.1
    out["cov_of_sums"] = m.compute_covariance_of_sums(d_ij + d_ij.T, 3, [0.5, -1.5])
    return out


def main():
    manifest = {}

    # ---- config 1: HarmonicOscillatorsTestCase defaults, K=5, N=5000, seed 0 -----------------
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config1(seed=0)
    rx, ru, rN, rs = ref_ts.HarmonicOscillatorsTestCase().sample(N_k=[1000] * 5, mode="u_kn", seed=0)
    assert np.array_equal(ru, u_kn) and np.array_equal(rx, x_n), "generator differs from reference sampler"
    out = dict(u_kn=u_kn, N_k=N_k)
    out.update(mbar_block(u_kn, N_k))
    rng = np.random.RandomState(11)
    f_eval = out["f_k"] + 0.3 * rng.standard_normal(5)
    f_eval[0] = 0.0
    out.update(l1_block(u_kn, N_k, f_eval))
    out.update(adaptive_block(u_kn, N_k, np.zeros(5), min_sc_iter=0))
    out["precond_sample"] = ref.precondition_u_kn(u_kn, 1.0 * N_k, f_eval)[:, ::97].copy()
    np.savez_compressed(os.path.join(HERE, "config1_ho_K5_N5000.npz"), **out)
    manifest["config1_ho_K5_N5000.npz"] = "HarmonicOscillatorsTestCase defaults, N_k=[1000]*5, seed=0; default protocol"

    # ---- reference test_mbar fixture shape: N_k=[1000,500,0,800] (an unsampled state) ---------
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([1, 2, 3, 4], [0.5, 1.0, 1.5, 2.0], [1000, 500, 0, 800], seed=3)
    rx, ru, rN, rs = ref_ts.HarmonicOscillatorsTestCase([1, 2, 3, 4], [0.5, 1.0, 1.5, 2.0]).sample(
        N_k=[1000, 500, 0, 800], mode="u_kn", seed=3)
    assert np.array_equal(ru, u_kn)
    out = dict(u_kn=u_kn, N_k=N_k)
    out.update(mbar_block(u_kn, N_k))
    out_r = mbar_block(u_kn, N_k, solver_protocol="robust")
    out["f_k_robust"] = out_r["f_k"]
    np.savez_compressed(os.path.join(HERE, "ho_unsampled_K4_N2300.npz"), **out)
    manifest["ho_unsampled_K4_N2300.npz"] = "test_mbar.py fixture shape (tests/test_mbar.py:44-60), seed=3"

    # ---- expectations / perturbed free energies / entropy-enthalpy on two data sets -------------------------
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config1(seed=0)
    np.savez_compressed(os.path.join(HERE, "expectations_config1.npz"), **expectations_block(x_n, u_kn, N_k))
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([1, 2, 3, 4], [0.5, 1.0, 1.5, 2.0], [1000, 500, 0, 800], seed=3)
    np.savez_compressed(os.path.join(HERE, "expectations_unsampled.npz"), **expectations_block(x_n, u_kn, N_k))
    manifest["expectations_config1.npz"] = "Log_W_nk consumers (mbar.py:732-1681) on config1 (seed 0); inputs regenerated from seed"
    manifest["expectations_unsampled.npz"] = "same on the K=4 fixture with an unsampled state (seed 3)"

    # ---- exponentials 20 x 50 (tests/test_mbar_solvers.py:28 shape, smaller) ---------------------
    rates = np.linspace(1, 3, 20)
    x_n, u_kn, N_k, s_n = ts.exponential_u_kn(rates, [50] * 20, seed=5)
    rx, ru, rN, rs = ref_ts.ExponentialTestCase(rates).sample([50] * 20, mode="u_kn", seed=5)
    assert np.array_equal(ru, u_kn)
    out = dict(u_kn=u_kn, N_k=N_k)
    out.update(mbar_block(u_kn, N_k))
    rng = np.random.RandomState(12)
    f_eval = out["f_k"] + 0.2 * rng.standard_normal(20)
    f_eval[0] = 0.0
    out.update(l1_block(u_kn, N_k, f_eval))
    np.savez_compressed(os.path.join(HERE, "exp_K20_N1000.npz"), **out)
    manifest["exp_K20_N1000.npz"] = "ExponentialTestCase rates=linspace(1,3,20), N_k=[50]*20, seed=5"

    # ---- oscillators 50 x 100 (tests/test_mbar_solvers.py:13-16 shape), outputs only -------------
    O_k = np.linspace(1, 5, 50)
    K_k = np.linspace(1, 3, 50)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, [100] * 50, seed=7)
    out = dict(N_k=N_k)
    out.update(mbar_block(u_kn, N_k))
    rng = np.random.RandomState(13)
    f_eval = out["f_k"] + 0.1 * rng.standard_normal(50)
    f_eval[0] = 0.0
    out.update(l1_block(u_kn, N_k, f_eval))
    out.update(adaptive_block(u_kn, N_k, np.zeros(50), min_sc_iter=0))
    np.savez_compressed(os.path.join(HERE, "osc_K50_N5000.npz"), **out)
    manifest["osc_K50_N5000.npz"] = "oscillators(50,100): O=linspace(1,5), K=linspace(1,3), seed=7; inputs regenerated from seed"

    # ---- config 2 shape at reduced N (K=32, N=32000): adaptive + SCI iteration counts -----------
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config2(seed=0, K=32, N=32000)
    out = dict(N_k=N_k)
    out.update(adaptive_block(u_kn, N_k, np.zeros(32), min_sc_iter=0))
    ad2 = adaptive_block(u_kn, N_k, np.zeros(32))  # default min_sc_iter=2
    out["f_adaptive_msc2"] = ad2["f_adaptive"]
    out["adaptive_iters_msc2"] = ad2["adaptive_iters"]
    out["adaptive_nr_msc2"] = ad2["adaptive_nr"]
    out["adaptive_sci_msc2"] = ad2["adaptive_sci"]
    # pure SCI loop of SURVEY.md 3.3 on the reference's own self_consistent_update
    f = np.zeros(32)
    Nf = 1.0 * N_k
    it = 0
    for it in range(1, 10001):
        fn = ref.self_consistent_update(u_kn, Nf, f)
        fn = fn - fn[0]
        div = np.abs(fn[1:])
        div[div < 1e-12] = 1.0
        delta = np.max(np.abs(fn[1:] - f[1:]) / div)
        f = fn
        if delta < 1e-12:
            break
    out["f_sci"] = f
    out["sci_iters"] = np.int64(it)
    out.update(mbar_block(u_kn, N_k))
    np.savez_compressed(os.path.join(HERE, "ladder_K32_N32000.npz"), **out)
    manifest["ladder_K32_N32000.npz"] = "config-2 generator at K=32, N=32000, seed=0; inputs regenerated from seed"

    # ---- config 5: alchemical-shaped K=40, N=95000 with two unsampled states ---------------------
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    out = dict(N_k=N_k)
    out.update(mbar_block(u_kn, N_k))
    out_a = mbar_block(u_kn, N_k, solver_protocol=(dict(method="adaptive", options=dict(min_sc_iter=0)),))
    out["f_k_adaptive_protocol"] = out_a["f_k"]
    np.savez_compressed(os.path.join(HERE, "config5_alch_K40_N95000.npz"), **out)
    manifest["config5_alch_K40_N95000.npz"] = "config5(seed=0): K=40, 2500/state, states 7 and 23 unsampled; inputs regenerated from seed"

    manifest["_versions"] = dict(python=sys.version.split()[0], numpy=np.__version__,
                                 scipy=__import__("scipy").__version__,
                                 reference="choderalab/pymbar snapshot 2025-11-14 (/root/reference), numpy backend")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote", sorted(manifest))


if __name__ == "__main__":
    main()
