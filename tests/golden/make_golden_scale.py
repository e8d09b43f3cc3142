"""Reference answers AT THE SIZES THE METRIC IS QUOTED ON, from the UNMODIFIED reference.

Run (build container only; /root/reference does not exist on the GPU box), one stage at a time:

    cd /root/repo
    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_scale.py c2     # ~5 min,  ~3 GB
    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_scale.py k128   # ~10 min, ~8 GB
    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_scale.py c3     # ~30 min, ~52 GB RSS

Inputs are NOT stored (config 3's matrix is 10.24 GB): they are regenerated from the seed by the host
generators of ``pymbar_amd/testsystems.py`` (legacy ``np.random.seed`` + ``normal`` draws in state order, i.e. exactly
what pymbar/testsystems/harmonic_oscillators.py:154-188 draws -- asserted below for a column block).  Outputs are a few
KB each: what the reference computes on those inputs.

  c2    BASELINE.json config 2, full size: K=32, N=1e6, seed 0.  Pure self-consistent loop on the reference's own
        ``self_consistent_update`` (mbar_solvers.py:231-242; stopping rule = ``adaptive``'s, :627-640) -> ``f_sci``,
        ``sci_iters``; ``solve_mbar_once(method="adaptive")`` (:510-667) with min_sc_iter 0 and 2 -> ``f_adaptive``,
        iteration / NR / SCI counts and the per-iteration choice; ``pymbar.MBAR`` default protocol ->
        ``f_k``, ``Delta_f``, ``dDelta_f`` (svd-ew).
  k128  the headline state count at the largest N the reference's whole class journey fits comfortably:
        K=128, N=1e6, seed 0 -- the same quantities (no pure SCI loop: ~1 h on this host).
  c3    BASELINE.json config 3 ITSELF: K=128, N=1e7, seed 0: ``solve_mbar_once(method="adaptive", tol=1e-12,
        min_sc_iter=0)`` from f=0 -> ``f_adaptive`` + counts.  Nothing else (one gradient call of the reference
        needs ~40 GB there).
"""
import importlib.util
import json
import os
import resource
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference, asserts it is /root/reference)

ref, ts, pymbar, ref_ts = mg.ref, mg.ts, mg.pymbar, mg.ref_ts


def _check_generator(O_k, K_k, n_per_state=64, seed=0):
    N_k = [n_per_state] * len(O_k)
    _, u, _, _ = ts.harmonic_u_kn(O_k, K_k, N_k, seed=seed)
    _, ru, _, _ = ref_ts.HarmonicOscillatorsTestCase(O_k, K_k).sample(N_k=N_k, mode="u_kn", seed=seed)
    assert np.array_equal(u, ru), "generator differs from the reference sampler"


def _mbar_small(u_kn, N_k):
    m = pymbar.MBAR(u_kn, N_k)
    r = m.compute_free_energy_differences(uncertainty_method="svd-ew")
    return dict(f_k=np.array(m.f_k), Delta_f=r["Delta_f"], dDelta_f_svd_ew=r["dDelta_f"])


def _rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def _save(name, out, t0, note):
    out["wall_s"] = np.float64(time.time() - t0)
    out["peak_rss_gb"] = np.float64(_rss_gb())
    np.savez_compressed(os.path.join(HERE, name), **out)
    mpath = os.path.join(HERE, "MANIFEST.json")
    with open(mpath) as fh:
        manifest = json.load(fh)
    manifest[name] = note
    with open(mpath, "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote", name, {k: (v.shape if getattr(v, "ndim", 0) else v) for k, v in out.items()}, flush=True)


def stage_c2():
    t0 = time.time()
    K, N = 32, 1_000_000
    O_k, K_k = ts.ladder_params(K)
    _check_generator(O_k, K_k)
    x_n, u_kn, N_k, s_n, _, _ = ts.config2(seed=0, K=K, N=N)
    out = dict(N_k=N_k, seed=np.int64(0))
    out.update(mg.adaptive_block(u_kn, N_k, np.zeros(K), min_sc_iter=0))
    ad2 = mg.adaptive_block(u_kn, N_k, np.zeros(K))
    for key in ("f_adaptive", "adaptive_iters", "adaptive_nr", "adaptive_sci", "adaptive_choices"):
        out[key + "_msc2"] = ad2[key]
    f = np.zeros(K)
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
    out.update(_mbar_small(u_kn, N_k))
    _save("scale_config2_K32_N1e6.npz", out, t0,
          "config 2 FULL SIZE (K=32, N=1e6, seed 0): pure SCI loop, adaptive (min_sc_iter 0 and 2), MBAR default "
          "protocol f_k / Delta_f / dDelta_f(svd-ew); outputs only, inputs regenerated from the seed (make_golden_scale.py c2)")


def stage_k128():
    t0 = time.time()
    K, N = 128, 1_000_000
    O_k, K_k, N_k = ts.config3_params(K, N)
    _check_generator(O_k, K_k)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
    out = dict(N_k=N_k, seed=np.int64(0))
    out.update(mg.adaptive_block(u_kn, N_k, np.zeros(K), min_sc_iter=0))
    print("adaptive done", time.time() - t0, flush=True)
    out.update(_mbar_small(u_kn, N_k))
    _save("scale_K128_N1e6.npz", out, t0,
          "K=128, N=1e6, seed 0 (config-3 generator): adaptive from zeros (min_sc_iter 0) with counts and choices, "
          "MBAR default protocol f_k / Delta_f / dDelta_f(svd-ew); outputs only (make_golden_scale.py k128)")


def stage_c3(N=10_000_000):
    t0 = time.time()
    K = 128
    O_k, K_k, N_k = ts.config3_params(K, N)
    _check_generator(O_k, K_k)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
    del x_n, s_n
    print("generated", time.time() - t0, _rss_gb(), flush=True)
    out = dict(N_k=N_k, seed=np.int64(0))
    out.update(mg.adaptive_block(u_kn, N_k, np.zeros(K), min_sc_iter=0))
    name = "scale_config3_K128_N1e7.npz" if N == 10_000_000 else "scale_config3_K128_N%d.npz" % N
    _save(name, out, t0,
          "config 3 ITSELF (K=128, N=%d, seed 0): solve_mbar_once(method='adaptive', tol=1e-12, min_sc_iter=0) from "
          "zeros: f_adaptive, counts, per-iteration choice; outputs only (make_golden_scale.py c3)" % N)


if __name__ == "__main__":
    stage = sys.argv[1]
    if stage == "c2":
        stage_c2()
    elif stage == "k128":
        stage_k128()
    elif stage == "c3":
        stage_c3(int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000)
    else:
        raise SystemExit("stage must be c2, k128 or c3")
