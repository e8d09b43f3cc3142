"""Golden fixture for ``initialize="BAR"``, generated from the UNMODIFIED reference:

    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_bar.py

Inputs are regenerated from the seeded generators of ``pymbar_amd/testsystems.py``; stored are the reference's
``bar_zero`` / ``bar(method="bisection")`` values for adjacent pairs, the chained initial guess of
``MBAR._initialize_with_bar`` and the free energies of ``MBAR(initialize="BAR")``.
"""
import importlib.util
import logging
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("ts", os.path.join(ROOT, "pymbar_amd", "testsystems.py"))
ts = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ts)

logging.disable(logging.WARNING)
import pymbar  # noqa: E402
from pymbar.other_estimators import bar, bar_zero  # noqa: E402

assert os.path.realpath(pymbar.__file__).startswith("/root/reference"), pymbar.__file__


def block(tag, u_kn, N_k):
    N_k = np.asarray(N_k)
    m = pymbar.MBAR(u_kn, N_k, initialize="BAR")
    m0 = pymbar.MBAR(u_kn, N_k)
    f_init = m0._initialize_with_bar(m0.u_kn)  # the chained guess itself (does not touch the solved f_k)
    order = np.where(N_k > 0)[0]
    x = m0.x_kindices
    pairs, zeros, dfs = [], [], []
    for k, l in zip(order[:-1], order[1:]):
        w_F = u_kn[l, x == k] - u_kn[k, x == k]
        w_R = u_kn[k, x == l] - u_kn[l, x == l]
        pairs.append((k, l))
        zeros.append([bar_zero(w_F, w_R, d) for d in (-1.0, 0.0, 0.7)])
        dfs.append(bar(w_F, w_R, method="bisection", relative_tolerance=1e-5, maximum_iterations=100,
                       compute_uncertainty=False)["Delta_f"])
    return {tag + "_f_init": f_init, tag + "_f_k": m.f_k, tag + "_pairs": np.array(pairs), tag + "_bar_zero": np.array(zeros),
            tag + "_bar_delta_f": np.array(dfs)}


def main():
    out = {}
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config1(seed=0)
    out.update(block("config1", u_kn, N_k))
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn([1, 2, 3, 4], [0.5, 1.0, 1.5, 2.0], [1000, 500, 0, 800], seed=3)
    out.update(block("unsampled", u_kn, N_k))
    x_n, u_kn, N_k, s_n, O_k, K_k = ts.config5(seed=0)
    out.update(block("config5", u_kn, N_k))
    np.savez_compressed(os.path.join(HERE, "bar_init.npz"), **out)
    print("wrote bar_init.npz", sorted(out))


if __name__ == "__main__":
    main()
