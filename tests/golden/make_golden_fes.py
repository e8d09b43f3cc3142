"""Golden fixture for the FES weight extraction, generated from the UNMODIFIED reference:

    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_fes.py

The system is the 1-D umbrella-sampling example of the reference's own tests/test_fes.py (7 umbrellas x 1000 samples,
K0 = 20, Ku = 100, 15 bins), seeded.  Stored: the inputs, the reference's log_w_n (fes.py:410), its sample labels /
bin order, histogram_data["f"] (fes.py:585) and get_fes(..., uncertainty_method="analytical") for both reference-point
modes (fes.py:1362-1415)."""
import logging
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
logging.disable(logging.WARNING)
import pymbar  # noqa: E402
from pymbar import FES  # noqa: E402

assert os.path.realpath(pymbar.__file__).startswith("/root/reference"), pymbar.__file__


def main():
    np.random.seed(1234)
    beta, K0, Ku, gridscale, nsamples, nbins = 1.0, 20.0, 100.0, 0.2, 1000, 15
    centers = gridscale * np.arange(-3, 4, dtype=float)
    K = len(centers)
    x_n = np.zeros([K * nsamples, 1])
    for i, c in enumerate(centers):
        sigma = 1.0 / (K0 + Ku)
        mu = sigma * (0.0 * K0 + c * Ku)
        x_n[i * nsamples:(i + 1) * nsamples, 0] = np.random.normal(mu, np.sqrt(sigma), [nsamples])
    u_n = beta * (K0 / 2) * np.sum(x_n ** 2, axis=1)
    u_kn = np.zeros([K, K * nsamples])
    for k, c in enumerate(centers):
        u_kn[k] = u_n + beta * (Ku / 2.0) * (x_n[:, 0] - c) ** 2
    N_k = nsamples * np.ones(K, int)
    xmin, xmax = gridscale * (-3 - 0.5), gridscale * (3 + 0.5)
    bin_edges = np.linspace(xmin, xmax, nbins + 1)
    bin_centers = (0.5 * (bin_edges[1:] + bin_edges[:-1]))[:, None]

    fes = FES(u_kn, N_k)
    fes.generate_fes(u_n, x_n, histogram_parameters={"bin_edges": bin_edges})
    hd = fes.histogram_data
    order = hd["bin_order"]
    sample_label = np.array([order[l] if l in order else -1 for l in hd["sample_label"]], dtype=np.int64)
    log_w_n = fes.mbar._computeUnnormalizedLogWeights(u_n)
    populated = sorted(order, key=lambda l: order[l])           # grid labels in bin order
    # np.digitize puts samples right of the last edge into grid label `nbins`, which the reference keeps as a bin of its
    # own (fes.py:525); it has no centre to query, but it is a column of the covariance like every other bin
    query = np.array([order[l] for l in populated if 0 <= l < nbins], dtype=np.int64)   # bin-order indices of the in-grid bins
    centers_in_order = bin_centers[np.array([l for l in populated if 0 <= l < nbins], dtype=int)]
    lowest = fes.get_fes(centers_in_order, reference_point="from-lowest", uncertainty_method="analytical")
    spec = fes.get_fes(centers_in_order, reference_point="from-specified", fes_reference=0.0, uncertainty_method="analytical")
    np.savez_compressed(os.path.join(HERE, "fes_umbrella_1d.npz"), u_kn=u_kn, N_k=N_k, u_n=u_n, x_n=x_n, bin_edges=bin_edges,
                        f_k=fes.mbar.f_k, sample_label=sample_label, log_w_n=log_w_n, f_raw=hd["f"],
                        grid_of_label=np.array(populated, dtype=np.int64), query=query,
                        f_lowest=lowest["f_i"], df_lowest=lowest["df_i"], f_specified=spec["f_i"], df_specified=spec["df_i"],
                        specified_label=np.int64(order[int(np.digitize(0.0, bin_edges) - 1)]))
    print("nbins populated:", len(populated), "f range", hd["f"].min(), hd["f"].max(), "df", lowest["df_i"][:4])


if __name__ == "__main__":
    main()
