"""FES weight extraction (pymbar/fes.py:403-416, 1383-1406) against the reference's own FES on its
1-D umbrella-sampling example (tests/golden/fes_umbrella_1d.npz, tests/golden/make_golden_fes.py): log weights in the
target potential, bin free energies and analytical uncertainties for both reference-point modes.  The histogram bins are
extra rows of a DEVICE matrix (``mbar_ctx_fill_masked_rows``); no N x (K + nbins) array exists on the host."""
import numpy as np
import pytest

import pymbar_amd
from pymbar_amd import fes as amd_fes
from tests.conftest import load_golden


def _check(mbar, g):
    np.testing.assert_allclose(mbar.f_k, g["f_k"], atol=1e-9)
    log_w = amd_fes.unnormalized_log_weights(mbar, g["u_n"])
    np.testing.assert_allclose(log_w, g["log_w_n"], rtol=1e-10, atol=1e-9)
    labels, grid = amd_fes.label_samples(g["x_n"], g["bin_edges"])
    np.testing.assert_array_equal(labels, g["sample_label"])
    q = g["query"]
    low = amd_fes.histogram_fes(mbar, g["u_n"], labels, reference="from-lowest")
    np.testing.assert_allclose(low["f_raw"], g["f_raw"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(low["f_i"][q], g["f_lowest"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(low["df_i"][q], g["df_lowest"], rtol=1e-7, atol=1e-9)
    spec = amd_fes.histogram_fes(mbar, g["u_n"], labels, reference="from-specified", reference_label=int(g["specified_label"]))
    np.testing.assert_allclose(spec["f_i"][q], g["f_specified"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(spec["df_i"][q], g["df_specified"], rtol=1e-7, atol=1e-9)
    # each bin column of the augmented weight matrix is normalised, and the bin probabilities add up to one
    assert abs(np.exp(-low["f_raw"]).sum() / np.exp(log_w).sum() - 1.0) < 1e-10
    with pytest.raises(Exception):
        amd_fes.histogram_fes(mbar, g["u_n"], np.where(labels == 3, 2, labels))  # bin 3 emptied


@pytest.mark.gpu
def test_histogram_fes_on_the_device_matches_reference():
    g = load_golden("fes_umbrella_1d.npz")
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"])
    try:
        _check(mbar, g)
    finally:
        mbar.close()


def test_histogram_fes_host_logic_on_standin(monkeypatch):
    import pymbar_amd.device
    from tests.cpu_standin import OracleMatrix

    monkeypatch.setattr(pymbar_amd.device, "DeviceMatrix", OracleMatrix)
    g = load_golden("fes_umbrella_1d.npz")
    mbar = pymbar_amd.MBAR(g["u_kn"], g["N_k"])
    _check(mbar, g)
