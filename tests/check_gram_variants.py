"""GPU check (stand-alone script; lives under tests/ because it uses the oracle): the three NB = 8 Gram kernels (operand exchange, paired, single wave with pinned accumulator classes)
must agree with each other to round-off and with the oracle on a small K = 128 problem."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pymbar_amd import testsystems
from pymbar_amd.device import DeviceMatrix
from oracle import mbar_oracle as orc

K, n = 128, 777
O_k, K_k = testsystems.ladder_params(K)
N_k = np.full(K, n, dtype=np.int64)
dm = DeviceMatrix.harmonic(O_k, K_k, N_k, seed=3)
dm.set_Nk(N_k)
u = dm.to_host()
rng = np.random.default_rng(0)
f = rng.normal(size=K) * 0.1
f -= f[0]
ref_H = orc.mbar_hessian(u, N_k, f)
out = {}
for v in (0, 1, 2):
    dm.set_option("gram_variant", v)
    psum, sld, G = dm.eval(f, gram=True)
    out[v] = G
    H = np.diag(psum[0]) - G
    err = np.max(np.abs(H - ref_H)) / np.max(np.abs(ref_H))
    print("variant", v, "max rel err of Hessian vs oracle: %.3e" % err, " symmetric:", np.array_equal(G, G.T))
    assert err < 1e-11
for v in (1, 2):
    d = np.max(np.abs(out[v] - out[0])) / np.max(np.abs(out[0]))
    print("variant", v, "vs 0: %.3e" % d)
    assert d < 1e-13
print("OK")
