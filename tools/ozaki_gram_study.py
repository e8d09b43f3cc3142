#!/usr/bin/env python
"""Sizing study for an int8 (Ozaki-style fixed-point) Gram matrix G = sum_n p_n p_n^T of MBAR probabilities p in [0, 1].

Host arithmetic only (numpy int64): what an int8 matrix-core kernel WOULD compute, so that the number of 7-bit slices and of
slice pairs it needs can be read off before anything is built.  p is cut into S signed 7-bit digits d_0 .. d_{S-1} of a
fixed-point number (p ~ sum_i d_i 2^(-7 (i + 1)), |d_i| <= 64, round to nearest at the last digit); a pair (a, b) of slices
contributes D_ab = sum_n d_a d_b^T (exact in int32 for <= 2^31 / 64^2 / 64 ~ 8000 k-steps of 64 samples) at scale 2^(-7 (a + b + 2));
only pairs with a + b < T are formed.  Compared with: numpy's fp64 product (what the fp64 matrix cores compute up to summation
order) and the all-pairs S = 9 result (exact for the 63-bit quantised operands)."""
import sys, os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.special import logsumexp  # noqa: E402
from pymbar_amd import testsystems as ts  # noqa: E402


def digits(p, S, balanced=True):
    """Base-128 digits of p in [0, 1]: p = sum_i d_i 128^-(i+1) + O(128^-S / 2); balanced: d in [-64, 63], else [0, 127]
    (both are int8 values; plain digits need no carry chain but their products do not cancel: 4x less int32 headroom)."""
    q = np.rint(np.ldexp(p, 7 * S)).astype(np.int64)
    out = []
    for i in range(S):  # least significant first
        d = ((q + 64) & 127) - 64 if balanced else q & 127
        out.append(np.asarray(d, dtype=np.float64))  # (fp64 holds these integers and every sum below exactly: BLAS speed)
        q = (q - d) >> 7
    assert np.all(q >= 0) and np.all(q <= 1)  # the carry out of the top digit: p close to 1
    out[-1] = out[-1] + 128.0 * np.asarray(q, dtype=np.float64)  # fold it back (top digit then reaches 128: one of the cases a kernel must handle)
    return out[::-1]  # most significant first


def gram_sliced(p, S, T, balanced=True):
    d = digits(p, S, balanced)
    K = p.shape[0]
    G = np.zeros((K, K), dtype=np.float64)
    npairs = 0
    for lvl in range(min(T, 2 * S - 1) - 1, -1, -1):  # small contributions first
        D = np.zeros((K, K), dtype=np.float64)  # exact: |D| <= S 128^2 N < 2^53
        for a in range(S):
            b = lvl - a
            if 0 <= b < S:
                D += d[a] @ d[b].T
                npairs += 1
        G += np.ldexp(D, -7 * (lvl + 2))
    return G, npairs


def main():
    K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 200_000)
    O_k, K_k, N_k = ts.config3_params(K, N)
    x_n, u_kn, N_k, s_n = ts.harmonic_u_kn(O_k, K_k, N_k, seed=0)
    f = 0.5 * np.log(K_k / K_k[0])  # analytic free energies of the harmonic ladder: near the solution, as in the last iterations
    a = f + np.log(N_k)                                   # a_k = f_k + ln N_k
    logden = logsumexp(a[:, None] - u_kn, axis=0)          # mbar_solvers.py:238
    p = np.exp(a[:, None] - u_kn - logden[None, :])        # probabilities N_k W_nk: columns sum to one
    print(f"K={K} N={u_kn.shape[1]}  column sums within {np.abs(p.sum(0) - 1).max():.1e}; max p {p.max():.3f}")
    G64 = p @ p.T
    Gx, _ = gram_sliced(p, 9, 99)
    scale = np.abs(Gx).max()
    print(f"fp64 product vs 63-bit all-pairs: max |dG| / max |G| = {np.abs(G64 - Gx).max() / scale:.2e}, "
          f"max entrywise relative {np.max(np.abs(G64 - Gx) / np.maximum(np.abs(Gx), 1e-300)):.2e}")
    H = np.diag(p.sum(1)) - Gx
    g = np.random.default_rng(0).normal(size=K) * 1e-3
    x_ref = np.linalg.lstsq(H[1:, 1:], g[1:], rcond=None)[0]
    for S, T, bal in [(S, T, True) for S in (5, 6, 7, 8) for T in range(S - 1, S + 3)] + [(6, 7, False), (7, 7, False), (7, 8, False), (8, 8, False)]:
        if True:
            G, npairs = gram_sliced(p, S, T, bal)
            dn = np.abs(G - Gx).max() / scale
            de = np.max(np.abs(G - Gx) / np.maximum(np.abs(Gx), 1e-300))
            x = np.linalg.lstsq((np.diag(p.sum(1)) - G)[1:, 1:], g[1:], rcond=None)[0]
            print(f"S={S} {'balanced' if bal else 'plain   '} digits, levels a+b<{T}: {npairs:2d} pairs  max|dG|/max|G| {dn:.2e}  entrywise rel {de:.2e}  "
                  f"Newton step rel change {np.abs(x - x_ref).max() / np.abs(x_ref).max():.2e}")


if __name__ == "__main__":
    main()
