#!/usr/bin/env python
"""Lane-level numpy model of the blocked LDL^T Newton solve of k_select_newton (pymbar_amd/csrc/mbar_k_solver.hip,
newton_body_ldlt): every "register" is a 64-vector, the matrix instruction is its documented layout, LDS is a dict.  It pins the
layout algebra of the kernel on the CPU (no GPU in the build container) and is the executable statement of its algorithm:

  * rows / columns = states 0 .. Kp-1 in their natural order (16 x 16 blocks = the blocks of the reduced Gram record);
    live = sampled and not the gauge state; every other row is an identity row; row 0 -- never live: state 0 is the gauge state
    or unsampled -- carries the right-hand side and rides along (mbar_solvers.py:581-583: lstsq(H, g) minus its gauge component);
  * pivots from the LAST row upwards (A = U D U^T); block (i, j), i <= j, is held TRANSPOSED in the accumulator layout of
    v_mfma_f64_16x16x4_f64 (lane (g, r), register t <-> row 16 i + r, column 16 j + 4 t + g), so that both the pivot row of the
    diagonal block and column p of a panel block are operands as they stand (lane group g = p & 3 of register t = p >> 2);
  * per pivot ONE rank-1 matrix instruction per block of the panel, the diagonal block eliminated redundantly by all four
    waves (no barrier inside a block column); per block column one exchange of the frozen panel through LDS and one rank-16
    update (four matrix instructions) per block of the trailing matrix;
  * x by forward substitution on W = -V / d with x_0 = -1.

    python tools/newton_ldlt_model.py        # random problems against numpy.linalg.solve
"""
import numpy as np

LANES = np.arange(64)
G = LANES >> 4   # lane group = K index of an operand, row offset of an accumulator register
R = LANES & 15   # row (A operand) / column (B operand, accumulator)


def mfma(a, b, acc):
    """acc[t][lane (g, j)] = D[4 t + g][j] += sum_k A[i][k] B[k][j]; A: lane (k, i), B: lane (k, j)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[R, G] = a
    B[G, R] = b
    D = A @ B
    out = [acc[t].copy() for t in range(4)]
    for t in range(4):
        out[t] += D[4 * t + G, R]
    return out


def owner(i, j):
    return (i + j) & 3


def solve(Ahat, nb, first_pivot_floor=1):
    """Ahat: (16 nb) x (16 nb) symmetric, identity rows for dead states, row/column 0 = right-hand side.  Returns x (x[0] = -1)."""
    # set-up: wave w holds blocks (i, j), i < j, with owner(i, j) = w; the diagonal masters at wave j & 3
    blk = {}
    for j in range(nb):
        for i in range(j + 1):
            sub = Ahat[16 * i:16 * i + 16, 16 * j:16 * j + 16]
            blk[(i, j)] = [sub[R, 4 * t + G].copy() for t in range(4)]   # lane (g, r) reg t = A[16 i + r][16 j + 4 t + g]
    lds_w = {}      # (i, kb) -> scaled frozen panel
    lds_dw = {}     # kb -> scaled frozen diagonal block
    pivots = np.ones(16 * nb)
    for kb in range(nb - 1, -1, -1):
        # every wave: working copy of the diagonal block + its panel blocks
        D = [v.copy() for v in blk[(kb, kb)]]
        panels = {i: blk[(i, kb)] for i in range(kb)}
        RV = [np.zeros(64) for _ in range(4)]
        for p in range(15, (0 if kb else first_pivot_floor) - 1, -1):
            t, g = p >> 2, p & 3
            d = D[t][16 * g + p]                      # readlane
            pivots[16 * kb + p] = d
            nr = -1.0 / d
            X = np.where((G == g) & (R < p), D[t], 0.0)
            D = mfma(X, X * nr, D)
            for i in panels:
                Y = np.where(G == g, panels[i][t] * nr, 0.0)
                panels[i] = mfma(X, Y, panels[i])
            RV[t] = np.where(G == g, nr, RV[t])
        if kb == 0:
            RV[0] = np.where(G == 0, 0.0, RV[0])      # (pivot 0 never happens)
        # publish: plain V (this step only) and W = V * (-1/d) (kept for the substitution)
        lds_v = {i: panels[i] for i in panels}
        for i in panels:
            lds_w[(i, kb)] = [panels[i][t] * RV[t] for t in range(4)]
        # frozen diagonal block: entry [c][r'] for r' < c is the pivot row of pivot c; scaled by -1/d_c (c = 4 t + g)
        lds_dw[kb] = [np.where(R < 4 * t + G, D[t] * RV[t], 0.0) for t in range(4)]
        # trailing update: B_ij[c][r] += sum_P V_j[c][P] W_i[r][P]
        for j in range(kb):
            for i in range(j + 1):
                acc = blk[(i, j)]
                for t in range(4):
                    acc = mfma(lds_v[j][t], lds_w[(i, kb)][t], acc)
                blk[(i, j)] = acc
    # forward substitution: x_P = sum_{r < P} W[r][P] x_r, x_0 = -1
    n = 16 * nb
    x = np.zeros(n)
    for kb in range(nb):
        accP = np.zeros(16)
        for i in range(kb):
            Wb = lds_w[(i, kb)]
            for P in range(16):
                t, g = P >> 2, P & 3
                accP[P] += np.dot(Wb[t][16 * g:16 * g + 16], x[16 * i:16 * i + 16])
        Wd = lds_dw[kb]
        xb = np.zeros(16)
        for P in range(16):
            t, g = P >> 2, P & 3
            if kb == 0 and P == 0:
                xb[0] = -1.0
                continue
            xb[P] = accP[P] + np.dot(Wd[t][16 * g:16 * g + 16], xb)   # entries r >= P are zero by the mask
        x[16 * kb:16 * kb + 16] = xb
    return x, pivots


def build_problem(rng, K, sampled, first):
    Kp = (K + 15) // 16 * 16
    n = 400
    w = rng.random((n, K))
    w /= w.sum(1, keepdims=True)
    Gm = w.T @ w
    ps = w.sum(0)
    H = np.diag(ps) - Gm
    g = rng.standard_normal(K)
    live = np.zeros(Kp, bool)
    live[:K] = sampled
    live[first] = False
    A = np.eye(Kp)
    idx = np.where(live)[0]
    A[np.ix_(idx, idx)] = H[np.ix_(idx, idx)]
    A[0, idx] = g[idx]
    A[idx, 0] = g[idx]
    xref = np.zeros(Kp)
    xref[idx] = np.linalg.solve(H[np.ix_(idx, idx)], g[idx])
    return A, Kp // 16, xref, live


def main():
    rng = np.random.default_rng(0)
    worst = 0.0
    for K in (5, 16, 17, 40, 64, 100, 127, 128):
        for trial in range(3):
            sampled = np.ones(K, bool)
            if trial == 1 and K > 4:
                sampled[rng.choice(K, K // 5, replace=False)] = False
            if trial == 2:
                sampled[0] = False
            first = int(np.where(sampled)[0][0])
            A, nb, xref, live = build_problem(rng, K, sampled, first)
            x, piv = solve(A, nb)
            err = np.max(np.abs(x[live] - xref[live])) / max(1e-300, np.max(np.abs(xref)))
            dead = np.max(np.abs(x[1:][~live[1:]])) if (~live[1:]).any() else 0.0
            worst = max(worst, err)
            print(f"K={K:4d} trial {trial}: nb={nb} unknowns={live.sum():4d} rel err {err:.2e} dead {dead:.1e} x0 {x[0]:+.0f} min pivot {piv[live].min():.3e}")
    print("worst", worst)
    assert worst < 1e-9


if __name__ == "__main__":
    main()
