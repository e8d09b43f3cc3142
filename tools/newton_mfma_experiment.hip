// NOT COMPILED -- the source of a round-5 experiment, kept for the record (profiles/r5_newton_mfma_four_pivot.txt).
// It was a device function of pymbar_amd/csrc/mbar_k_solver.hip (same includes, same helpers: gram_elem, newton_tail, recip_fast,
// wave_max, AdaptArgs) called from k_select_newton in place of newton_body<16, 8>, parity-green on seven problems (64 .. 127
// unknowns, unsampled states, group sizes 1 .. 4) to 3e-15 in f -- and no faster than the register version: the step is a chain
// of dependent fp64 instructions at ~45 clocks each on a lone wave, not arithmetic.
// ---------------------------------------------------------------------------------------------
// The same gauge-fixed Gauss-Jordan solve for 64 .. 127 unknowns with the eliminations on the fp64 MATRIX cores: FOUR pivots per
// barrier, the rank-4 update of every live 16 x 16 block one v_mfma_f64_16x16x4 (the two-pivot register version above spends
// ~870 clocks per step on ~180 vector FMAs per thread and ~900 on its latency chain, 64 steps; here a step is one chain and at
// most sixteen matrix instructions per wave, 32 steps).
//   Layout: the augmented matrix [A | b] (128 x 128, b in column 127, row 127 padding) lives in the ACCUMULATOR layout of the
//   matrix instruction -- block (I, J), lane l, register r <-> row 16 I + 4 r + (l >> 4), column 16 J + (l & 15); wave w owns block
//   rows 2 w and 2 w + 1 (sixteen blocks, 128 registers).
//   Step (pivots j0 .. j0 + 3, all inside block column J0 = j0 / 16): the four pivot columns are published through LDS (their
//   owners are the lanes with (l & 15) in [j0 % 16, + 4) of block column J0; b_j travels in slot 127 as in the register version);
//   one barrier; every lane factors the 4 x 4 pivot block P = L D L^T for itself (d_k are exactly the pivots of the sequential
//   elimination: recorded, and compared with the threshold) and solves P x = e_k for ITS operand column k = l >> 4; the
//   A operand is -M with M = C P^-1 (rows of the pivot block: I - D P^-1, which leaves them as d_q (P^-1 R)_q, diagonal in the
//   pivot columns); the B operand is the pivot rows R = C^T (symmetry of the live block; columns left of the pivots take
//   garbage and are never read again); acc(I, J) += (-M_I) R_J for J >= J0.  x_i = b_i / d_i at the end: no triangular solves.
//   A last group with fewer than four pivots (M not a multiple of 4, or index 127) masks the pivot block to the identity there.
// ---------------------------------------------------------------------------------------------
typedef double v4d_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double gram_elem_lds(const double* g, int nb, int ki, int kj) {
    if (ki > kj) {
        const int t = ki;
        ki = kj;
        kj = t;
    }
    const int I = ki >> 4, J = kj >> 4;
    const int b = I * nb - (I * (I - 1)) / 2 + (J - I);
    return g[b * (16 * 17) + (ki & 15) * 17 + (kj & 15)];  // (SELECT_GRAM_PITCH = 17, defined with the selection below)
}
__device__ __forceinline__ void newton_body_mfma(const AdaptArgs& q, const double* gram_lds, long long* st = nullptr) {  // 256 threads
    constexpr int NC = 128;
    __shared__ double colbuf[2][4][NC];  // [parity of the step][pivot column k][row]; [k][127] = b_{j0 + k}
    __shared__ double pv[NC], rh[NC], xs[NC + 1];
    __shared__ double s_f[128], s_ps[128], s_nk[128], s_ln[128];
    __shared__ double s_cc[128], s_a0[128];
    __shared__ int smp[NC + 1], pos[128];
    if (q.ctl[CTL_DONE] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lr = lane >> 4;
    const int M = q.m - 1, nb = q.Kp / 16;
    for (int k = tid; k < q.Kp; k += 256) {
        s_f[k] = k < q.K ? q.f[k] : 0.0;
        s_ps[k] = q.psum[k];
        s_nk[k] = q.Nk[k];
        s_ln[k] = q.lnNk[k];
        s_cc[k] = q.pmode ? (q.fused ? q.cgram[k] : q.ccur[k]) : 1.0;
        s_a0[k] = q.pmode ? q.a0[k] : 0.0;
        pos[k] = 0;
    }
    for (int i = tid; i < q.m; i += 256) smp[i] = q.sampled[i];
    __syncthreads();
    for (int i = tid; i < q.m; i += 256) pos[smp[i]] = i;

    // ---- tile set-up in the accumulator layout
    v4d_t acc[2][8];
    {
        int kj[8];
        double cj[8];
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            const int k = 16 * J + lc;
            kj[J] = k < M ? smp[k + 1] : 0;
            cj[J] = s_cc[kj[J]];
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (2 * w + ii) + 4 * r + lr;
                const int ki = i < M ? smp[i + 1] : 0;
                const double ci = s_cc[ki], psi = s_ps[ki], gi = psi - s_nk[ki];
#pragma unroll
                for (int J = 0; J < 8; ++J) {
                    const int k = 16 * J + lc;
                    double v = -(gram_lds ? gram_elem_lds(gram_lds, nb, ki, kj[J]) : gram_elem(q.gram_red, nb, ki, kj[J]));
                    if (q.pmode) v *= ci * cj[J];
                    if (i < M) {
                        if (k < M) {
                            if (i == k) v += psi;
                        } else {
                            v = (k == NC - 1) ? gi : 0.0;
                        }
                    } else {
                        v = (i == k && k != NC - 1) ? 1.0 : 0.0;
                    }
                    acc[ii][J][r] = v;
                }
            }
        }
    }
    double pmax = 0.0;
    {
        __shared__ double s_pmax[4];
        double v = 0.0;
        for (int i = tid; i < q.m; i += 256) v = fmax(v, s_ps[smp[i]]);
        v = wave_max(v);
        if (lane == 0) s_pmax[w] = v;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; ++x) pmax = fmax(pmax, s_pmax[x]);
    }
    const double piv_thr = pmax * 2.220446049250313e-16 * (double)(M > 0 ? M : 1);
    if (st && tid == 0) st[2] = clock64();

    // ---- elimination
    bool bad = false;
    int par = 0;
    const double e0 = lr == 0 ? 1.0 : 0.0, e1 = lr == 1 ? 1.0 : 0.0, e2 = lr == 2 ? 1.0 : 0.0, e3 = lr == 3 ? 1.0 : 0.0;
#pragma unroll
    for (int J0 = 0; J0 < 8; ++J0) {
        if (16 * J0 < M) {
            for (int jl = 0; jl < 16; jl += 4) {
                const int j0 = 16 * J0 + jl;
                if (j0 >= M) break;
                const int npiv = M - j0 < 4 ? M - j0 : 4;
                long long* sp = (st && tid == 0 && J0 == 1 && jl == 4) ? q.stamps + 8 * 64 : nullptr;  // (debug: phases of one step)
                if (sp) sp[0] = clock64();
                double(*cb)[NC] = colbuf[par];
                par ^= 1;
                // 1. the pivot columns as they stand (slot 127 belongs to b)
                const int cl = lc - jl;
                if (cl >= 0 && cl < 4) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * (2 * w + ii) + 4 * r + lr;
                            if (i != NC - 1) cb[cl][i] = acc[ii][J0][r];
                        }
                }
                if (w == (J0 >> 1) && lc == 15) {  // b_{j0 + k}: block (J0, 7), register jl / 4, lanes 15 + 16 k
                    const v4d_t t = acc[J0 & 1][7];
                    cb[lr][NC - 1] = jl == 0 ? t[0] : (jl == 4 ? t[1] : (jl == 8 ? t[2] : t[3]));
                }
                __syncthreads();
                if (sp) sp[1] = clock64();
                // 2. the 4 x 4 pivot block P = [[A, B^T], [B, C]] in 2 x 2 blocks (identity where the group has no pivot): two
                // reciprocals deep (det A, then the determinant of the Schur complement S = C - B A^-1 B^T) instead of the four of a
                // scalar L D L^T; the pivots of the sequential elimination are d = (a00, det A / a00, s00, det S / s00).
                // Everything the lane reads from the published columns is requested first: the factorisation below is a latency chain.
                double p00 = cb[0][j0], p10 = cb[0][j0 + 1], p20 = cb[0][j0 + 2], p30 = cb[0][j0 + 3];
                double p11 = cb[1][j0 + 1], p21 = cb[1][j0 + 2], p31 = cb[1][j0 + 3];
                double p22 = cb[2][j0 + 2], p32 = cb[2][j0 + 3], p33 = cb[3][j0 + 3];
                double cr[2][4], rb[8];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int k = 0; k < 4; ++k) cr[ii][k] = cb[k][16 * (2 * w + ii) + lc];
#pragma unroll
                for (int J = J0; J < 8; ++J) rb[J] = cb[lr][16 * J + lc];
                if (npiv < 4) {
                    p30 = p31 = p32 = 0.0;
                    p33 = 1.0;
                    if (npiv < 3) {
                        p20 = p21 = 0.0;
                        p22 = 1.0;
                    }
                    if (npiv < 2) {
                        p10 = 0.0;
                        p11 = 1.0;
                    }
                }
                const double detA = fma(p00, p11, -p10 * p10);
                const double iA = recip_fast(detA), i00 = recip_fast(p00);
                // W = B A^-1
                const double w00 = fma(p20, p11, -p21 * p10) * iA, w01 = fma(p21, p00, -p20 * p10) * iA;
                const double w10 = fma(p30, p11, -p31 * p10) * iA, w11 = fma(p31, p00, -p30 * p10) * iA;
                const double s00 = fma(-w01, p21, fma(-w00, p20, p22));
                const double s10 = fma(-w11, p21, fma(-w10, p20, p32));
                const double s11 = fma(-w11, p31, fma(-w10, p30, p33));
                const double detS = fma(s00, s11, -s10 * s10);
                const double iS = recip_fast(detS), is0 = recip_fast(s00);
                const double d0 = p00, d1 = detA * i00, d2 = s00, d3 = detS * is0;
                if (tid == 0) {
                    pv[j0] = d0;
                    if (npiv > 1) pv[j0 + 1] = d1;
                    if (npiv > 2) pv[j0 + 2] = d2;
                    if (npiv > 3) pv[j0 + 3] = d3;
                }
                // (padding pivots are exactly 1: the test passes for them)
                if (!(d0 > piv_thr) || !isfinite(d0) || !(d1 > piv_thr || npiv < 2) || !isfinite(d1) || !(d2 > piv_thr || npiv < 3) ||
                    !isfinite(d2) || !(d3 > piv_thr || npiv < 4) || !isfinite(d3))
                    bad = true;  // the same in every thread
                if (sp) sp[2] = (long long)(d3 != 123.0) * 0 + clock64();
                // column lr of P^-1 by block elimination: r = e_hi - W e_lo, x_hi = S^-1 r, x_lo = A^-1 e_lo - W^T x_hi
                const double r2 = fma(-w01, e1, fma(-w00, e0, e2)), r3 = fma(-w11, e1, fma(-w10, e0, e3));
                const double x2 = fma(s11, r2, -s10 * r3) * iS, x3 = fma(s00, r3, -s10 * r2) * iS;
                const double x0 = fma(-w10, x3, fma(-w00, x2, fma(p11, e0, -p10 * e1) * iA));
                const double x1 = fma(-w11, x3, fma(-w01, x2, fma(p00, e1, -p10 * e0) * iA));
                if (sp) sp[3] = (long long)(x0 != 123.0) * 0 + clock64();
                // 3. A operand: -M, lane (row lc of the block, pivot lr)
                double am[2];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int row = 16 * (2 * w + ii) + lc;
                    double m = fma(cr[ii][3], x3, fma(cr[ii][2], x2, fma(cr[ii][1], x1, cr[ii][0] * x0)));
                    const int qd = row - j0;  // pivot row q of this group?
                    if (qd >= 0 && qd < npiv) {
                        const double dq = qd == 0 ? d0 : (qd == 1 ? d1 : (qd == 2 ? d2 : d3));
                        const double xq = qd == 0 ? x0 : (qd == 1 ? x1 : (qd == 2 ? x2 : x3));
                        m = fma(-dq, xq, qd == lr ? 1.0 : 0.0);
                    }
                    if (lr >= npiv) m = 0.0;
                    am[ii] = -m;
                }
                if (sp) sp[4] = (long long)(am[0] != 123.0) * 0 + (long long)(am[1] != 123.0) * 0 + clock64();
                // 4. B operand (pivot rows by symmetry; column 127 = b) and the rank-4 updates
#pragma unroll
                for (int J = J0; J < 8; ++J) {
                    acc[0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0], rb[J], acc[0][J], 0, 0, 0);
                    acc[1][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1], rb[J], acc[1][J], 0, 0, 0);
                }
                if (sp) sp[5] = clock64();
                if (sp) sp[6] = (long long)(acc[1][7][0] != 123.0) * 0 + clock64();
            }
        }
    }
    if (st && tid == 0) st[3] = clock64();
    if (lc == 15) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) rh[16 * (2 * w + ii) + 4 * r + lr] = acc[ii][7][r];
    }
    __syncthreads();
    if (tid == 0) xs[0] = 0.0;
    if (tid < M) xs[tid + 1] = rh[tid] / pv[tid];
    __syncthreads();
    if (st && tid == 0) st[4] = clock64();
    newton_tail<256>(q, xs, bad, s_f, s_ps, s_nk, s_ln, s_a0, smp, pos, tid);
}

