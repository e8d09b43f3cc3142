// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path.
//
// Data layout.  u is the (Kp x ld) row-major fp64 matrix of reduced potentials u[k][n] of this
// rank's column shard (Kp = padded state count, ld = N rounded up to 16; padding is zero-filled and
// masked).  The fast kernels process "wave tiles" of 16 consecutive samples x all states: one
// 128-byte line per state row, staged into a wave-private LDS buffer by LDS-DMA
// (global_load_lds_dwordx4, 8 rows per instruction).  Inside LDS, row k is stored rotated by
// (k & 14) doubles so that the MFMA-operand read -- lane l holds state 16*I + (l & 15) of sample
// 4*g + (l >> 4), the A/B layout of v_mfma_f64_16x16x4_f64 -- is a conflict-free ds_read_b64.
// In that layout
//   * the per-sample reduction over states (log-sum-exp denominator, mbar_solvers.py:238) is an
//     in-register reduction over the block index I plus a 16-lane DPP butterfly,
//   * the per-state reduction over samples (numerator sums, mbar_solvers.py:240-241) is a plain
//     per-lane accumulation, and
//   * the K x K contraction W^T W of the Hessian (mbar_solvers.py:407) is
//     acc[I][J] += mfma_f64_16x16x4(p[I], p[J]) with no data movement at all.
// Waves never synchronise with each other; every wave streams its own tiles (tile index strided by
// the number of waves in the grid) with a one-tile DMA prefetch.
#include "mbar_internal.h"

#include <hip/hip_ext.h>
#include <math.h>
#include <type_traits>

namespace mbar {

typedef double v4d __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// All-reduce over the 16 lanes of a DPP row (= the 16 states a sample has in one register):
// quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror.
__device__ __forceinline__ double row16_max(double x) {
    x = fmax(x, dpp_move<0xB1>(x));
    x = fmax(x, dpp_move<0x4E>(x));
    x = fmax(x, dpp_move<0x141>(x));
    x = fmax(x, dpp_move<0x140>(x));
    return x;
}
// Lane N of every 16-lane row to all lanes of that row (gfx90a+: the one DPP control 64-bit moves accept).
template <int N>
__device__ __forceinline__ double row16_bcast(double x) {
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(N));
    return r;
}
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_move<0xB1>(x);
    x += dpp_move<0x4E>(x);
    x += dpp_move<0x141>(x);
    x += dpp_move<0x140>(x);
    return x;
}
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = fmax(x, __shfl_xor(x, m));
    return x;
}

// exp built for instruction count (gfx950 issues one fp64 VALU op per ~5.6 cycles per SIMD and fp64 MFMA does
// not overlap with VALU -- profiles/r1_*microbench.txt -- so every fp64 instruction here is kernel time).
// 2^(t/S) for t = S x log2(e), S = 2^EXP2_BITS = 2048:   s = rint(max(t, EXP2_CLAMP));  z = t - s in [-1/2, 1/2];
//   j = s & (S-1), q = s >> EXP2_BITS;   result = ldexp(T[j] * P(z), q),  T[j] = 2^(j/S) from a 16 KB LDS table at
//   LDS offset 0, P = degree-3 polynomial of 2^(z/S) (error 9e-18; tools/gen_exp2_table.py).  Nine fp64 + three
//   integer instructions (the library exp needs ~25).  exp(-inf) = 0 through the clamp, overflow gives inf through
//   ldexp; NaN arguments are laundered to 0 by the clamp, which is why NaN / -inf entries of u_kn and non-finite f_k
//   are detected at the boundary instead (mbar_capi.cpp).
#include "exp2_table.inc"
#include "log_table.inc"
constexpr double LOG2E = 0x1.71547652b82fep+0, LN2 = 0x1.62e42fefa39efp-1;
constexpr double EXP2_S = (double)(1 << EXP2_BITS);
constexpr double LOG2E_S = EXP2_S * LOG2E, LN2_OVER_S = LN2 / EXP2_S;
constexpr double EXP2_CLAMP = -1100.0 * EXP2_S;
constexpr int EXP2_TABLE_BYTES = (1 << EXP2_BITS) * 8;
constexpr int LOG_TABLE_BYTES = 256 * 8;
constexpr int EXP_TABLE_BYTES = EXP2_TABLE_BYTES + LOG_TABLE_BYTES;  // LDS reserved for both look-up tables
typedef __attribute__((address_space(3))) const double lds_cdouble;
// Every thread block copies the table to LDS offset 0 (its dynamic LDS starts there: the kernels have no static
// __shared__), so a look-up address is just the masked integer -- no base add.  Callers barrier afterwards.
__device__ __forceinline__ void exp_table_init(char* smem) {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    for (int i = threadIdx.x; i < (1 << EXP2_BITS); i += blockDim.x) reinterpret_cast<double*>(smem)[i] = EXP2_TABLE[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        reinterpret_cast<double*>(smem + EXP2_TABLE_BYTES)[i] = LOG_TABLE[i];
}
__device__ __forceinline__ double exp2_table_at(int si) {
    return *(lds_cdouble*)(uintptr_t)(uint32_t)((si << 3) & (EXP2_TABLE_BYTES - 8));
}
__device__ __forceinline__ double exp2_poly(double z) {
    double p = EXP2_POLY[EXP2_DEG];
#pragma unroll
    for (int k = EXP2_DEG - 1; k >= 0; --k) p = fma(p, z, EXP2_POLY[k]);
    return p;
}
__device__ __forceinline__ double exp2s_fast(double ts) {  // 2^(ts / S)
    const double t = fmax(ts, EXP2_CLAMP);
    const double s = __builtin_rint(t);
    const double z = t - s;
    const int si = (int)s;
    const double T = exp2_table_at(si);
    return ldexp(T * exp2_poly(z), si >> EXP2_BITS);
}
// log s for positive finite s (the per-sample sums of the evaluation sweep), 12 fp64 + 2 integer instructions
// (the library log is ~50 and keeps a dozen constants in registers):  s = 2^e m, m in [1/2, 1); the top 7 mantissa
// bits pick c_j with |m / c_j - 1| <= 2^-8 from the 2 KB LDS table behind the exp table (tools/gen_log_table.py);
// log s = e ln2 + log c_j + log1p(r), r = m / c_j - 1, log1p by its degree-6 Taylor polynomial (error 2e-18).
// Absolute error ~2e-16 (it is added to a shift of order one or more).  s = 0 / negative are not supported.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double log_pos(double s) {
    const double m = __builtin_amdgcn_frexp_mant(s);
    const double ed = (double)__builtin_amdgcn_frexp_exp(s);
    const uint32_t off = (uint32_t)(__double2hiint(m) >> 9) & 0x7f0u;
    const v2d tc = *(__attribute__((address_space(3))) const v2d*)(uintptr_t)(uint32_t)(EXP2_TABLE_BYTES + off);
    const double r = fma(m, tc.x, -1.0);
    double q = -1.0 / 6.0;
    q = fma(q, r, 0.2);
    q = fma(q, r, -0.25);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    q = fma(q, r, 1.0);
    return fma(ed, LN2, tc.y) + q * r;
}
// The same exp for N independent arguments, written as three stages separated by scheduling barriers: with one
// or two waves per SIMD nothing else hides the LDS latency of the table look-up, and hipcc's own schedule leaves
// only a handful of instructions between each ds_read and its use.  Stage 1 issues all N table reads, stage 2 (the
// polynomials) runs while they are in flight, stage 3 combines.  x[] in: ts, out: 2^(ts/S).
// CLAMP = false: the arguments are known to be finite (callers substitute finite sentinels for their -inf cases and the
// matrix holds no +inf): one v_max_f64 per element less.  Finite arguments of any size are safe without the clamp --
// v_cvt_i32_f64 saturates, so a hugely negative argument ends in ldexp(..., -2^20) = 0.
template <int N, bool CLAMP = true>
__device__ __forceinline__ void exp2s_batch(double (&x)[N]) {
#ifdef MBAR_EXPERIMENT_NOEXP  // timing experiment only (profiles/r1_noexp_experiment.txt): results are garbage
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = fma(x[i], 1e-9, 1.0);
    return;
#endif
    double T[N];
    int q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t = CLAMP ? fmax(x[i], EXP2_CLAMP) : x[i];
        const double s = __builtin_rint(t);
        x[i] = t - s;
        const int si = (int)s;
        q[i] = si >> EXP2_BITS;
        T[i] = exp2_table_at(si);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = exp2_poly(x[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = ldexp(T[i] * x[i], q[i]);
}
template <int N>
__device__ __forceinline__ void exp2s_batch2(double (&x0)[N], double (&x1)[N]) {  // two argument sets, one pipeline
#ifdef MBAR_EXPERIMENT_NOEXP
#pragma unroll
    for (int i = 0; i < N; ++i) { x0[i] = fma(x0[i], 1e-9, 1.0); x1[i] = fma(x1[i], 1e-9, 1.0); }
    return;
#endif
    double T0[N], T1[N];
    int q0[N], q1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t0 = fmax(x0[i], EXP2_CLAMP), t1 = fmax(x1[i], EXP2_CLAMP);
        const double s0 = __builtin_rint(t0), s1 = __builtin_rint(t1);
        x0[i] = t0 - s0;
        x1[i] = t1 - s1;
        const int i0 = (int)s0, i1 = (int)s1;
        q0[i] = i0 >> EXP2_BITS;
        q1[i] = i1 >> EXP2_BITS;
        T0[i] = exp2_table_at(i0);
        T1[i] = exp2_table_at(i1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x0[i] = exp2_poly(x0[i]);
        x1[i] = exp2_poly(x1[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x0[i] = ldexp(T0[i] * x0[i], q0[i]);
        x1[i] = ldexp(T1[i] * x1[i], q1[i]);
    }
}
// 1 / s for s > 0: hardware estimate + two Newton steps (the divide expansion costs twice as many instructions)
__device__ __forceinline__ double recip_fast(double s) {
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    return r;
}

// In-lane reductions over the NB registers of a sample as a pairwise tree (dependent depth log2 NB instead of NB:
// with one or two waves per SIMD the chain latency of fp64 ops is exposed).
template <int NB>
__device__ __forceinline__ double tree_max(const double (&x)[NB]) {
    double t[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) t[i] = x[i];
#pragma unroll
    for (int n = NB; n > 1; n -= n / 2) {
#pragma unroll
        for (int i = 0; i < n / 2; ++i) t[i] = fmax(t[i], t[n - 1 - i]);
    }
    return t[0];
}
template <int NB>
__device__ __forceinline__ double tree_sum(const double (&x)[NB]) {
    double t[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) t[i] = x[i];
#pragma unroll
    for (int n = NB; n > 1; n -= n / 2) {
#pragma unroll
        for (int i = 0; i < n / 2; ++i) t[i] += t[n - 1 - i];
    }
    return t[0];
}
// sum_i x_i c_i as two interleaved FMA chains (NB + 1 instructions; a product array + tree_sum takes 2 NB - 1)
template <int NB>
__device__ __forceinline__ double dot_sum(const double (&x)[NB], const double (&c)[NB]) {
    if constexpr (NB == 1) {
        return x[0] * c[0];
    } else {
        double e = x[0] * c[0], o = x[1] * c[1];
#pragma unroll
        for (int i = 2; i + 1 < NB; i += 2) {
            e = fma(x[i], c[i], e);
            o = fma(x[i + 1], c[i + 1], o);
        }
        if constexpr (NB & 1) e = fma(x[NB - 1], c[NB - 1], e);
        return e + o;
    }
}
// 16-lane all-reduce of two independent values at once (the two dependency chains interleave)
__device__ __forceinline__ void row16_max2(double& a, double& b) {
    a = fmax(a, dpp_move<0xB1>(a));   b = fmax(b, dpp_move<0xB1>(b));
    a = fmax(a, dpp_move<0x4E>(a));   b = fmax(b, dpp_move<0x4E>(b));
    a = fmax(a, dpp_move<0x141>(a));  b = fmax(b, dpp_move<0x141>(b));
    a = fmax(a, dpp_move<0x140>(a));  b = fmax(b, dpp_move<0x140>(b));
}
__device__ __forceinline__ void row16_sum2(double& a, double& b) {
    a += dpp_move<0xB1>(a);   b += dpp_move<0xB1>(b);
    a += dpp_move<0x4E>(a);   b += dpp_move<0x4E>(b);
    a += dpp_move<0x141>(a);  b += dpp_move<0x141>(b);
    a += dpp_move<0x140>(a);  b += dpp_move<0x140>(b);
}
// Log-sum-exp step for TWO 4-sample groups of a tile at once (independent chains interleaved):
//   x = a - u;  m2 = 32 log2(e) max_k x;  e = 2^((32 log2(e) x - m2)/32);  s = sum_k e;  acc[0] += e / s
// A second candidate f' costs no second exp: exp(a'_k - u_kn - m) = e_kn * c_k with the per-state constant
// c_k = exp(a'_k - a_k), so  e' = e c,  s' = sum_k e',  acc[1] += e' / s'  (3 fp64 ops per element instead of ~20).
// logden_f = m2 ln2/32 + log s_f for both candidates (same shift m2).
// logden_n = shift + log(sum) for every candidate with ONE log per tile: all 16 lanes of a DPP row hold the sums of
// all candidates of their sample, so lanes ks in [4f, 4f+4) evaluate candidate f ((ks & 3) = the group whose
// sample this lane kept).  objl accumulates this lane's objective terms; its candidate is (ks >> 2).
template <int NF>
__device__ __forceinline__ void logden_out(double mm, const double (&ss)[NF], int ks, bool sample_ok, int64_t n,
                                           double wn, double* __restrict__ logden0, double* __restrict__ logden1,
                                           const double* __restrict__ dn, double& objl) {
    const bool second = NF == 2 && (ks & 4);
    const double s_first = ss[0], s_second = ss[NF - 1];  // (scalars: a select on ss[] itself becomes a scratch array)
    const double ldv = fma(mm, LN2_OVER_S, log_pos(second ? s_second : s_first));
    if (sample_ok && ks < 4 * NF) {
        double* out = second ? logden1 : logden0;
        if (out) out[n] = ldv;
        objl = fma(wn, dn ? (ldv - dn[n]) : ldv, objl);
    }
}
template <int NF>
__device__ __forceinline__ void objective_out(double objl, int ks, int lane, double* __restrict__ obj_part, int64_t rec) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double o = wave_sum((ks >> 2) == f ? objl : 0.0);
        if (lane == 0) obj_part[rec * NF + f] = o;
    }
}
template <int NB>
__device__ __forceinline__ void lse_load2(const char* cbuf, int rd0, int rd1, const double (&a)[NB],
                                          double (&x0)[NB], double (&x1)[NB]) {
    // all 2 NB reads are issued before the first use: left to itself hipcc interleaves the subtractions with the reads in
    // three batches, and every batch ends in an s_waitcnt that exposes a full LDS round trip to the lone wave
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd0);
        x1[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = a[I] - x0[I];
        x1[I] = a[I] - x1[I];
    }
}
template <int NB, int NF, bool SPLIT = false>
__device__ __forceinline__ void lse_math2(double (&x0)[NB], double (&x1)[NB], const double (&c)[NB],
                                          double (&acc)[NF][NB], double w0, double w1, double& m2_0, double& m2_1,
                                          double (&s0)[NF], double (&s1)[NF]) {
    double m0 = tree_max<NB>(x0), m1 = tree_max<NB>(x1);
    row16_max2(m0, m1);
    m2_0 = m0 * LOG2E_S;
    m2_1 = m1 * LOG2E_S;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = fma(x0[I], LOG2E_S, -m2_0);
        x1[I] = fma(x1[I], LOG2E_S, -m2_1);
    }
    if constexpr (SPLIT) {  // (register budget of two waves per SIMD: half the look-ups in flight at a time)
        exp2s_batch<NB>(x0);
        exp2s_batch<NB>(x1);
    } else {
        exp2s_batch2<NB>(x0, x1);
    }
    // Second candidate: e'_k = e_k c_k.  Only its sum needs the products (an FMA dot instead of NB multiplies + a
    // tree of adds); the per-state accumulator takes the UNSCALED e_k r' and the constant c_k is applied once to
    // the reduced sums on the host (mbar_capi.cpp: eval_core).
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        s0[f] = f == 0 ? tree_sum<NB>(x0) : dot_sum<NB>(x0, c);
        s1[f] = f == 0 ? tree_sum<NB>(x1) : dot_sum<NB>(x1, c);
        row16_sum2(s0[f], s1[f]);
        const double r0 = w0 * recip_fast(s0[f]), r1 = w1 * recip_fast(s1[f]);  // w: sample multiplicity (0 on padding)
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = fma(x1[I], r1, fma(x0[I], r0, acc[f][I]));
    }
}
template <int NB, int NF>
__device__ __forceinline__ void lse_two_groups(const char* cbuf, int rd0, int rd1,
                                               const double (&a)[NB], const double (&c)[NB], double (&acc)[NF][NB],
                                               double w0, double w1, double& m2_0, double& m2_1,
                                               double (&s0)[NF], double (&s1)[NF]) {
    double x0[NB], x1[NB];
    lse_load2<NB>(cbuf, rd0, rd1, a, x0, x1);
    lse_math2<NB, NF>(x0, x1, c, acc, w0, w1, m2_0, m2_1, s0, s1);
}

// Single-group versions for wide panels (NB > 8), where two groups in the exp pipeline at once would spill registers.
template <int NB>
__device__ __forceinline__ void lse_load1(const char* cbuf, int rd0, const double (&a)[NB], double (&x0)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) x0[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int I = 0; I < NB; ++I) x0[I] = a[I] - x0[I];
}
template <int NB, int NF>
__device__ __forceinline__ void lse_math1(double (&x0)[NB], const double (&c)[NB], double (&acc)[NF][NB], double w0,
                                          double& m2_0, double (&s0)[NF]) {
    m2_0 = row16_max(tree_max<NB>(x0)) * LOG2E_S;
#pragma unroll
    for (int I = 0; I < NB; ++I) x0[I] = fma(x0[I], LOG2E_S, -m2_0);
    exp2s_batch<NB>(x0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        s0[f] = row16_sum(f == 0 ? tree_sum<NB>(x0) : dot_sum<NB>(x0, c));
        const double r0 = w0 * recip_fast(s0[f]);
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = fma(x0[I], r0, acc[f][I]);
    }
}
template <int NB, int NF>
__device__ __forceinline__ void lse_one_group(const char* cbuf, int rd0, const double (&a)[NB],
                                              const double (&c)[NB], double (&acc)[NF][NB], double w0, double& m2_0,
                                              double (&s0)[NF]) {
    double x0[NB];
    lse_load1<NB>(cbuf, rd0, a, x0);
    lse_math1<NB, NF>(x0, c, acc, w0, m2_0, s0);
}
// Two consecutive groups g, g+1 of a tile; (mm, ss[]) capture the (shift, sums) of the sample this lane will write.
template <int NB, int NF>
__device__ __forceinline__ void lse_group_pair(const char* cbuf, const char* wslot, int rd_base,
                                               const int (&pos)[GROUPS], int g, const double (&a)[NB],
                                               const double (&c)[NB], double (&acc)[NF][NB], int ks, int ns,
                                               double& mm, double (&ss)[NF]) {
    double m2a, m2b, sa[NF], sb[NF];
    // per-sample multiplicities (1 for plain data, bootstrap counts otherwise, 0 on the padding) from the tile's slot
    const double va = *reinterpret_cast<const double*>(wslot + (4 * g + ns) * 8);
    const double vb = *reinterpret_cast<const double*>(wslot + (4 * (g + 1) + ns) * 8);
    if constexpr (NB <= 8) {
        lse_two_groups<NB, NF>(cbuf, rd_base + pos[g], rd_base + pos[g + 1], a, c, acc, va, vb, m2a, m2b, sa, sb);
    } else {
        lse_one_group<NB, NF>(cbuf, rd_base + pos[g], a, c, acc, va, m2a, sa);
        lse_one_group<NB, NF>(cbuf, rd_base + pos[g + 1], a, c, acc, vb, m2b, sb);
    }
    if ((ks & 3) == g) {
        mm = m2a;
#pragma unroll
        for (int f = 0; f < NF; ++f) ss[f] = sa[f];
    }
    if ((ks & 3) == g + 1) {
        mm = m2b;
#pragma unroll
        for (int f = 0; f < NF; ++f) ss[f] = sb[f];
    }
}

// Stage one wave tile (ROWS state rows x 16 samples starting at column n0) into `dst`.
// DMA instruction j fills LDS bytes [1024 j, 1024 j + 1024): lane l -> row 8j + (l >> 3),
// positions 2(l & 7), 2(l & 7)+1 of that row, which hold samples (pos - (row & 14)) & 15.
// The per-lane part of the source address depends on j only through its parity (row & 14 = ((l >> 3) & 6) |
// 8 (j & 1)), so it is two loop-invariant 32-bit byte offsets (StageOffsets) added to a wave-uniform base:
// the DMA is issued as `global_load_lds_dwordx4 voff, s[base]` with no per-instruction VALU address math.
// rowmap(tile_row) gives the global row (8-row groups never straddle a panel).
// WIDE: the row pitch is so large (N_local >= 7.6e7) that 7 ld 8 + 120 does not fit 32 bits; the lane offsets are then
// 64-bit and every DMA pays one 64-bit VALU add (own kernel instantiations, selected by the launchers).
template <bool WIDE>
struct StageOffsetsT {
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type off_t;
    off_t off[2];
};
typedef StageOffsetsT<false> StageOffsets;
template <bool WIDE = false>
__device__ __forceinline__ StageOffsetsT<WIDE> make_stage_offsets(int64_t ld, int lane) {
    StageOffsetsT<WIDE> so;
    const int r = lane >> 3, pos = 2 * (lane & 7);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int smp = (pos - ((r & 6) | (par << 3))) & 15;
        so.off[par] = (typename StageOffsetsT<WIDE>::off_t)(((int64_t)r * ld + smp) * 8);
    }
    return so;
}
// cache-policy bits of the tile loads (aux operand of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1): the default policy measured
// best (profiles/r3_ab_streaming_hints.txt)
#ifndef MBAR_DMA_AUX
#define MBAR_DMA_AUX 0
#endif
template <bool DMA>
__device__ __forceinline__ void stage_piece(const double* __restrict__ ubase /*wave-uniform*/, uint64_t voff,
                                            char* dst /*wave-uniform*/, int lane) {
    const char* src = reinterpret_cast<const char*>(ubase) + voff;
    if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, MBAR_DMA_AUX);
    } else {
        *reinterpret_cast<double2*>(dst + lane * 16) = *reinterpret_cast<const double2*>(src);
    }
}
template <bool DMA>
__device__ __forceinline__ void stage_piece(const double* __restrict__ ubase /*wave-uniform*/, uint32_t voff,
                                            char* dst /*wave-uniform*/, int lane) {
    // Launder the base through an SGPR constraint: loop strength reduction otherwise turns every DMA address of
    // the tile loop into its own 64-bit per-lane induction variable (2 VGPRs + a 64-bit VALU add per instruction
    // per tile) and the instruction loses its scalar-base form.
    // The 32-bit lane offset is laundered too, so that its zero-extension stays in the block of the DMA (instruction
    // selection is per basic block and only matches base + zext(offset) when it sees both).
    uint64_t ub = reinterpret_cast<uint64_t>(ubase);
    asm("" : "+s"(ub));
    asm("" : "+v"(voff));
    const char* src = reinterpret_cast<const char*>(ub) + voff;
    if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, MBAR_DMA_AUX);
    } else {
        *reinterpret_cast<double2*>(dst + lane * 16) = *reinterpret_cast<const double2*>(src);
    }
}
template <int ROWS, bool DMA, int J0, int JSTEP, typename RowMap, typename SO>
__device__ __forceinline__ void stage_tile(const double* __restrict__ u, int64_t ld, int64_t n0, char* dst, int lane,
                                           const SO& so, RowMap rowmap) {
    constexpr int NDMA = ROWS / 8;
#pragma unroll
    for (int j = J0; j < NDMA; j += JSTEP)
        stage_piece<DMA>(u + rowmap(8 * j) * ld + rowmap.cols(j, n0), so.off[j & 1], dst + j * 1024, lane);
}

// Stage the 16 per-sample values v[n0 .. n0+16) (128 bytes) behind a tile: lanes 0..7 move 16 bytes each.
// Going through LDS-DMA (instead of an ordinary VGPR load) keeps hipcc from draining the whole DMA
// prefetch with an s_waitcnt vmcnt(0) at the first use of the loaded register.
template <bool DMA>
__device__ __forceinline__ void stage_vec16(const double* __restrict__ v, int64_t n0, char* dst, int lane) {
    if (lane < 8) stage_piece<DMA>(v + n0, (uint32_t)(lane * 16), dst, lane);
}

// Pin a loaded value into its register *now*: the compiler must place the s_waitcnt for the load here
// (before any DMA is in flight) instead of a conservative vmcnt(0) at the first use inside the loop.
__device__ __forceinline__ void settle(double& x) { asm volatile("" : "+v"(x)); }

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// `live`: bit j set = the 8 rows of DMA piece j matter.  A piece whose rows all carry a per-state constant that makes their
// terms exactly zero (exp(-inf - u): padding rows of the device matrix, states without samples) is requested from the FIRST tile's
// columns instead of the current tile's: the same 1 KB every time (an L2 hit instead of HBM traffic), the same instruction
// stream, the same LDS layout, and whatever lands there is multiplied by zero like the real rows would have been.
struct RowIdentity {
    int64_t row0;
    uint32_t live = 0xffffffffu;
    __device__ __forceinline__ int64_t operator()(int tr) const { return row0 + tr; }
    __device__ __forceinline__ int64_t cols(int j, int64_t n0) const { return ((live >> j) & 1u) ? n0 : 0; }
};
struct RowTwoPanels {
    int64_t row_i0, row_j0;
    int split;
    uint32_t live = 0xffffffffu;
    __device__ __forceinline__ int64_t operator()(int tr) const {
        return tr < split ? row_i0 + tr : row_j0 + (tr - split);
    }
    __device__ __forceinline__ int64_t cols(int j, int64_t n0) const { return ((live >> j) & 1u) ? n0 : 0; }
};
// Live mask of a panel's DMA pieces from the per-state constants of its rows in the Gram / evaluation layout (lane & 15 = state
// within block I): piece 2 I + h is dead when its 8 constants all equal `dead` (-inf exponent constants, zero multipliers).
template <int NB>
__device__ __forceinline__ uint32_t live_piece_mask(const double (&a)[NB], double dead) {
    uint32_t m = 0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        const unsigned long long b = __ballot(a[I] != dead);
        m |= ((b & 0xffull) ? 1u : 0u) << (2 * I);
        m |= ((b & 0xff00ull) ? 1u : 0u) << (2 * I + 1);
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass: per-sample log-sum-exp over states + per-state sums of p_nk, for NF candidates f.
//   aden[0][k] = f_k + ln N_k of the first candidate (-inf for unsampled / padded states)
//   aden[1][k] = c_k = exp(aden'_k - aden_k) of the second candidate relative to the first (NF == 2)
//   logden_n   = log sum_k exp(aden_k - u_kn)                    (mbar_solvers.py:238)
//   p_nk       = exp(aden_k - u_kn - logden_n),  psum_k = sum_n p_nk   (= N_k sum_n W_nk)
// One exp per matrix element in total: e = exp(x - max) is kept in registers, normalised by the reciprocal of
// its sum, and re-used for the second candidate through the per-state ratio c_k.
// ---------------------------------------------------------------------------------------------
// Waves per workgroup of the default sweep: a 16-sample tile of few states is small, so more waves fit into LDS next to
// the tables and the SIMDs get 2-4 waves each to hide the latency of the tile stream (K <= 64 ran one wave per SIMD at
// 4.9 TB/s).
constexpr int lse_waves(int nb) { return nb <= 2 ? 16 : (nb <= 4 ? 8 : (nb <= 8 ? 4 : 2)); }
template <int NB, int NF, bool DMA, bool WIDE>
__global__ void __launch_bounds__(64 * lse_waves(NB))
k_lse(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
      const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
      double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
      double* __restrict__ obj_part, const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {  // device-resident solver loop: stop flag + rotating logden slots (logden0 = base of the three vectors)
        if (ctl[CTL_DONE] != 0) return;
        const int s = ctl[CTL_SLOT];
        logden1 = logden0 + (int64_t)((s + 2) % 3) * slot_stride;
        logden0 = logden0 + (int64_t)((s + 1) % 3) * slot_stride;
    }
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the 16 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    __syncthreads();
    char* buf = smem + EXP_TABLE_BYTES + wave * (2 * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);
    // one global store per tile (lanes ks < 4 write logden0, 4 <= ks < 8 logden1) unless neither vector is wanted
    const bool has_store = logden0 != nullptr || logden1 != nullptr;

    double a[NB], c[NB], acc[NF][NB], objl = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        c[I] = NF == 2 ? aden[ROWS + 16 * I + ks] : 1.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        settle(a[I]);
        if (NF == 2) settle(c[I]);
    }
    rows.live = live_piece_mask<NB>(a, -INFINITY);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = 0.0;
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    int64_t t = gw;
    int cur = 0;
    if constexpr (DMA) {
        if (t < ntiles) {
            stage_tile<ROWS, true, 0, 1>(u, ld, t * TS, buf, lane, so, rows);
            stage_vec16<true>(cw, t * TS, buf + U_BYTES, lane);
        }
    }
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        if constexpr (DMA) {
            const int64_t tn = t + W;
            if (tn < ntiles) {
                char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
                // The next tile is requested in two halves -- even DMA pieces here, odd ones after the first group pair
                // below -- which smooths the request stream of the 1024 waves (measured: -1.5 % on this sweep).
                stage_tile<ROWS, true, 0, 2>(u, ld, tn * TS, nbuf, lane, so, rows);
                stage_vec16<true>(cw, tn * TS, nbuf + U_BYTES, lane);
                constexpr int NEVEN = (ROWS / 8 + 1) / 2 + 1;  // even pieces + the weight slot
                // vmcnt counts stores too (in issue order with the loads on gfx9); the queue here is
                // [even(t)][odd(t)][logden store of tile t - W][even(tn)], and tile t is needed now:
                if (has_store && t != gw)
                    wait_vm<NEVEN + 1>();
                else
                    wait_vm<NEVEN>();
            } else {
                wait_vm<0>();
            }
        } else {
            stage_tile<ROWS, false, 0, 1>(u, ld, t * TS, cbuf, lane, so, rows);
            stage_vec16<false>(cw, t * TS, cbuf + U_BYTES, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        double mm = 0.0, ss[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) ss[f] = 1.0;
        lse_group_pair<NB, NF>(cbuf, cbuf + U_BYTES, rd_base, pos, 0, a, c, acc, ks, ns, mm, ss);
        if constexpr (DMA) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + W < ntiles) stage_tile<ROWS, true, 1, 2>(u, ld, (t + W) * TS, buf + (cur ^ 1) * TILE_BYTES, lane, so, rows);
            __builtin_amdgcn_sched_barrier(0);
        }
        lse_group_pair<NB, NF>(cbuf, cbuf + U_BYTES, rd_base, pos, 2, a, c, acc, ks, ns, mm, ss);
        // lanes with (ks & 3) == g hold (shift, sums) of sample 4 g + ns: one log per candidate per tile
        {
            const int64_t n = t * TS + 4 * (ks & 3) + ns;
            const double wn = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * (ks & 3) + ns) * 8);
            logden_out<NF>(mm, ss, ks, n < N, n, wn, logden0, logden1, dn, objl);
        }
        cur ^= 1;
    }
    // per-wave partial sums: fold the four sample sub-lanes, lanes 0..15 own one state each
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(gw * NF + f) * ROWS + 16 * I + lane] = v;
        }
    }
    objective_out<NF>(objl, ks, lane, obj_part, gw);
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass with early refill (5 <= NB <= 8, LDS-DMA staging).  All LDS operands of a tile are in registers
// by the middle of the tile; the buffer is handed back to the DMA engine right there instead of at the next loop
// top:
//   NBUF = 2 (default): one wave per SIMD, two tile buffers; the refill is the tile after next, so every DMA has
//             ~1.5 tile periods to land instead of 1 (the k_lse loop spends ~20% of its time in s_waitcnt vmcnt).
//   NBUF = 1: two waves per SIMD (8 per workgroup), one buffer each; the refill is the wave's next tile and lands
//             while this wave finishes the tile and its SIMD partner works.
// ---------------------------------------------------------------------------------------------
template <int NB, int NF, int NBUF>
__global__ void __launch_bounds__(NBUF == 1 ? 512 : 256, NBUF == 1 ? 2 : 1)
k_lse_early(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
         const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
         double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
         double* __restrict__ obj_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the 16 sample weights of the tile
    constexpr int NDMA = ROWS / 8 + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    __syncthreads();
    char* buf0 = smem + EXP_TABLE_BYTES + wave * (NBUF * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsets so = make_stage_offsets(ld, lane);
    const bool has_store = logden0 != nullptr || logden1 != nullptr;  // one store instruction per tile

    double a[NB], c[NB], acc[NF][NB], objl = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        c[I] = NF == 2 ? aden[ROWS + 16 * I + ks] : 1.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        settle(a[I]);
        if (NF == 2) settle(c[I]);
    }
    rows.live = live_piece_mask<NB>(a, -INFINITY);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = 0.0;
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    int64_t t = gw;
    int cur = 0;
#pragma unroll
    for (int b = 0; b < NBUF; ++b)
        if (t + b * W < ntiles) {
            stage_tile<ROWS, true, 0, 1>(u, ld, (t + b * W) * TS, buf0 + b * TILE_BYTES, lane, so, rows);
            stage_vec16<true>(cw, (t + b * W) * TS, buf0 + b * TILE_BYTES + U_BYTES, lane);
        }
    for (; t < ntiles; t += W) {
        char* buf = buf0 + cur * TILE_BYTES;
        const char* wslot = buf + U_BYTES;
        // vmcnt counts the logden stores too, in issue order: [tile t][store][tile t + W][store] for two buffers,
        // [tile t][store] for one
        if (NBUF == 2 && t + W < ntiles) {
            if (!has_store || t == gw)
                wait_vm<NDMA>();      // first tile: [tile t][tile t + W]
            else if (t == gw + W)
                wait_vm<NDMA + 1>();  // second: [tile t][tile t + W][store]
            else
                wait_vm<NDMA + 2>();
        } else if (NBUF == 1 && has_store && t != gw) {
            wait_vm<1>();
        } else {
            wait_vm<0>();
        }
        // the weight slot is refilled together with the tile: take what this lane needs from it first
        double w[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) w[g] = *reinterpret_cast<const double*>(wslot + (4 * g + ns) * 8);
        const double wn = *reinterpret_cast<const double*>(wslot + (4 * (ks & 3) + ns) * 8);
        double x0[NB], x1[NB], m2a, m2b, sa[NF], sb[NF], mm = 0.0, ss[NF];
        const int gq = ks & 3;  // this lane keeps (shift, sums) of sample 4 gq + ns for the log below
        auto refill = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS operand of this tile is in registers
            if (t + NBUF * W < ntiles) {
                stage_tile<ROWS, true, 0, 1>(u, ld, (t + NBUF * W) * TS, buf, lane, so, rows);
                stage_vec16<true>(cw, (t + NBUF * W) * TS, buf + U_BYTES, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (NBUF == 1) {
            // the whole tile goes to registers first: the refill then has a full tile period to land
            double x2[NB], x3[NB];
            lse_load2<NB>(buf, pos[0], pos[1], a, x0, x1);
            lse_load2<NB>(buf, pos[2], pos[3], a, x2, x3);
            refill();
            lse_math2<NB, NF, true>(x0, x1, c, acc, w[0], w[1], m2a, m2b, sa, sb);
            mm = gq == 0 ? m2a : m2b;
#pragma unroll
            for (int f = 0; f < NF; ++f) ss[f] = gq == 0 ? sa[f] : sb[f];
            lse_math2<NB, NF, true>(x2, x3, c, acc, w[2], w[3], m2a, m2b, sa, sb);
        } else {
            lse_load2<NB>(buf, pos[0], pos[1], a, x0, x1);
            lse_math2<NB, NF>(x0, x1, c, acc, w[0], w[1], m2a, m2b, sa, sb);
            mm = gq == 0 ? m2a : m2b;
#pragma unroll
            for (int f = 0; f < NF; ++f) ss[f] = gq == 0 ? sa[f] : sb[f];
            lse_load2<NB>(buf, pos[2], pos[3], a, x0, x1);
            refill();
            lse_math2<NB, NF>(x0, x1, c, acc, w[2], w[3], m2a, m2b, sa, sb);
        }
        if (gq >= 2) {
            mm = gq == 2 ? m2a : m2b;
#pragma unroll
            for (int f = 0; f < NF; ++f) ss[f] = gq == 2 ? sa[f] : sb[f];
        }
        {
            const int64_t n = t * TS + 4 * gq + ns;
            logden_out<NF>(mm, ss, ks, n < N, n, wn, logden0, logden1, dn, objl);
        }
        cur ^= NBUF - 1;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(gw * NF + f) * ROWS + 16 * I + lane] = v;
        }
    }
    objective_out<NF>(objl, ks, lane, obj_part, gw);
}

// SCI step on the reduced per-state sums, executed by (at least) 256 threads of ONE workgroup; threads >= 256 only
// take part in the barriers.  part: nparts records of `rows` doubles.  psum: Kp + 256 doubles of LDS, red: 5.
//   f_k <- f_k - log(psum_k / N_k), gauge f_first = 0, aden_k = f_k + ln N_k, delta = max relative change (:627-633)
struct SciArgs {
    const double* Nk;
    const double* lnNk;
    int64_t K, Kp;
    int first;
    double tol;
    double* f;
    double* aden;
    double* f_hist;
    double* delta_out;
};
__device__ __forceinline__ void sci_update_block(const double* part, int64_t nparts, int64_t rows, const SciArgs& q,
                                                 double* psum, double* red) {
    const int tid = threadIdx.x;
    const bool act = tid < 256;
    double* scr = psum + q.Kp;
    const int64_t Kp = q.Kp, K = q.K;
    // all 256 threads share the partial-record sum: thread (g, kk) adds records g, g + G, ... of state kk
    for (int64_t k0 = 0; k0 < Kp; k0 += 256) {
        const int64_t kw = Kp - k0 < 256 ? Kp - k0 : 256;   // states in this pass
        int KW = 1;
        while (KW < kw) KW <<= 1;                            // power of two >= kw, <= 256
        const int G = 256 / KW, g = tid / KW, kk = tid % KW;
        if (act) {
            double sm = 0.0;
            if (kk < kw) {
#pragma unroll 8  // (same summation order; the loads of eight records are in flight together)
                for (int64_t p = g; p < nparts; p += G) sm += part[p * rows + k0 + kk];
            }
            scr[tid] = sm;
        }
        __syncthreads();
        if (tid < kw) {
            double tot = 0.0;
            for (int gg = 0; gg < G; ++gg) tot += scr[gg * KW + tid];
            psum[k0 + tid] = tot;
        }
        __syncthreads();
    }
    if (tid == 0) red[4] = q.f[q.first] - log(psum[q.first] / q.Nk[q.first]);
    __syncthreads();
    const double f0new = red[4];
    double dmax = 0.0;
    const double small = q.tol < 1e-8 ? q.tol : 1e-8;
    if (act)
        for (int64_t k = tid; k < Kp; k += 256) {
            if (k < K && q.Nk[k] > 0.0) {
                const double fo = q.f[k];
                const double fn = fo - log(psum[k] / q.Nk[k]) - f0new;
                q.f[k] = fn;
                q.f_hist[k] = fn;
                q.aden[k] = fn + q.lnNk[k];
                if (k != q.first) {
                    const double div = fabs(fn) < small ? 1.0 : fabs(fn);
                    const double d = fabs(fn - fo) / div;
                    dmax = (d > dmax || d != d) ? d : dmax;  // propagate NaN
                }
            } else {
                q.aden[k] = -INFINITY;
                q.f_hist[k] = k < K ? q.f[k] : 0.0;
            }
        }
    // NaN-propagating max
    double m = dmax;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const double o = __shfl_xor(m, sft);
        m = (o > m || o != o) ? o : m;
    }
    if (act && (tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        double r = red[0];
        for (int w = 1; w < 4; ++w) r = (red[w] > r || red[w] != red[w]) ? red[w] : r;
        *q.delta_out = r;
    }
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass for few states (K <= 32, one candidate): one SAMPLE per lane, all states of that sample in the
// lane's registers.  In the MFMA operand layout of the other kernels a sample's states are spread over 16 lanes, so
// every max / sum over states costs a 4-step DPP butterfly per 4-sample group; with 16-32 states that is more than
// half of the instruction stream (and the matrix cores are not used by this pass anyway).  Here the reductions
// over states are in-register trees and the only cross-lane work is one reduction of the per-state accumulators at
// the end of the kernel.  A tile is 64 consecutive samples x all state rows (512 contiguous bytes per row, LDS row
// k = bytes [512 k, 512 k + 512): the column read of lane n is conflict-free); each LDS-DMA instruction moves two
// rows.  The tile is read into registers in one go, so its single buffer is refilled immediately (two waves per
// SIMD, 8 per workgroup).  Requires a row pitch that is a multiple of 64 (mbar_ctx_create pads it for K <= 32).
// ---------------------------------------------------------------------------------------------
constexpr int TSS = 64;  // samples per tile of the small-K kernel
template <int NB>
__global__ void __launch_bounds__(512, 2)
k_lse_small(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
            const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
            const double* __restrict__ dn, double* __restrict__ psum_part, double* __restrict__ obj_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TSS * 8;
    constexpr int TILE_BYTES = U_BYTES + TSS * 8;  // + the 64 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    exp_table_init(smem);
    __syncthreads();
    char* buf = smem + EXP_TABLE_BYTES + wave * TILE_BYTES;
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    const bool has_store = logden0 != nullptr;
    // DMA instruction j moves rows 2j, 2j+1: lane l -> row 2j + (l >> 5), bytes [16 (l & 31), +16) of its 512
    const uint32_t voff = (uint32_t)(((int64_t)(lane >> 5) * ld + 2 * (lane & 31)) * 8);

    double a[ROWS], acc[ROWS], objl = 0.0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        a[k] = aden[k];
        acc[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) settle(a[k]);
    // pairs of rows whose exponent constants are both -inf (padding, states without samples): requested from the first tile's
    // columns every time -- an L2 hit instead of HBM traffic (see RowIdentity::cols); 5 states in a 16-row matrix: half the bytes
    uint32_t live = 0;
#pragma unroll
    for (int j = 0; j < ROWS / 2; ++j) live |= (__ballot(a[2 * j] != -INFINITY || a[2 * j + 1] != -INFINITY) ? 1u : 0u) << j;

    auto stage = [&](int64_t tile) {
#pragma unroll
        for (int j = 0; j < ROWS / 2; ++j)
            stage_piece<true>(u + (int64_t)(2 * j) * ld + (((live >> j) & 1u) ? tile * TSS : 0), voff, buf + j * 1024, lane);
        if (lane < 32) stage_piece<true>(cw + tile * TSS, (uint32_t)(lane * 16), buf + U_BYTES, lane);
    };

    int64_t t = gw;
    if (t < ntiles) stage(t);
    for (; t < ntiles; t += W) {
        if (has_store && t != gw)
            wait_vm<1>();  // [this tile][logden store of the previous one]: vmcnt counts stores too
        else
            wait_vm<0>();
        double x[ROWS];
        const double w = *reinterpret_cast<const double*>(buf + U_BYTES + lane * 8);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) x[k] = *reinterpret_cast<const double*>(buf + k * (TSS * 8) + lane * 8);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile is in registers: refill its buffer
        if (t + W < ntiles) stage(t + W);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) x[k] = a[k] - x[k];
        const double m = tree_max<ROWS>(x);
        const double m2 = m * LOG2E_S;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) x[k] = fma(x[k], LOG2E_S, -m2);
#pragma unroll
        for (int k0 = 0; k0 < ROWS; k0 += 8) {
            double e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = x[k0 + i];
            exp2s_batch<8>(e);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[k0 + i] = e[i];
        }
        const double ssum = tree_sum<ROWS>(x);
        const double r = w * recip_fast(ssum);  // w: sample multiplicity (0 on the padding)
#pragma unroll
        for (int k = 0; k < ROWS; ++k) acc[k] = fma(x[k], r, acc[k]);
        const double ldv = m + log_pos(ssum);
        const int64_t n = t * TSS + lane;
        if (n < N) {
            if (logden0) logden0[n] = ldv;
            objl = fma(w, dn ? (ldv - dn[n]) : ldv, objl);
        }
    }
    // one partial record per WORKGROUP (the 8 waves are folded through LDS in a fixed order): few enough records for
    // the SCI update kernel to sum directly, which saves the level-1 reduction launch of the device-resident loop
    __syncthreads();  // every wave is done with its tile buffer: re-use the LDS behind the tables
    double* fold = reinterpret_cast<double*>(smem + EXP_TABLE_BYTES);
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) fold[wave * (ROWS + 1) + k] = v;
    }
    const double o = wave_sum(objl);
    if (lane == 0) fold[wave * (ROWS + 1) + ROWS] = o;
    __syncthreads();
    if (threadIdx.x <= ROWS) {
        double tot = 0.0;
        for (int w = 0; w < nwv; ++w) tot += fold[w * (ROWS + 1) + threadIdx.x];
        if (threadIdx.x < ROWS)
            psum_part[(int64_t)blockIdx.x * ROWS + threadIdx.x] = tot;
        else
            obj_part[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass for wide panels (129 <= K <= 256: NB = 12 or 16).  A 16-sample tile is 24-33 KB here, so the
// double-buffered k_lse fits only TWO waves per CU and half the SIMDs idle.  This variant gives every wave ONE tile
// buffer (four waves per CU): groups 0 and 1 are processed straight from LDS, the operands of groups 2 and 3 are
// pulled into registers together, and the buffer is refilled at that point -- half a tile period before it is needed.
// ---------------------------------------------------------------------------------------------
template <int NB, int NF>
__global__ void __launch_bounds__(256, 1)
k_lse_wide(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
           const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
           double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
           double* __restrict__ obj_part, const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {  // device-resident solver loop: stop flag + rotating logden slots (logden0 = base of the three vectors)
        if (ctl[CTL_DONE] != 0) return;
        const int s = ctl[CTL_SLOT];
        logden1 = logden0 + (int64_t)((s + 2) % 3) * slot_stride;
        logden0 = logden0 + (int64_t)((s + 1) % 3) * slot_stride;
    }
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the 16 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    __syncthreads();
    char* buf = smem + EXP_TABLE_BYTES + wave * TILE_BYTES;
    const char* wslot = buf + U_BYTES;
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsets so = make_stage_offsets(ld, lane);
    const bool has_store = logden0 != nullptr || logden1 != nullptr;  // one store instruction per tile

    double a[NB], c[NB], acc[NF][NB], objl = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        c[I] = NF == 2 ? aden[ROWS + 16 * I + ks] : 1.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        settle(a[I]);
        if (NF == 2) settle(c[I]);
    }
    rows.live = live_piece_mask<NB>(a, -INFINITY);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = 0.0;
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    int64_t t = gw;
    if (t < ntiles) {
        stage_tile<ROWS, true, 0, 1>(u, ld, t * TS, buf, lane, so, rows);
        stage_vec16<true>(cw, t * TS, buf + U_BYTES, lane);
    }
    for (; t < ntiles; t += W) {
        if (has_store && t != gw)
            wait_vm<1>();  // [this tile][logden store of the previous one]: vmcnt counts stores too
        else
            wait_vm<0>();
        double w[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) w[g] = *reinterpret_cast<const double*>(wslot + (4 * g + ns) * 8);
        const double wn = *reinterpret_cast<const double*>(wslot + (4 * (ks & 3) + ns) * 8);
        const int gq = ks & 3;  // this lane keeps (shift, sums) of sample 4 gq + ns for the log below
        double x0[NB], x1[NB], m2, sg[NF], mm = 0.0, ss[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) ss[f] = 1.0;
        auto keep = [&](int g) {
            if (gq == g) {
                mm = m2;
#pragma unroll
                for (int f = 0; f < NF; ++f) ss[f] = sg[f];
            }
        };
        lse_load1<NB>(buf, pos[0], a, x0);
        lse_math1<NB, NF>(x0, c, acc, w[0], m2, sg);
        keep(0);
        lse_load1<NB>(buf, pos[1], a, x0);
        lse_math1<NB, NF>(x0, c, acc, w[1], m2, sg);
        keep(1);
        lse_load1<NB>(buf, pos[2], a, x0);
        lse_load1<NB>(buf, pos[3], a, x1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS operand of this tile is in registers
        if (t + W < ntiles) {
            stage_tile<ROWS, true, 0, 1>(u, ld, (t + W) * TS, buf, lane, so, rows);
            stage_vec16<true>(cw, (t + W) * TS, buf + U_BYTES, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        lse_math1<NB, NF>(x0, c, acc, w[2], m2, sg);
        keep(2);
        lse_math1<NB, NF>(x1, c, acc, w[3], m2, sg);
        keep(3);
        {
            const int64_t n = t * TS + 4 * gq + ns;
            logden_out<NF>(mm, ss, ks, n < N, n, wn, logden0, logden1, dn, objl);
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(gw * NF + f) * ROWS + 16 * I + lane] = v;
        }
    }
    objective_out<NF>(objl, ks, lane, obj_part, gw);
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass for 257 .. 512 states in ONE read of the matrix (the layout-agnostic path below reads it twice: a log-sum-exp
// pass and a column-sum pass).  A 16-sample tile of 512 rows is 64 KB -- too much for one wave's registers and LDS share -- so
// the EIGHT waves of a workgroup split the rows of one tile (16 NBW rows each, their own LDS-DMA, their own double buffer) and
// meet twice per tile through two small LDS vectors: the per-sample maxima (so that every wave uses the same shift and there
// is ONE exponential per element) and the per-sample sums.  Rows past the allocated pitch are never requested (their LDS rows
// stay zero and their a_k is -inf).  Partial records: one per workgroup, `rows` entries + one objective term.
// ---------------------------------------------------------------------------------------------
template <int NBW, int NF>
__global__ void __launch_bounds__(512, 1)
k_lse_split(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, int64_t rows,
            const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden,
            double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
            double* __restrict__ obj_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = NBW * 16;                  // rows per wave
    constexpr int NP = RW / 8;                    // LDS-DMA pieces per wave and tile
    constexpr int U_BYTES = RW * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the tile's 16 sample weights
    constexpr int NW = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    double* xmax = reinterpret_cast<double*>(smem + EXP_TABLE_BYTES);  // [NW][TS]
    double* xsum = xmax + NW * TS;                                     // [NF][NW][TS]
    char* buf = smem + EXP_TABLE_BYTES + (1 + NF) * NW * TS * 8 + wave * (2 * TILE_BYTES);
    const int64_t r0 = (int64_t)wave * RW;
    const StageOffsets so = make_stage_offsets(ld, lane);
    // rows this wave never requests: zero once, in both buffers
    for (int j = 0; j < NP; ++j)
        if (r0 + 8 * j >= rows) {
            for (int bsel = 0; bsel < 2; ++bsel) *reinterpret_cast<double2*>(buf + bsel * TILE_BYTES + j * 1024 + lane * 16) = double2{0.0, 0.0};
        }
    __syncthreads();
    // second candidate (NF == 2): aden[rows + k] holds the ratio c_k = exp(a'_k - a_k); its exponentials are the first one's
    // times c_k (one exponential per element for both), its per-state sums are accumulated without c_k (applied by the caller)
    double a[NBW], c[NBW], acc[NF][NBW], objl[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) objl[f] = 0.0;
#pragma unroll
    for (int I = 0; I < NBW; ++I) {
        const int64_t r = r0 + 16 * I + ks;
        a[I] = r < rows ? aden[r] : -INFINITY;
        c[I] = (NF == 2 && r < rows) ? aden[rows + r] : 0.0;
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f][I] = 0.0;
    }
#pragma unroll
    for (int I = 0; I < NBW; ++I) {
        settle(a[I]);
        if (NF == 2) settle(c[I]);
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage = [&](int64_t tile, char* dst) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (r0 + 8 * j < rows) stage_piece<true>(u + (r0 + 8 * j) * ld + tile * TS, so.off[j & 1], dst + j * 1024, lane);
        stage_vec16<true>(cw, tile * TS, dst + U_BYTES, lane);
    };
    const int64_t G = gridDim.x;
    int64_t t = blockIdx.x;
    int cur = 0;
    if (t < ntiles) stage(t, buf);
    for (; t < ntiles; t += G) {
        char* cbuf = buf + cur * TILE_BYTES;
        wait_vm<0>();  // this tile (requested a tile period ago) and the previous tile's logden stores
        if (t + G < ntiles) stage(t + G, buf + (cur ^ 1) * TILE_BYTES);
        double x[GROUPS][NBW], w[GROUPS], mloc[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            w[g] = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
#pragma unroll
            for (int I = 0; I < NBW; ++I) x[g][I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g]);
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int I = 0; I < NBW; ++I) x[g][I] = a[I] - x[g][I];
            mloc[g] = row16_max(tree_max<NBW>(x[g]));
            if (ks == 0) xmax[wave * TS + 4 * g + ns] = mloc[g];
        }
        __syncthreads();
        double m2[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            double m = xmax[4 * g + ns];
#pragma unroll
            for (int wv = 1; wv < NW; ++wv) m = fmax(m, xmax[wv * TS + 4 * g + ns]);
            m2[g] = m * LOG2E_S;
#pragma unroll
            for (int I = 0; I < NBW; ++I) x[g][I] = fma(x[g][I], LOG2E_S, -m2[g]);
            exp2s_batch<NBW>(x[g]);
            double s0 = tree_sum<NBW>(x[g]), s1 = NF == 2 ? dot_sum<NBW>(x[g], c) : 0.0;
            if constexpr (NF == 2) row16_sum2(s0, s1); else s0 = row16_sum(s0);
            if (ks == 0) {
                xsum[wave * TS + 4 * g + ns] = s0;
                if constexpr (NF == 2) xsum[NW * TS + wave * TS + 4 * g + ns] = s1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            double ssum[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                ssum[f] = xsum[f * NW * TS + 4 * g + ns];
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) ssum[f] += xsum[f * NW * TS + wv * TS + 4 * g + ns];  // (fixed order: every wave gets the same bits)
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const double r = w[g] * recip_fast(ssum[f]);
#pragma unroll
                for (int I = 0; I < NBW; ++I) acc[f][I] = fma(x[g][I], r, acc[f][I]);
            }
            if (wave == 0 && ks < NF) {  // logden_n = shift + log(sum), one lane per sample and candidate
                const int64_t n = t * TS + 4 * g + ns;
                if (n < N) {
                    const double sv = ks == 0 ? ssum[0] : ssum[NF - 1];
                    const double ldv = fma(m2[g], LN2_OVER_S, log_pos(sv));
                    double* out = ks == 0 ? logden : logden1;
                    if (out) out[n] = ldv;
                    const double term = w[g] * (dn ? (ldv - dn[n]) : ldv);
                    if (ks == 0) objl[0] += term; else objl[NF - 1] += term;
                }
            }
        }
        cur ^= 1;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int I = 0; I < NBW; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int64_t r = r0 + 16 * I + lane;
            if (lane < 16 && r < rows) psum_part[((int64_t)blockIdx.x * NF + f) * rows + r] = v;
        }
    if (wave == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double o = wave_sum(objl[f]);
            if (lane == 0) obj_part[(int64_t)blockIdx.x * NF + f] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Evaluation pass, paired-wave variant for wide panels (NB >= 6), where one 16-sample tile is 12-32 KB
// and LDS (not registers) would limit the unpaired kernel to one wave per SIMD.  Two waves share each
// tile stream (workgroup = STREAMS streams x 2 halves, one barrier per tile); half h evaluates the 4-sample
// groups {2h, 2h+1} of every tile for all NF candidates.  Partial record of half h of stream gs: 2 gs + h.
// ---------------------------------------------------------------------------------------------
template <int NB, int NF, bool DMA, int HALF, int STREAMS>
__device__ __forceinline__ void lse_pair_body(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                              const double* __restrict__ aden, const double* __restrict__ cw,
                                              double* __restrict__ logden0, double* __restrict__ logden1,
                                              const double* __restrict__ dn, double* __restrict__ psum_part,
                                              double* __restrict__ obj_part, char* smem, int lane, int stream) {
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;
    constexpr int G0 = 2 * HALF;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + EXP_TABLE_BYTES + stream * (2 * TILE_BYTES);
    const int64_t gs = (int64_t)blockIdx.x * STREAMS + stream;
    const int64_t S = (int64_t)gridDim.x * STREAMS;
    const int64_t gs0 = (int64_t)blockIdx.x * STREAMS;
    const int64_t niter = ntiles > gs0 ? (ntiles - gs0 + S - 1) / S : 0;  // block-uniform trip count
    RowIdentity rows{0};
    const StageOffsets so = make_stage_offsets(ld, lane);

    double a[NB], c[NB], acc[NF][NB], objl = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        c[I] = NF == 2 ? aden[ROWS + 16 * I + ks] : 1.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        settle(a[I]);
        if (NF == 2) settle(c[I]);
    }
    rows.live = live_piece_mask<NB>(a, -INFINITY);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = 0.0;
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage_mine = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, DMA, HALF, 2>(u, ld, tile * TS, dst, lane, so, rows);
        if constexpr (HALF == 0) stage_vec16<DMA>(cw, tile * TS, dst + U_BYTES, lane);
    };

    int64_t t = gs;
    int cur = 0;
    if (t < ntiles) stage_mine(t, buf);
    for (int64_t it = 0; it < niter; ++it, t += S) {
        wait_vm<0>();
        __syncthreads();
        const bool active = t < ntiles;
        char* cbuf = buf + cur * TILE_BYTES;
        if (active && t + S < ntiles) stage_mine(t + S, buf + (cur ^ 1) * TILE_BYTES);
        if (active) {
            double mm = 0.0, ss[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) ss[f] = 1.0;
            lse_group_pair<NB, NF>(cbuf, cbuf + U_BYTES, rd_base, pos, G0, a, c, acc, ks, ns, mm, ss);
            const int64_t n = t * TS + 4 * (ks & 3) + ns;
            const bool mine = ((ks & 3) >= G0) && ((ks & 3) < G0 + 2) && (n < N);
            const double wn = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * (ks & 3) + ns) * 8);
            logden_out<NF>(mm, ss, ks, mine, n, wn, logden0, logden1, dn, objl);
        }
        cur ^= 1;
    }
    const int64_t rec = 2 * gs + HALF;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(rec * NF + f) * ROWS + 16 * I + lane] = v;
        }
    }
    objective_out<NF>(objl, ks, lane, obj_part, rec);
}

template <int NB, int NF, bool DMA>
__global__ void __launch_bounds__(NB <= 8 ? 512 : 256, 2)
k_lse_pair(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
           const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
           double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
           double* __restrict__ obj_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STREAMS = NB <= 8 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = wave % STREAMS;
    exp_table_init(smem);
    __syncthreads();
    if (wave / STREAMS == 0)
        lse_pair_body<NB, NF, DMA, 0, STREAMS>(u, ld, N, ntiles, aden, cw, logden0, logden1, dn, psum_part, obj_part,
                                               smem, lane, stream);
    else
        lse_pair_body<NB, NF, DMA, 1, STREAMS>(u, ld, N, ntiles, aden, cw, logden0, logden1, dn, psum_part, obj_part,
                                               smem, lane, stream);
}

// ---------------------------------------------------------------------------------------------
// Gram pass with known logden:  p = exp(anum_k - u_kn - logden_n)  (no cross-state dependency),
// acc[I][J] += p[I]^T p[J] on the fp64 matrix cores.  DIAG: one panel against itself, upper
// triangular blocks only; otherwise an NBI x NBJ rectangle between two panels.
// Output block b, register r, lane l  ->  element (row = (l >> 4) + 4 r, col = l & 15) of block b.
// ---------------------------------------------------------------------------------------------
constexpr int GRAM_AGPR_BLOCKS = 31;
// Finite stand-ins for the infinities of the operand exponent when the sweep runs without the exponential's clamp
constexpr double GRAM_NEG_HUGE = -2.9e303, LOGDEN_HUGE = 1e300;
// PMODE: `u` is the resident probability matrix and `logden` the vector of reciprocals 1 / s_n (P mode, see k_psweep):
// the MFMA operand is P_kn / s_n, ONE multiply per element, no exponential, no table.
template <int NBI, int NBJ, bool DIAG, bool DMA, bool WIDE, bool CLAMP = true, bool PMODE = false>
__global__ void __launch_bounds__(256, 1)
k_gram(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
       const double* __restrict__ anum_i, const double* __restrict__ anum_j,
       const double* __restrict__ logden, int64_t row_i0, int64_t row_j0,
       double* __restrict__ gram_part, double* __restrict__ psum_part, const int* __restrict__ ctl,
       int64_t slot_stride, int cond_needgram) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {  // device-resident solver loop: stop flag + the logden slot of the current f
        if (ctl[CTL_DONE] != 0) return;
        if (cond_needgram && ctl[CTL_NEEDGRAM] == 0) return;  // the fused sweep already produced this Gram matrix
        logden += (int64_t)ctl[CTL_SLOT] * slot_stride;
    }
    constexpr int NBT = DIAG ? NBI : NBI + NBJ;  // blocks of 16 states staged per tile
    constexpr int ROWS = NBT * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the tile's 16 logden values
    constexpr int NDMA = ROWS / 8 + 1;
    constexpr int NBLK = DIAG ? NBI * (NBI + 1) / 2 : NBI * NBJ;
    // More than 31 blocks (36 for the full diagonal panel, 32 for the 64 x 128 rectangle) do not fit the 256 AGPRs next
    // to anything else, and hipcc then rotates every accumulator through v_accvgpr copies.  The register class is
    // pinned per block instead: the first GRAM_AGPR_BLOCKS live in AGPRs, the rest in VGPRs (one wave per SIMD owns
    // the whole register file).
    constexpr bool PINNED = NBLK > GRAM_AGPR_BLOCKS;
    // Tiles of more than 128 rows (the rectangle stages 192) get ONE buffer per wave: every LDS operand of a tile is
    // in registers right after the loop top, so the buffer is refilled there and the DMA still has the whole tile
    // period to land.
    // (also for narrow diagonal panels, NB <= 5: half the LDS lets as many workgroups share a CU as the registers allow)
    constexpr int NBUF = (ROWS > 128 || (DIAG && NBI <= 5)) ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    if constexpr (!PMODE) {
        exp_table_init(smem);
        __syncthreads();
    }
    char* buf = smem + EXP_TABLE_BYTES + wave * (NBUF * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowTwoPanels rows{row_i0, row_j0, DIAG ? ROWS : NBI * 16};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);

    double a[NBT];
#pragma unroll
    for (int I = 0; I < NBT; ++I) a[I] = PMODE ? 0.0 : ((DIAG || I < NBI) ? anum_i[16 * I + ks] : anum_j[16 * (I - NBI) + ks]);
    if constexpr (!PMODE) {
#pragma unroll
        for (int I = 0; I < NBT; ++I) settle(a[I]);
        // (panels of one or two blocks only: there the sweep is bound by HBM and up to half of the rows can be padding; from
        // three blocks on the ~2 scalar instructions per piece and tile cost the matrix pipe more than the bytes save --
        // profiles/r3_ab_padding_rows_from_l2.txt)
        if constexpr (NBT <= 2) rows.live = live_piece_mask<NBT>(a, -INFINITY);
    }
    double aS[NBT];  // exponent arguments are formed directly in table units: t = (a - logden - u) S log2(e)
#pragma unroll
    for (int I = 0; I < NBT; ++I) aS[I] = CLAMP ? a[I] * LOG2E_S : fmax(a[I] * LOG2E_S, GRAM_NEG_HUGE);  // (-inf: unsampled / padded states)
    v4d acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};

    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    int64_t t = gw;
    int cur = 0;
    if constexpr (DMA) {
        if (t < ntiles) {
            stage_tile<ROWS, true, 0, 1>(u, ld, t * TS, buf, lane, so, rows);
            stage_vec16<true>(logden, t * TS, buf + U_BYTES, lane);
        }
    }
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        if constexpr (DMA && NBUF == 1) {
            wait_vm<0>();
        } else if constexpr (DMA) {
            const int64_t tn = t + W;
            if (tn < ntiles) {
                char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
                stage_tile<ROWS, true, 0, 1>(u, ld, tn * TS, nbuf, lane, so, rows);
                stage_vec16<true>(logden, tn * TS, nbuf + U_BYTES, lane);
                wait_vm<NDMA>();
            } else {
                wait_vm<0>();
            }
        } else {
            stage_tile<ROWS, false, 0, 1>(u, ld, t * TS, cbuf, lane, so, rows);
            stage_vec16<false>(logden, t * TS, cbuf + U_BYTES, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // every LDS operand of the tile is requested up front (one exposed LDS round trip per tile)
        double ldc[GROUPS], uv[GROUPS][NBT];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            ldc[g] = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
#pragma unroll
            for (int I = 0; I < NBT; ++I)
                uv[g][I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd_base + pos[g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DMA && NBUF == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the whole tile is in registers: refill its buffer
            if (t + W < ntiles) {
                stage_tile<ROWS, true, 0, 1>(u, ld, (t + W) * TS, cbuf, lane, so, rows);
                stage_vec16<true>(logden, (t + W) * TS, cbuf + U_BYTES, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const bool valid = (t * TS + 4 * g + ns) < N;
            double p[NBT];
            if constexpr (PMODE) {
                const double rin = valid ? ldc[g] : 0.0;  // 1 / s_n (times sqrt(c_n) when weighted); padded samples: 0
#pragma unroll
                for (int I = 0; I < NBT; ++I) p[I] = uv[g][I] * rin;
            } else {
                // padded samples (and samples of multiplicity zero, logden = +inf): exp(-inf) = 0
                const double lde = CLAMP ? (valid ? ldc[g] : INFINITY) : (valid ? fmin(ldc[g], LOGDEN_HUGE) : LOGDEN_HUGE);
#pragma unroll
                for (int I = 0; I < NBT; ++I) p[I] = fma(uv[g][I], -LOG2E_S, aS[I] - lde * LOG2E_S);
                exp2s_batch<NBT, CLAMP>(p);
            }
            auto mfma = [&](int b, double x, double y) {
                if constexpr (PINNED) {
                    if (b < GRAM_AGPR_BLOCKS)
                        asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(x), "v"(y));
                    else
                        asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(x), "v"(y));
                } else {
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[b], 0, 0, 0);
                }
            };
            // The asm MFMAs are opaque to the scheduler and to the hazard recogniser.  Left free, hipcc interleaves the next
            // group's VALU work between them and the 64 x 128 rectangle then produced wrong blocks (a matrix-core
            // hazard the compiler could not see); fenced, the block is issued as written, behind one conservative s_nop.
            if constexpr (PINNED) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            if constexpr (DIAG) {
                int b = 0;
#pragma unroll
                for (int I = 0; I < NBI; ++I)
#pragma unroll
                    for (int J = I; J < NBI; ++J) {
#ifdef MBAR_EXPERIMENT_NODIAG  // timing experiment only (profiles/r2_gram_ceiling.txt): the diagonal blocks are never computed
                        if (I == J) { ++b; continue; }
#endif
                        mfma(b++, p[I], p[J]);
                    }
            } else {
#pragma unroll
                for (int I = 0; I < NBI; ++I)
#pragma unroll
                    for (int J = 0; J < NBJ; ++J) mfma(I * NBJ + J, p[I], p[NBI + J]);
            }
            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= NBUF - 1;
    }
    if constexpr (PINNED) {
        // the asm MFMAs are opaque to the hazard recogniser: cover the matrix-result -> VALU read distance by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) gram_part[((gw * NBLK + b) * 4 + r) * 64 + lane] = acc[b][r];
}

// ---------------------------------------------------------------------------------------------
// Gram pass for 129 .. 256 states in ONE read of the matrix.  A panel of NBT = 12 / 16 blocks of 16 states has 78 / 136
// upper-triangular 16 x 16 blocks -- four times what one wave's register file holds (k_gram above covers such panels with
// four launches: two diagonal 128-state panels and two 64 x 128 rectangles, 2.5 reads of the matrix and every exponential
// computed 2.2 times).  Here the FOUR waves of a workgroup (one per SIMD, one workgroup per CU) share one tile stream:
//   [wait own LDS-DMA: a quarter of the tile's rows + an own copy of its 16 logden values]
//   [each wave turns its quarter of the rows into operands IN PLACE: p = exp(a - u - logden), or P / s in P mode]
//   [ONE barrier] [the next tile is requested into the other buffer behind the first matrix instructions]
//   [each wave reads the operands of all NBT row blocks, group by group, and issues ITS blocks: every fourth
//   block of the row-major upper triangle -- 34 of 136 (20 / 19 of 78), in the pinned AGPR / VGPR classes of k_gram]
// so the matrix is read once, every exponential is computed once (16 per lane and tile instead of 52), and the matrix pipe
// runs 34 x 4 x 64 = 8704 cycles per tile and SIMD against ~1500 (P mode: ~150) cycles of operand work and one barrier.
// Partial records: ONE per workgroup (NBLK blocks of 256 doubles, block b = (I, J), I <= J, row-major: the layout of the
// single-panel kernel, so the reduction, the K x K solve and the host-side unpacking are the ones of K <= 128).
// ---------------------------------------------------------------------------------------------
constexpr int quad_blocks_of(int nbt, int w) { return (nbt * (nbt + 1) / 2 - w + 3) / 4; }
// A wave's blocks into the workgroup's record (the NBT panel's layout: block (I, J), I <= J, row-major): live blocks -- every
// fourth of the NBM triangle -- from the accumulators, the blocks of the padding rows (every fourth of those) as zeros.
template <int NBT, int NBM, int WV, typename Acc>
__device__ __forceinline__ void quad_store_blocks(double* __restrict__ rec, const Acc& acc, int lane) {
    int b = 0, live = 0, dead = 0, mine = 0;
#pragma unroll
    for (int I = 0; I < NBT; ++I)
#pragma unroll
        for (int J = I; J < NBT; ++J) {
            if (I < NBM && J < NBM) {
                if ((live & 3) == WV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rec[(b * 4 + r) * 64 + lane] = acc[mine][r];
                    ++mine;
                }
                ++live;
            } else {
                if ((dead & 3) == WV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rec[(b * 4 + r) * 64 + lane] = 0.0;
                }
                ++dead;
            }
            ++b;
        }
}
// STOREP (classic operands only): the operand tile IS the normalised probability matrix exp(a - u - logden) when `logden` are the
// log-denominators at `a` -- each wave also writes its quarter of it out (coalesced 16-byte stores that mirror the LDS-DMA
// pattern, behind the blocks of group 1): the build of the resident probability matrix for 129 .. 256 states rides on the Gram
// sweep at the anchor.
// NBM <= NBT: blocks of 16 states that hold real states (a 160-state problem in the 192-row panel: 10 of 12).  The rows
// beyond are padding: they are neither staged nor turned into operands, and their blocks are left out (55 matrix instructions
// per k-step instead of 78) -- the record keeps the panel's layout, with zeros there.
template <int NBT, int WV, bool WIDE, bool PMODE, bool STOREP = false, int NBM = NBT>
__device__ __forceinline__ void gram_quad_body(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                               const double* __restrict__ anum, const double* __restrict__ logden,
                                               double* __restrict__ gram_part, char* smem, int lane, double* __restrict__ Pout = nullptr) {
    constexpr int ROWS = NBT * 16, NQ = NBT / 4, QDMA = ROWS / 4 / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 4 * 1024;  // + one copy of the tile's 16 logden values per wave (a 1 KB LDS-DMA piece each)
    constexpr int NBLK = NBT * (NBT + 1) / 2, NMINE = quad_blocks_of(NBM, WV);
    static_assert(NBM <= NBT && quad_blocks_of(NBM, 3) >= QDMA + 2, "every wave needs QDMA + 2 blocks to hang its LDS-DMA behind");
    // (hand-placed asm matrix instructions also for the 192-state panel, whose 19 / 20 blocks per wave the compiler could manage:
    // left to it, K = 192 ran at 0.565 of the matrix peak against 0.600 this way)
    constexpr bool PINNED = true;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + EXP_TABLE_BYTES;  // two tile buffers shared by the four waves
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);
    const int64_t G = gridDim.x;

    double aS[NQ];  // exponent constants of the rows this wave turns into operands, in table units
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double a = (PMODE || WV * NQ + i >= NBM) ? 0.0 : anum[16 * (WV * NQ + i) + ks];
        if constexpr (!PMODE) settle(a);
        aS[i] = a * LOG2E_S;
    }
    v4d acc[NMINE];
#pragma unroll
    for (int b = 0; b < NMINE; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    // this wave's quarter of a tile's rows (QDMA LDS-DMA instructions of 8 rows) and its OWN copy of the tile's 16 logden values
    // (behind the tile, one 128-byte slot per wave): everything a wave needs to turn its rows into operands it has staged
    // itself, so that step needs no barrier
    auto stage_piece_j = [&](int64_t tile, char* dst, int j) {
        if (j < 2 * NBM) stage_piece<true>(u + rows(8 * j) * ld + tile * TS, so.off[j & 1], dst + j * 1024, lane);  // (j is a constant)
    };
    // (a full-wave LDS-DMA of 1 KB: lanes 0 .. 7 bring the 16 values, the others repeat them -- no exec-masked branch among the
    // matrix instructions: the register allocator handles the pinned accumulators only in straight-line code)
    const char* lsrc = reinterpret_cast<const char*>(logden) + (lane & 7) * 16;
    auto stage_l = [&](int64_t tile, char* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + tile * (TS * 8)),
                                         (__attribute__((address_space(3))) void*)(dst + U_BYTES + WV * 1024), 16, 0, 0);
    };
    auto stage = [&](int64_t tile, char* dst) {
#pragma unroll
        for (int j = WV * QDMA; j < (WV + 1) * QDMA; ++j) stage_piece_j(tile, dst, j);
        stage_l(tile, dst);
    };
    auto read_group = [&](const char* tb, int g, double (&x)[NBT]) {
#pragma unroll
        for (int I = 0; I < NBM; ++I) x[I] = *reinterpret_cast<const double*>(tb + I * (16 * TS * 8) + rd_base + pos[g]);
    };
    auto mfma = [&](int b, double x, double y) {
        if constexpr (PINNED) {
            if (b < GRAM_AGPR_BLOCKS)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(x), "v"(y));
            else
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(x), "v"(y));
        } else {
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[b], 0, 0, 0);
        }
    };

    // Per tile: [own LDS-DMA landed] [own rows -> operands, in place] [ONE barrier] [request the next tile into the other
    // buffer, behind the first matrix instructions] [own blocks].  The barrier of tile t says "every wave has finished the
    // blocks of tile t - 1", which is what frees the other buffer; the request then has the whole block phase to land.
    // this wave's own rows of a tile (and its logden values) out of LDS: requested for tile t + 1 behind the last blocks of tile t,
    // so that the operand step at the loop top starts on registers
    double x[GROUPS * NQ], ldc[GROUPS];
    auto read_own = [&](const char* tb) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            ldc[g] = *reinterpret_cast<const double*>(tb + U_BYTES + WV * 1024 + (4 * g + ns) * 8);
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                x[g * NQ + i] = WV * NQ + i < NBM ? *reinterpret_cast<const double*>(tb + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]) : 0.0;
        }
    };
    int64_t t = blockIdx.x;
    int cur = 0;
    constexpr bool PREFETCH = PINNED;
    if (t < ntiles) {
        stage(t, buf);
        if constexpr (PREFETCH) {
            wait_vm<0>();
            read_own(buf);
        }
    }
    for (; t < ntiles; t += G) {
        char* cbuf = buf + cur * TILE_BYTES;
        char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
        const int64_t tnext = t + G < ntiles ? t + G : t;  // (past the end this tile is requested again and never looked at)
        if constexpr (!PREFETCH) {
            wait_vm<0>();
            read_own(cbuf);
        }
        // ---- operands of this wave's rows, in place
        {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const bool valid = (t * TS + 4 * g + ns) < N;
                if constexpr (PMODE) {
                    const double rin = valid ? ldc[g] : 0.0;  // 1 / s_n (times sqrt(c_n) when weighted); padded samples: 0
#pragma unroll
                    for (int i = 0; i < NQ; ++i) x[g * NQ + i] *= rin;
                } else {
                    const double lde = (valid ? ldc[g] : INFINITY) * LOG2E_S;  // padded samples / multiplicity zero: exp(-inf) = 0
#pragma unroll
                    for (int i = 0; i < NQ; ++i) x[g * NQ + i] = fma(x[g * NQ + i], -LOG2E_S, aS[i] - lde);
                }
            }
            if constexpr (!PMODE) exp2s_batch<GROUPS * NQ>(x);
            if constexpr (STOREP) {  // (what is kept as P: entries below the normal range are flushed to zero)
#pragma unroll
                for (int e = 0; e < GROUPS * NQ; ++e) x[e] = x[e] >= 2.3e-308 ? x[e] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < GROUPS; ++g)
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (WV * NQ + i < NBM)
                        *reinterpret_cast<double*>(cbuf + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]) = x[g * NQ + i];
        }
        __syncthreads();  // every row of tile t holds operands; every wave is done with the other buffer
        // ---- this wave's blocks, group by group; the operands of the next group are requested behind the first block, the
        // LDS-DMA pieces of the next tile behind the blocks that follow in group 0
        double p[2][NBT];
        read_group(cbuf, 0, p[0]);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if constexpr (PINNED) {  // (asm matrix instructions are opaque to the scheduler and the hazard recogniser: see k_gram)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            int b = 0, mine = 0;  // (b counts the blocks of the LIVE triangle here)
#pragma unroll
            for (int I = 0; I < NBM; ++I)
#pragma unroll
                for (int J = I; J < NBM; ++J) {
                    if ((b & 3) == WV) {
                        mfma(mine, p[g & 1][I], p[g & 1][J]);
                        if (mine == 0 && g < GROUPS - 1) {
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                            read_group(cbuf, g + 1, p[(g + 1) & 1]);
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                        }
                        if (g == 0 && mine >= 1 && mine <= QDMA + 1) {  // one piece behind each of the next blocks (the last: logden)
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                            if (mine <= QDMA)
                                stage_piece_j(tnext, nbuf, WV * QDMA + mine - 1);
                            else
                                stage_l(tnext, nbuf);
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                        }
                        if (STOREP && g == 1 && mine >= 1 && mine <= QDMA && WV * QDMA + mine - 1 < 2 * NBM) {  // this wave's quarter of the operand tile out as P
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                            const int j = WV * QDMA + mine - 1;
                            const double2 pv = *reinterpret_cast<const double2*>(cbuf + j * 1024 + lane * 16);
                            *reinterpret_cast<double2*>(reinterpret_cast<char*>(Pout + rows(8 * j) * ld + t * TS) + so.off[j & 1]) = pv;
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                        }
                        if (PREFETCH && g == GROUPS - 1 && mine == 1) {  // the next tile was requested three groups ago: its rows into registers
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                            wait_vm<0>();
                            read_own(nbuf);
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                        }
                        ++mine;
                    }
                    ++b;
                }
            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    }
    if constexpr (PINNED) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // matrix result -> VALU read distance
    quad_store_blocks<NBT, NBM, WV>(gram_part + (int64_t)blockIdx.x * NBLK * 256, acc, lane);
}

template <int NBT, bool WIDE, bool PMODE, bool STOREP = false, int NBM = NBT>
__global__ void __launch_bounds__(256, 1)
k_gram_quad(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ anum,
            const double* __restrict__ logden, double* __restrict__ gram_part, const int* __restrict__ ctl,
            int64_t slot_stride, int cond_needgram, double* __restrict__ Pout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {  // device-resident solver loop: stop flag + the logden / reciprocal slot of the current f
        if (ctl[CTL_DONE] != 0) return;
        if (cond_needgram && ctl[CTL_NEEDGRAM] == 0) return;
        logden += (int64_t)ctl[CTL_SLOT] * slot_stride;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (!PMODE) {
        exp_table_init(smem);
        __syncthreads();
    }
    switch (wave) {
        case 0: gram_quad_body<NBT, 0, WIDE, PMODE, STOREP, NBM>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, Pout); break;
        case 1: gram_quad_body<NBT, 1, WIDE, PMODE, STOREP, NBM>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, Pout); break;
        case 2: gram_quad_body<NBT, 2, WIDE, PMODE, STOREP, NBM>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, Pout); break;
        default: gram_quad_body<NBT, 3, WIDE, PMODE, STOREP, NBM>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, Pout); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused sweep for 129 .. 256 states (P mode): k_gram_quad's tile stream -- four waves, one shared tile, each wave its quarter of
// the rows and every fourth block -- carrying what k_fused carries for one panel: the normalisers 1 / s_n of both candidates
// (each wave's partial dot products over ITS rows meet in a 1 KB LDS table: one more barrier per tile), the per-state sums of
// both (each wave for its rows: the four waves' records are disjoint, nothing to fold), the reciprocals into the slot vectors
// (wave w stores group w), and the Gram matrix of the second multiplier row on the matrix cores.
//   [own rows in registers (requested behind the last blocks of the previous tile)] [partial normalisers -> LDS] [barrier]
//   [normalisers, reciprocals, per-state sums; operands P / s written in place] [barrier] [own blocks; next tile's LDS-DMA
//   behind the first of them]
// ---------------------------------------------------------------------------------------------
template <int NBT, int WV, bool WIDE, int NBM = NBT>
__device__ __forceinline__ void fused_quad_body(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles,
                                                const double* __restrict__ cmul, const double* __restrict__ cw,
                                                const double* __restrict__ wsq, double* __restrict__ rinv0,
                                                double* __restrict__ rinv1, double* __restrict__ gram_part,
                                                double* __restrict__ psum_part, char* smem, int lane) {
    constexpr int ROWS = NBT * 16, NQ = NBT / 4, QDMA = ROWS / 4 / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 4 * 1024;  // + per wave: the tile's 16 multiplicities and their 16 roots (a 1 KB LDS-DMA piece)
    constexpr int NBLK = NBT * (NBT + 1) / 2, NMINE = quad_blocks_of(NBM, WV);
    static_assert(NBM <= NBT && quad_blocks_of(NBM, 3) >= QDMA + 2, "every wave needs QDMA + 2 blocks to hang its LDS-DMA behind");
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem;  // two tile buffers shared by the four waves; behind them the table of partial normalisers
    double* xs = reinterpret_cast<double*>(smem + 2 * TILE_BYTES);  // [wave][candidate][16 samples]
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);
    const int64_t G = gridDim.x;

    double c0[NQ], c1[NQ], acc0[NQ], acc1[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        c0[i] = WV * NQ + i < NBM ? cmul[16 * (WV * NQ + i) + ks] : 0.0;
        c1[i] = WV * NQ + i < NBM ? cmul[ROWS + 16 * (WV * NQ + i) + ks] : 0.0;
        acc0[i] = acc1[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        settle(c0[i]);
        settle(c1[i]);
    }
    v4d acc[NMINE];
#pragma unroll
    for (int b = 0; b < NMINE; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage_piece_j = [&](int64_t tile, char* dst, int j) {
        if (j < 2 * NBM) stage_piece<true>(P + rows(8 * j) * ld + tile * TS, so.off[j & 1], dst + j * 1024, lane);  // (j is a constant)
    };
    // multiplicities and their roots behind the tile, one full-wave piece per wave: even 128-byte rows of it take cw, odd rows wsq
    const char* wsrc = reinterpret_cast<const char*>(((lane >> 3) & 1) ? wsq : cw) + (lane & 7) * 16;
    auto stage_w = [&](int64_t tile, char* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + tile * (TS * 8)),
                                         (__attribute__((address_space(3))) void*)(dst + U_BYTES + WV * 1024), 16, 0, 0);
    };
    auto stage = [&](int64_t tile, char* dst) {
#pragma unroll
        for (int j = WV * QDMA; j < (WV + 1) * QDMA; ++j) stage_piece_j(tile, dst, j);
        stage_w(tile, dst);
    };
    auto read_group = [&](const char* tb, int g, double (&x)[NBT]) {
#pragma unroll
        for (int I = 0; I < NBM; ++I) x[I] = *reinterpret_cast<const double*>(tb + I * (16 * TS * 8) + rd_base + pos[g]);
    };
    auto mfma = [&](int b, double x, double y) {
        if (b < GRAM_AGPR_BLOCKS)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(x), "v"(y));
        else
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(x), "v"(y));
    };
    double x[GROUPS * NQ], w[GROUPS], sw[GROUPS];
    auto read_own = [&](const char* tb) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            w[g] = *reinterpret_cast<const double*>(tb + U_BYTES + WV * 1024 + (4 * g + ns) * 8);
            sw[g] = *reinterpret_cast<const double*>(tb + U_BYTES + WV * 1024 + TS * 8 + (4 * g + ns) * 8);
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                x[g * NQ + i] = WV * NQ + i < NBM ? *reinterpret_cast<const double*>(tb + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]) : 0.0;
        }
    };
    int64_t t = blockIdx.x;
    int cur = 0;
    if (t < ntiles) {
        stage(t, buf);
        wait_vm<0>();
        read_own(buf);
    }
    for (; t < ntiles; t += G) {
        char* cbuf = buf + cur * TILE_BYTES;
        char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
        const int64_t tnext = t + G < ntiles ? t + G : t;  // (past the end this tile is requested again and never looked at)
        // ---- this wave's share of the normalisers s_n = sum_k P_kn c_k of both candidates
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                d0 = fma(x[g * NQ + i], c0[i], d0);
                d1 = fma(x[g * NQ + i], c1[i], d1);
            }
            row16_sum2(d0, d1);
            if (ks < 2) xs[(WV * 2 + ks) * TS + 4 * g + ns] = ks == 0 ? d0 : d1;
        }
        __syncthreads();  // (the four partial sums of every sample are in the table; every wave is done with the other buffer)
        // ---- reciprocals, per-state sums, operands in place
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int sidx = 4 * g + ns;
            const double s0 = (xs[(0 * 2 + 0) * TS + sidx] + xs[(1 * 2 + 0) * TS + sidx]) + (xs[(2 * 2 + 0) * TS + sidx] + xs[(3 * 2 + 0) * TS + sidx]);
            const double s1 = (xs[(0 * 2 + 1) * TS + sidx] + xs[(1 * 2 + 1) * TS + sidx]) + (xs[(2 * 2 + 1) * TS + sidx] + xs[(3 * 2 + 1) * TS + sidx]);
            // (a padded sample has an all-zero column: keep its reciprocal finite, its multiplicity and the root of it are 0)
            const double r0 = recip_fast(fmax(s0, 1e-300)), r1 = recip_fast(fmax(s1, 1e-300));
            const double q0 = w[g] * r0, q1 = w[g] * r1, rin = r1 * sw[g];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (WV * NQ + i >= NBM) continue;  // (padding rows: never staged, never read)
                acc0[i] = fma(x[g * NQ + i], q0, acc0[i]);
                acc1[i] = fma(x[g * NQ + i], q1, acc1[i]);
                *reinterpret_cast<double*>(cbuf + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]) = x[g * NQ + i] * rin;
            }
            if (g == WV) {  // this wave stores the reciprocals of group WV (one store instruction per wave and tile)
                const int64_t n = t * TS + sidx;
                if (n < N && ks < 2) (ks == 0 ? rinv0 : rinv1)[n] = ks == 0 ? r0 : r1;
            }
        }
        __syncthreads();  // every row of tile t holds operands
        // ---- this wave's blocks (see k_gram_quad)
        double p[2][NBT];
        read_group(cbuf, 0, p[0]);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7");
            int b = 0, mine = 0;  // (b counts the blocks of the LIVE triangle here)
#pragma unroll
            for (int I = 0; I < NBM; ++I)
#pragma unroll
                for (int J = I; J < NBM; ++J) {
                    if ((b & 3) == WV) {
                        mfma(mine, p[g & 1][I], p[g & 1][J]);
                        if (mine == 0 && g < GROUPS - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            read_group(cbuf, g + 1, p[(g + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (g == 0 && mine >= 1 && mine <= QDMA + 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (mine <= QDMA)
                                stage_piece_j(tnext, nbuf, WV * QDMA + mine - 1);
                            else
                                stage_w(tnext, nbuf);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (g == GROUPS - 1 && mine == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            wait_vm<0>();
                            read_own(nbuf);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ++mine;
                    }
                    ++b;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // matrix result -> VALU read distance
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double v0 = acc0[i], v1 = acc1[i];
        v0 += __shfl_xor(v0, 16);
        v0 += __shfl_xor(v0, 32);
        v1 += __shfl_xor(v1, 16);
        v1 += __shfl_xor(v1, 32);
        if (lane < 16) {
            psum_part[((int64_t)blockIdx.x * 2 + 0) * ROWS + 16 * (WV * NQ + i) + lane] = v0;
            psum_part[((int64_t)blockIdx.x * 2 + 1) * ROWS + 16 * (WV * NQ + i) + lane] = v1;
        }
    }
    quad_store_blocks<NBT, NBM, WV>(gram_part + (int64_t)blockIdx.x * NBLK * 256, acc, lane);
}

template <int NBT, bool WIDE, int NBM = NBT>
__global__ void __launch_bounds__(256, 1)
k_fused_quad(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ cmul,
             const double* __restrict__ cw, const double* __restrict__ wsq, double* __restrict__ rinv0,
             double* __restrict__ gram_part, double* __restrict__ psum_part, const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl[CTL_DONE] != 0) return;
    const int s = ctl[CTL_SLOT];
    double* rinv1 = rinv0 + (int64_t)((s + 2) % 3) * slot_stride;
    rinv0 = rinv0 + (int64_t)((s + 1) % 3) * slot_stride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: fused_quad_body<NBT, 0, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        case 1: fused_quad_body<NBT, 1, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        case 2: fused_quad_body<NBT, 2, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        default: fused_quad_body<NBT, 3, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
    }
}

// P = exp(aden_k - u_kn - logden_n) for the rows / samples of a shard (padding: 0; entries below the normal range are flushed to
// zero): the resident probability matrix of 129 .. 256 states, built from the log-denominators an evaluation sweep left behind.
__global__ void __launch_bounds__(256)
k_make_p(const double* __restrict__ u, int64_t ld, int64_t N, int64_t rows, const double* __restrict__ aden,
         const double* __restrict__ logden, double* __restrict__ P) {
    const int64_t per_row = ld / 2;  // two samples per thread
    const int64_t total = rows * per_row;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / per_row, n = (e - k * per_row) * 2;
        const double a = aden[k];
        const double2 uv = *reinterpret_cast<const double2*>(u + k * ld + n);
        double2 pv;
        pv.x = (n < N) ? exp(a - uv.x - logden[n]) : 0.0;
        pv.y = (n + 1 < N) ? exp(a - uv.y - logden[n + 1]) : 0.0;
        if (!(pv.x >= 2.3e-308)) pv.x = 0.0;  // (also a = -inf: unsampled / padded state)
        if (!(pv.y >= 2.3e-308)) pv.y = 0.0;
        *reinterpret_cast<double2*>(P + k * ld + n) = pv;
    }
}
__global__ void __launch_bounds__(256) k_fill(double* __restrict__ v, double value, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = value;
}
// dst = sqrt(src): the roots of the sample multiplicities for the matrix-core operands of the weighted sweeps
__global__ void __launch_bounds__(256) k_sqrt_vec(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = sqrt(src[i]);
}

// ---------------------------------------------------------------------------------------------
// Gram pass, paired-wave variant for a full 128-state panel (NB = 8: 36 upper-triangular blocks =
// 288 accumulator registers, more than the 256 AGPRs one wave can hold without the compiler
// rotating accumulators through VGPRs after every MFMA).  A workgroup has 8 waves = 4 tile streams
// x 2 halves; the two waves of a stream share the LDS tile, each computes all the operands p (the
// exp work is duplicated, the matrix pipe is the bound) and owns every other block (18 blocks, 144
// AGPRs), so two waves fit per SIMD and the hardware overlaps one wave's exp with the other's MFMA.
// One workgroup barrier per tile: [wait own DMA] [barrier] [issue DMA of the next tile into the
// buffer everybody just finished] [compute].
// ---------------------------------------------------------------------------------------------
template <int NB, int H, int NBLK>
__device__ __forceinline__ void gram_half_group(const double (&p)[NB], v4d (&acc)[NBLK]) {
    int b = 0, mine = 0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) {
            if ((b & 1) == H) {
                acc[mine] = __builtin_amdgcn_mfma_f64_16x16x4f64(p[I], p[J], acc[mine], 0, 0, 0);
                ++mine;
            }
            ++b;
        }
}

template <int NB, bool DMA, int HALF>
__device__ __forceinline__ void gram_pair_body(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                               const double* __restrict__ anum, const double* __restrict__ logden,
                                               int64_t row0, double* __restrict__ gram_part,
                                               double* __restrict__ psum_part, char* smem, int lane, int stream) {
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;
    constexpr int NBLK = NB * (NB + 1) / 2;
    constexpr int NMINE = HALF == 0 ? (NBLK + 1) / 2 : NBLK / 2;  // blocks b with (b & 1) == HALF
    constexpr int STREAMS = 4;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + EXP_TABLE_BYTES + stream * (2 * TILE_BYTES);
    const int64_t gs = (int64_t)blockIdx.x * STREAMS + stream;   // global stream id = partial record
    const int64_t S = (int64_t)gridDim.x * STREAMS;
    const int64_t gs0 = (int64_t)blockIdx.x * STREAMS;
    const int64_t niter = ntiles > gs0 ? (ntiles - gs0 + S - 1) / S : 0;  // block-uniform trip count
    RowIdentity rows{row0};
    const StageOffsets so = make_stage_offsets(ld, lane);

    double a[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) a[I] = anum[16 * I + ks];
#pragma unroll
    for (int I = 0; I < NB; ++I) settle(a[I]);
    v4d acc[NMINE];
#pragma unroll
    for (int b = 0; b < NMINE; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};

    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    // this wave stages DMA instructions j with (j & 1) == HALF; half 0 also stages the logden slot
    auto stage_mine = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, DMA, HALF, 2>(u, ld, tile * TS, dst, lane, so, rows);
        if constexpr (HALF == 0) stage_vec16<DMA>(logden, tile * TS, dst + U_BYTES, lane);
    };

    int64_t t = gs;
    int cur = 0;
    if (t < ntiles) stage_mine(t, buf);
    for (int64_t it = 0; it < niter; ++it, t += S) {
        wait_vm<0>();
        __syncthreads();
        const bool active = t < ntiles;
        char* cbuf = buf + cur * TILE_BYTES;
        if (active && t + S < ntiles) stage_mine(t + S, buf + (cur ^ 1) * TILE_BYTES);
        if (active) {
            double ldc[GROUPS];
#pragma unroll
            for (int g = 0; g < GROUPS; ++g)
                ldc[g] = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const bool valid = (t * TS + 4 * g + ns) < N;
                const double lde = valid ? ldc[g] : INFINITY;
                double p[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    const double uv = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd_base + pos[g]);
                    p[I] = exp2s_fast(((a[I] - lde) - uv) * LOG2E_S);
                }
                gram_half_group<NB, HALF, NMINE>(p, acc);
            }
        }
        cur ^= 1;
    }
    // block b of the full enumeration lives in half (b & 1) at slot b >> 1
#pragma unroll
    for (int sl = 0; sl < NMINE; ++sl) {
        const int b = 2 * sl + HALF;
#pragma unroll
        for (int r = 0; r < 4; ++r) gram_part[((gs * NBLK + b) * 4 + r) * 64 + lane] = acc[sl][r];
    }
}

template <int NB, bool DMA>
__global__ void __launch_bounds__(512, 2)
k_gram_pair(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
            const double* __restrict__ anum, const double* __restrict__ logden, int64_t row0,
            double* __restrict__ gram_part, double* __restrict__ psum_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = wave & 3;
    exp_table_init(smem);
    __syncthreads();
    if ((wave >> 2) == 0)
        gram_pair_body<NB, DMA, 0>(u, ld, N, ntiles, anum, logden, row0, gram_part, psum_part, smem, lane, stream);
    else
        gram_pair_body<NB, DMA, 1>(u, ld, N, ntiles, anum, logden, row0, gram_part, psum_part, smem, lane, stream);
}

// ---------------------------------------------------------------------------------------------
// Gram pass, paired waves with operand exchange (default for a full 128-state panel).
// Measured on gfx950: VALU work does not overlap with v_mfma_f64 even across waves of a SIMD (the fp64
// matrix op occupies the vector ALU), so every exp the two waves of a stream compute twice costs real
// time.  Here half h computes p only for groups {2h, 2h+1} (16 exp per tile instead of 32), publishes
// them in LDS in operand order ([group][I][lane], conflict-free ds_write/ds_read_b64) and reads the
// partner's; each wave still owns every other upper-triangular block.  The u tile is single-buffered:
// it is dead once both waves have read their groups (barrier 2), and the DMA of the next tile then
// overlaps the MFMA phase.
//   [wait own DMA] [barrier 1] [read u, exp, write p] [barrier 2] [DMA next tile] [read partner p, MFMA]
// Partial records: gram per stream (each half writes its own blocks), psum per wave (2 gs + h).
// ---------------------------------------------------------------------------------------------
template <int NB, bool DMA, int HALF>
__device__ __forceinline__ void gram_xchg_body(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                               const double* __restrict__ anum, const double* __restrict__ logden,
                                               int64_t row0, double* __restrict__ gram_part,
                                               double* __restrict__ psum_part, char* smem, int lane, int stream) {
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int L_BYTES = TS * 8;
    constexpr int P_BYTES = GROUPS * NB * 64 * 8;
    constexpr int STREAM_BYTES = U_BYTES + L_BYTES + P_BYTES;
    constexpr int NBLK = NB * (NB + 1) / 2;
    constexpr int NMINE = HALF == 0 ? (NBLK + 1) / 2 : NBLK / 2;
    constexpr int STREAMS = 4;
    constexpr int G0 = 2 * HALF, P0 = 2 * (1 - HALF);  // own groups G0, G0+1; partner's P0, P0+1
    const int ks = lane & 15, ns = lane >> 4;
    char* ubuf = smem + EXP_TABLE_BYTES + stream * STREAM_BYTES;
    char* lbuf = ubuf + U_BYTES;
    char* pbuf = lbuf + L_BYTES;
    const int64_t gs = (int64_t)blockIdx.x * STREAMS + stream;
    const int64_t S = (int64_t)gridDim.x * STREAMS;
    const int64_t gs0 = (int64_t)blockIdx.x * STREAMS;
    const int64_t niter = ntiles > gs0 ? (ntiles - gs0 + S - 1) / S : 0;
    RowIdentity rows{row0};
    const StageOffsets so = make_stage_offsets(ld, lane);

    double a[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) a[I] = anum[16 * I + ks];
#pragma unroll
    for (int I = 0; I < NB; ++I) settle(a[I]);
    v4d acc[NMINE];
#pragma unroll
    for (int b = 0; b < NMINE; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};

    const int rd_base = ks * (TS * 8);
    int pos[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) pos[gi] = ((4 * (G0 + gi) + ns + (ks & 14)) & 15) * 8;

    auto stage_mine = [&](int64_t tile) {
        stage_tile<ROWS, DMA, HALF, 2>(u, ld, tile * TS, ubuf, lane, so, rows);
        if constexpr (HALF == 0) stage_vec16<DMA>(logden, tile * TS, lbuf, lane);
    };

    int64_t t = gs;
    if (t < ntiles) stage_mine(t);
    for (int64_t it = 0; it < niter; ++it, t += S) {
        wait_vm<0>();
        __syncthreads();  // barrier 1: tile landed; everybody is done with the previous tile's operands
        const bool active = t < ntiles;
        double p_own[2][NB];
        if (active) {
            double ldc[2];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {  // all LDS operands of this wave's two groups requested up front
                ldc[gi] = *reinterpret_cast<const double*>(lbuf + (4 * (G0 + gi) + ns) * 8);
#pragma unroll
                for (int I = 0; I < NB; ++I)
                    p_own[gi][I] = *reinterpret_cast<const double*>(ubuf + I * (16 * TS * 8) + rd_base + pos[gi]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int g = G0 + gi;
                const bool valid = (t * TS + 4 * g + ns) < N;
                const double lde = valid ? ldc[gi] : INFINITY;
#pragma unroll
                for (int I = 0; I < NB; ++I) p_own[gi][I] = ((a[I] - lde) - p_own[gi][I]) * LOG2E_S;
                exp2s_batch<NB>(p_own[gi]);
#pragma unroll
                for (int I = 0; I < NB; ++I)
                    *reinterpret_cast<double*>(pbuf + ((g * NB + I) * 64 + lane) * 8) = p_own[gi][I];
            }
        }
        __syncthreads();  // barrier 2: operands published, u tile dead
        if (active && t + S < ntiles) stage_mine(t + S);
        if (active) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) gram_half_group<NB, HALF, NMINE>(p_own[gi], acc);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                double pp[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I)
                    pp[I] = *reinterpret_cast<const double*>(pbuf + (((P0 + gi) * NB + I) * 64 + lane) * 8);
                gram_half_group<NB, HALF, NMINE>(pp, acc);
            }
        }
    }
#pragma unroll
    for (int sl = 0; sl < NMINE; ++sl) {
        const int b = 2 * sl + HALF;
#pragma unroll
        for (int r = 0; r < 4; ++r) gram_part[((gs * NBLK + b) * 4 + r) * 64 + lane] = acc[sl][r];
    }
}

template <int NB, bool DMA>
__global__ void __launch_bounds__(512, 2)
k_gram_xchg(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
            const double* __restrict__ anum, const double* __restrict__ logden, int64_t row0,
            double* __restrict__ gram_part, double* __restrict__ psum_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = wave & 3;
    exp_table_init(smem);
    __syncthreads();
    if ((wave >> 2) == 0)
        gram_xchg_body<NB, DMA, 0>(u, ld, N, ntiles, anum, logden, row0, gram_part, psum_part, smem, lane, stream);
    else
        gram_xchg_body<NB, DMA, 1>(u, ld, N, ntiles, anum, logden, row0, gram_part, psum_part, smem, lane, stream);
}

// ---------------------------------------------------------------------------------------------
// Layout-agnostic fallbacks (any K): lanes along n, one sample per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lse_generic(const double* __restrict__ u, int64_t ld, int64_t N, int64_t K,
              const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden,
              const double* __restrict__ dn, double* __restrict__ obj_part) {
    __shared__ double red[4];
    double obj = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        double m = -INFINITY, s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            const double ak = aden[k];
            if (ak == -INFINITY) continue;  // uniform: unsampled state
            const double x = ak - u[k * ld + n];
            if (m == -INFINITY) {
                m = x;
                s = 1.0;
            } else {
                const double d = x - m;
                const double e = exp(-fabs(d));
                s = d > 0.0 ? fma(s, e, 1.0) : s + e;
                m = fmax(m, x);
            }
        }
        const double ldv = m + log(s);
        if (logden) logden[n] = ldv;
        obj = fma(cw[n], dn ? (ldv - dn[n]) : ldv, obj);
    }
    obj = wave_sum(obj);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = obj;
    __syncthreads();
    if (threadIdx.x == 0) obj_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// psum_part[blockIdx.x][k] = sum over this block's samples of exp(anum_k - u_kn - logden_n);
// blockIdx.y selects a group of 8 states.
__global__ void __launch_bounds__(256)
k_colsum_generic(const double* __restrict__ u, int64_t ld, int64_t N, int64_t K,
                 const double* __restrict__ anum, const double* __restrict__ cw, const double* __restrict__ logden,
                 double* __restrict__ psum_part) {
    __shared__ double red[4][8];
    const int64_t k0 = (int64_t)blockIdx.y * 8;
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const double ldv = logden[n], wn = cw[n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = k0 + j;
            if (k < K) acc[j] = fma(wn, exp(anum[k] - u[k * ld + n] - ldv), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double v = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8 && k0 + threadIdx.x < K) {
        const int j = threadIdx.x;
        psum_part[(int64_t)blockIdx.x * K + k0 + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    }
}

// ---------------------------------------------------------------------------------------------
// Reductions and small kernels
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_reduce(const double* __restrict__ part, int64_t nparts, int64_t count, int64_t chunk,
         double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t p0 = (int64_t)blockIdx.y * chunk;
    const int64_t p1 = p0 + chunk < nparts ? p0 + chunk : nparts;
    double s = 0.0;
#pragma unroll 8  // (same summation order; eight loads in flight)
    for (int64_t p = p0; p < p1; ++p) s += part[p * count + i];
    out[(int64_t)blockIdx.y * count + i] = s;
}

// Robust per-state log-sum-exp over samples (log space, like the reference's second logsumexp, mbar_solvers.py:240-241):
//   lognum_k = log sum_n exp(anum_k - u_kn - logden_n)   for ALL states (unsampled ones have no usable shift a priori).
// One wave owns LN_ROWS state rows x a contiguous range of samples, ONE SAMPLE PER LANE per 64-sample tile, and keeps a
// running (max, scaled sum) per state in registers:  d = x - m;  e = exp(-|d|);  s = d > 0 ? s e + 1 : s + e;  m = max(m, x)
// -- one table exponential per matrix element, no cross-lane traffic inside the loop (the previous version reduced
// across the wave twice per state per 512 samples and called the library exp: VALU-bound at 3.8 TB/s).  The LN_ROWS
// row loads of a tile are independent 512-byte requests; two tiles are in flight per wave.  Each wave emits one
// (max, sum) record per state; k_lognum_merge combines them.
constexpr int LN_ROWS = 8;
constexpr int LN_TILE = 64;
template <bool MASKED>
__device__ __forceinline__ void lognum_tile(const double* __restrict__ u, int64_t ld, int64_t N, int64_t K, int64_t k0,
                                            int64_t n, const double* __restrict__ logden, const double (&a)[LN_ROWS],
                                            double (&m)[LN_ROWS], double (&s)[LN_ROWS]) {
    const bool ok = !MASKED || n < N;
    const int64_t nn = ok ? n : 0;
    double v[LN_ROWS];
#pragma unroll
    for (int i = 0; i < LN_ROWS; ++i) v[i] = (k0 + i < K) ? u[(k0 + i) * ld + nn] : 0.0;
    const double nl = -logden[nn];
#pragma unroll
    for (int i = 0; i < LN_ROWS; ++i) {
        double x = (a[i] + nl) - v[i];
        if (MASKED && !ok) x = -INFINITY;
        const double d = x - m[i];  // NaN only for -inf - -inf: laundered to e = 0 by the clamp, and "d > 0" is false
        const double e = exp2s_fast(-fabs(d) * LOG2E_S);
        s[i] = d > 0.0 ? fma(s[i], e, 1.0) : s[i] + e;
        m[i] = fmax(m[i], x);
    }
}
__global__ void __launch_bounds__(256)
k_lognum(const double* __restrict__ u, int64_t ld, int64_t N, int64_t K,
         const double* __restrict__ anum, const double* __restrict__ logden,
         double* __restrict__ pmax, double* __restrict__ psum, int64_t nchunks, int64_t tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    exp_table_init(smem);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nsb = (K + LN_ROWS - 1) / LN_ROWS;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;  // neighbouring waves: same sample range, different state rows
    const int64_t c = w / nsb, k0 = (w % nsb) * LN_ROWS;
    if (c >= nchunks) return;
    double a[LN_ROWS], m[LN_ROWS], s[LN_ROWS];
#pragma unroll
    for (int i = 0; i < LN_ROWS; ++i) {
        a[i] = (k0 + i < K) ? anum[k0 + i] : 0.0;
        m[i] = -INFINITY;
        s[i] = 0.0;
    }
    const int64_t ntiles = (N + LN_TILE - 1) / LN_TILE;
    const int64_t t0 = c * tiles_per_chunk;
    int64_t t1 = t0 + tiles_per_chunk;
    if (t1 > ntiles) t1 = ntiles;
    const int64_t tfull = (t1 * LN_TILE <= N) ? t1 : t1 - 1;  // only the very last tile of the matrix can be ragged
    int64_t t = t0;
    for (; t + 1 < tfull; t += 2) {
        lognum_tile<false>(u, ld, N, K, k0, t * LN_TILE + lane, logden, a, m, s);
        lognum_tile<false>(u, ld, N, K, k0, (t + 1) * LN_TILE + lane, logden, a, m, s);
    }
    for (; t < tfull; ++t) lognum_tile<false>(u, ld, N, K, k0, t * LN_TILE + lane, logden, a, m, s);
    for (; t < t1; ++t) lognum_tile<true>(u, ld, N, K, k0, t * LN_TILE + lane, logden, a, m, s);
#pragma unroll
    for (int i = 0; i < LN_ROWS; ++i) {
        const double mw = wave_max(m[i]);
        const double sc = (m[i] > -INFINITY) ? s[i] * exp(m[i] - mw) : 0.0;  // (mw = -inf only if every lane's m is)
        const double sw = wave_sum(sc);
        if (lane == 0 && k0 + i < K) {
            pmax[(k0 + i) * nchunks + c] = mw;
            psum[(k0 + i) * nchunks + c] = sw;
        }
    }
}

__global__ void __launch_bounds__(256)
k_lognum_merge(const double* __restrict__ pmax, const double* __restrict__ psum, int64_t nchunks,
               double* __restrict__ out_max, double* __restrict__ out_sum) {
    __shared__ double red[8];
    const int64_t k = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double m = -INFINITY;
    for (int64_t c = threadIdx.x; c < nchunks; c += blockDim.x) m = fmax(m, pmax[k * nchunks + c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    double s = 0.0;
    if (m > -INFINITY)
        for (int64_t c = threadIdx.x; c < nchunks; c += blockDim.x) {
            const double pm = pmax[k * nchunks + c];
            if (pm > -INFINITY) s += psum[k * nchunks + c] * exp(pm - m);
        }
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out_max[k] = m;
        out_sum[k] = red[4] + red[5] + red[6] + red[7];
    }
}

template <bool EXP>  // EXP: the weights themselves (mbar_solvers.py:476-486 takes exp of the log weights on the host)
__global__ void __launch_bounds__(256)
k_logw(const double* __restrict__ u, int64_t ld, int64_t N, const double* __restrict__ f,
       const double* __restrict__ logden, double* __restrict__ out, int64_t ld_out) {
    const int64_t k = blockIdx.y;
    const double fk = f[k];
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const double lw = fk - u[k * ld + n] - logden[n];
        out[k * ld_out + n] = EXP ? exp(lw) : lw;
    }
}

// Boundary check of the matrix: bit 0 = some entry is NaN, bit 1 = some entry is -inf, bit 2 = some entry is +inf (legal:
// such a sample simply has zero weight in that state).  The fast exp of the sweeps launders NaN, so a poisoned matrix
// is flagged here once and every reduced output is then reported as NaN, like the reference would compute.
__global__ void __launch_bounds__(256)
k_check_u(const double* __restrict__ u, int64_t ld, int64_t N, int* __restrict__ flags) {
    const int64_t k = blockIdx.y;
    int f = 0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const double v = u[k * ld + n];
        if (v != v) f |= 1;
        if (v == -INFINITY) f |= 2;
        if (v == INFINITY) f |= 4;  // legal (weight zero), but the unclamped Gram sweep must not see it
    }
    if (f) atomicOr(flags, f);
}

// out[n] = logden[n] - alpha * ln(cw[n]): folds per-sample multiplicities into the exponent of the kernels that take
// logden as an input (alpha = 1/2: each MFMA operand of the Gram sweep carries sqrt(c_n); alpha = 1: the log-space
// per-state reduction).  c_n = 0 gives +inf, i.e. weight zero.
__global__ void __launch_bounds__(256)
k_shift_logden(const double* __restrict__ logden, const double* __restrict__ cw, double alpha, int64_t N,
               double* __restrict__ out, const int* __restrict__ ctl, int64_t slot_stride) {
    if (ctl) {
        if (ctl[CTL_DONE] != 0) return;
        logden += (int64_t)ctl[CTL_SLOT] * slot_stride;
    }
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const double c = cw[n];
        out[n] = c > 0.0 ? logden[n] - alpha * log(c) : INFINITY;
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
k_generate_harmonic(double* __restrict__ u, int64_t ld, int64_t N, int64_t K, uint64_t seed,
                    const double* __restrict__ O_k, const double* __restrict__ K_k,
                    const int64_t* __restrict__ cumN, int64_t n_global0) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ng = n_global0 + n;
        int64_t lo = 0, hi = K;  // state s with cumN[s] <= ng < cumN[s+1]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (cumN[mid] <= ng) lo = mid; else hi = mid;
        }
        const uint64_t key = seed * 0xD1342543DE82EF95ull + 2ull * (uint64_t)ng;
        const uint64_t r1 = splitmix64(key), r2 = splitmix64(key + 1ull);
        const double u1 = ((double)(r1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(r2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
        const double x = O_k[lo] + z / sqrt(K_k[lo]);
        for (int64_t l = 0; l < K; ++l) {
            const double d = x - O_k[l];
            u[l * ld + n] = 0.5 * K_k[l] * d * d;
        }
    }
}

// One self-consistent step on the device (single block): sums the `nparts` partial records of psum (the last
// reduction level is folded in here to save a launch), then f'_k = f_k - log(psum_k / N_k) on sampled states
// (mbar_solvers.py:231-242 via s_k), gauge f'[first] = 0 (:588), relative change (:627-631).  The new f is also
// written to `f_hist` (the host picks the accepted iterate after a batch).
__global__ void __launch_bounds__(256)
k_sci_update(const double* __restrict__ part, int64_t nparts, int64_t rows, SciArgs q) {
    __shared__ double red[5];
    extern __shared__ double psum[];  // Kp doubles, then 256 doubles of scratch
    sci_update_block(part, nparts, rows, q, psum, red);
}

// fp64 MFMA peak probe: 4 independent accumulators per wave, nothing else in the loop
// (same kernel as tools/mfma_peak.hip; 64 cycles per instruction per SIMD on gfx950).
__global__ void k_mfma_peak(int iters, double* sink) {
    v4d c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = v4d{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v += c[i][0] + c[i][3];
    if (v == 12345.678) sink[threadIdx.x] = v;  // never true: keeps the loop alive
}

// ---------------------------------------------------------------------------------------------
// Resident probability matrix ("P mode") of the device-resident adaptive loop.
//   P_kn = exp(a0_k - u_kn - logden_n(a0))   (rows of a sample sum to 1; 0 for unsampled / padded states)
// is built ONCE per solve at the starting point a0 = f0 + ln N.  For any other f, with c_k = exp(a_k - a0_k):
//   s_n = sum_k P_kn c_k,   logden_n(f) = logden_n(a0) + log s_n,   p_kn(f) = P_kn c_k / s_n,
// so the two-candidate evaluation sweep needs NO exponential (one FMA dot product and one FMA accumulation per
// element and candidate) and the Gram sweep forms its MFMA operands with ONE multiply per element,
//   G = diag(c) [ sum_n (P_n / s_n)(P_n / s_n)^T ] diag(c),
// instead of the 14-instruction table exponential -- on gfx950 the fp64 matrix instructions and the fp64 VALU share
// one pipe, so every VALU instruction removed from the Gram sweep is kernel time (profiles/r2_gram_ceiling.txt).
// Costs one extra K x N array in HBM (288 GB are there for that) and one build sweep per solve.  Entries of P below
// 1e-308 are flushed to zero: with |a - a0| <= 250 enforced by k_newton (hand-back, then the host rebuilds at the
// current f) the mass lost that way is below 1e-199 of a sample's normaliser.
// ---------------------------------------------------------------------------------------------
// out[n] = rinv[slot][n] * sqrt(cw[n]): per-sample multiplicities folded into both MFMA operands of the P-mode Gram sweep
__global__ void __launch_bounds__(256)
k_rinv_weighted(const double* __restrict__ rinv, const double* __restrict__ cw, int64_t N, double* __restrict__ out,
                const int* __restrict__ ctl, int64_t slot_stride) {
    if (ctl) {
        if (ctl[CTL_DONE] != 0) return;
        rinv += (int64_t)ctl[CTL_SLOT] * slot_stride;
    }
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        out[n] = rinv[n] * sqrt(cw[n]);
}

// Build sweep of P mode: the single-candidate evaluation sweep at the anchor point a0 (log-sum-exp over states, per-state
// sums: the solver's initial gradient) that ALSO writes the normalised probabilities P_kn = e_kn / s_n.  They go back into
// the LDS tile in place of the energies they came from and leave with coalesced 16-byte stores that mirror the DMA
// pattern (8 lanes per 128-byte row), so the pass moves 8 K N bytes in and 8 K N out instead of the separate
// sweep + build (8 + 16).  The reciprocal slot of the anchor point is all ones.
template <int NB>
__device__ __forceinline__ void build_two_groups(char* cbuf, int rd0, int rd1, const double (&a)[NB], double (&acc)[NB],
                                                 double w0, double w1) {
    double x0[NB], x1[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd0);
        x1[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = a[I] - x0[I];
        x1[I] = a[I] - x1[I];
    }
    double m0 = tree_max<NB>(x0), m1 = tree_max<NB>(x1);
    row16_max2(m0, m1);
    const double m2_0 = m0 * LOG2E_S, m2_1 = m1 * LOG2E_S;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = fma(x0[I], LOG2E_S, -m2_0);
        x1[I] = fma(x1[I], LOG2E_S, -m2_1);
    }
    exp2s_batch2<NB>(x0, x1);
    double s0 = tree_sum<NB>(x0), s1 = tree_sum<NB>(x1);
    row16_sum2(s0, s1);
    const double ri0 = recip_fast(s0), ri1 = recip_fast(s1);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] *= ri0;
        x1[I] *= ri1;
        acc[I] = fma(x1[I], w1, fma(x0[I], w0, acc[I]));
        *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + rd0) = x0[I];
        *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + rd1) = x1[I];
    }
}
template <int NB, bool WIDE>
__global__ void __launch_bounds__(64 * lse_waves(NB))
k_build_sweep(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ aden,
              const double* __restrict__ cw, double* __restrict__ P, double* __restrict__ rinv_slot,
              double* __restrict__ psum_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int NDMA = ROWS / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the 16 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    __syncthreads();
    char* buf = smem + EXP_TABLE_BYTES + wave * (2 * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);

    double a[NB], acc[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        acc[I] = 0.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) settle(a[I]);
    rows.live = live_piece_mask<NB>(a, -INFINITY);
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    int64_t t = gw;
    int cur = 0;
    if (t < ntiles) {
        stage_tile<ROWS, true, 0, 1>(u, ld, t * TS, buf, lane, so, rows);
        stage_vec16<true>(cw, t * TS, buf + U_BYTES, lane);
    }
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        const char* wslot = cbuf + U_BYTES;
        const int64_t tn = t + W;
        if (tn < ntiles) {
            char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
            stage_tile<ROWS, true, 0, 1>(u, ld, tn * TS, nbuf, lane, so, rows);
            stage_vec16<true>(cw, tn * TS, nbuf + U_BYTES, lane);
            // vmcnt counts stores too, in issue order: [tile t: NDMA + 1][stores of tile t - W: NDMA + 1][tile tn: NDMA + 1]
            if (t != gw)
                wait_vm<2 * (NDMA + 1)>();
            else
                wait_vm<NDMA + 1>();
        } else {
            wait_vm<0>();
        }
        double w[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) w[g] = *reinterpret_cast<const double*>(wslot + (4 * g + ns) * 8);
        build_two_groups<NB>(cbuf, pos[0], pos[1], a, acc, w[0], w[1]);
        build_two_groups<NB>(cbuf, pos[2], pos[3], a, acc, w[2], w[3]);
        // the tile now holds P: out with it, 16 bytes per lane, 8 lanes per row (the LDS-DMA pattern backwards)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const double2 v = *reinterpret_cast<const double2*>(cbuf + j * 1024 + lane * 16);
            char* dst = reinterpret_cast<char*>(P + (int64_t)(8 * j) * ld + t * TS) + so.off[j & 1];
            *reinterpret_cast<double2*>(dst) = v;
        }
        {   // 1 / s_n = 1 at the anchor point (one store instruction per tile, like the sweeps' logden / reciprocal store)
            const int64_t n = t * TS + lane;
            if (lane < TS && n < N) rinv_slot[n] = 1.0;
        }
        cur ^= 1;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        double v = acc[I];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) psum_part[gw * ROWS + 16 * I + lane] = v;
    }
}

// Build sweep that ALSO accumulates the Gram matrix at the anchor point (the first Hessian of the solve) on the matrix
// cores: the normalised probabilities it writes to P are exactly the MFMA operands, so the separate first Gram sweep of
// the fused loop (one more pass over HBM) is not needed.  One group of 4 samples at a time (the 36 accumulator blocks
// of a 128-state panel leave no room for two groups of exponential temporaries).  wsq: sqrt of the sample multiplicities.
template <int NB, bool WIDE>
__global__ void __launch_bounds__(256, 1)
k_build_gram(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ aden,
             const double* __restrict__ cw, const double* __restrict__ wsq, double* __restrict__ P,
             double* __restrict__ rinv_slot, double* __restrict__ psum_part, double* __restrict__ gram_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int NDMA = ROWS / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 2 * TS * 8;  // + the tile's 16 sample weights and their square roots
    constexpr int NBLK = NB * (NB + 1) / 2;
    constexpr bool PINNED = NBLK > GRAM_AGPR_BLOCKS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    __syncthreads();
    char* buf = smem + EXP_TABLE_BYTES + wave * (2 * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);

    double a[NB], acc[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        a[I] = aden[16 * I + ks];
        acc[I] = 0.0;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) settle(a[I]);
    if constexpr (NB <= 2) rows.live = live_piece_mask<NB>(a, -INFINITY);  // (narrow panels only: see k_gram)
    v4d G[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) G[b] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, true, 0, 1>(u, ld, tile * TS, dst, lane, so, rows);
        stage_vec16<true>(cw, tile * TS, dst + U_BYTES, lane);
        stage_vec16<true>(wsq, tile * TS, dst + U_BYTES + TS * 8, lane);
    };
    int64_t t = gw;
    int cur = 0;
    if (t < ntiles) stage(t, buf);
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        const int64_t tn = t + W;
        if (tn < ntiles) {
            stage(tn, buf + (cur ^ 1) * TILE_BYTES);
            // vmcnt counts stores too, in issue order: [tile t: NDMA + 2][stores of tile t - W: NDMA + 1][tile tn: NDMA + 2]
            if (t != gw)
                wait_vm<(NDMA + 1) + (NDMA + 2)>();
            else
                wait_vm<NDMA + 2>();
        } else {
            wait_vm<0>();
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const double w = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
            const double sw = *reinterpret_cast<const double*>(cbuf + U_BYTES + TS * 8 + (4 * g + ns) * 8);
            double x[NB];
#pragma unroll
            for (int I = 0; I < NB; ++I) x[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int I = 0; I < NB; ++I) x[I] = a[I] - x[I];
            const double m2 = row16_max(tree_max<NB>(x)) * LOG2E_S;
#pragma unroll
            for (int I = 0; I < NB; ++I) x[I] = fma(x[I], LOG2E_S, -m2);
            exp2s_batch<NB>(x);
            const double ri = recip_fast(row16_sum(tree_sum<NB>(x)));
            const bool valid = (t * TS + 4 * g + ns) < N;
            const double opw = valid ? sw : 0.0;
            double p[NB];
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                x[I] *= ri;                                   // P_kn
                acc[I] = fma(x[I], w, acc[I]);                // per-state sums (gradient at the anchor)
                *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + pos[g]) = x[I];
                p[I] = x[I] * opw;                            // MFMA operand (sqrt of the multiplicity; 0 on the padding)
            }
            auto mfma = [&](int b, double xx, double yy) {
                if constexpr (PINNED) {
                    if (b < GRAM_AGPR_BLOCKS)
                        asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(G[b]) : "v"(xx), "v"(yy));
                    else
                        asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(G[b]) : "v"(xx), "v"(yy));
                } else {
                    G[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, yy, G[b], 0, 0, 0);
                }
            };
            if constexpr (PINNED) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            int b = 0;
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int J = I; J < NB; ++J) mfma(b++, p[I], p[J]);
            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
        }
        // the tile now holds P: out with it, 16 bytes per lane, 8 lanes per row (the LDS-DMA pattern backwards)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const double2 v = *reinterpret_cast<const double2*>(cbuf + j * 1024 + lane * 16);
            char* dst = reinterpret_cast<char*>(P + (int64_t)(8 * j) * ld + t * TS) + so.off[j & 1];
            *reinterpret_cast<double2*>(dst) = v;
        }
        {
            const int64_t n = t * TS + lane;
            if (lane < TS && n < N) rinv_slot[n] = 1.0;
        }
        cur ^= 1;
    }
    if constexpr (PINNED) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        double v = acc[I];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) psum_part[gw * ROWS + 16 * I + lane] = v;
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) gram_part[((gw * NBLK + b) * 4 + r) * 64 + lane] = G[b][r];
}

// Two 4-sample groups of a P tile for NF candidates: s = sum_k P c_k (FMA dot + 16-lane sum), r = 1 / s, acc += P w r.
template <int NB, int NF>
__device__ __forceinline__ void psweep_two_groups(const char* cbuf, int rd0, int rd1, const double (&c)[NF][NB],
                                                  double (&acc)[NF][NB], double w0, double w1, double (&r0)[NF],
                                                  double (&r1)[NF]) {
    double x0[NB], x1[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        x0[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd0);
        x1[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + rd1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        double s0 = dot_sum<NB>(x0, c[f]), s1 = dot_sum<NB>(x1, c[f]);
        row16_sum2(s0, s1);
        // (a padded sample has an all-zero column: keep its reciprocal finite, its multiplicity w is 0)
        r0[f] = recip_fast(fmax(s0, 1e-300));
        r1[f] = recip_fast(fmax(s1, 1e-300));
        const double q0 = w0 * r0[f], q1 = w1 * r1[f];
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = fma(x1[I], q1, fma(x0[I], q0, acc[f][I]));
    }
}

// Evaluation sweep over the resident probability matrix for NF candidates given by their multipliers
// cmul[f][k] = exp(a^f_k - a0_k) (0 for unsampled / padded states):
//   rinv^f_n = 1 / sum_k P_kn cmul[f][k]          -> slot vectors (base + slot * stride, like the logden slots)
//   psum_part[wave][f][k] = sum_n w_n P_kn rinv^f_n   (the caller multiplies by cmul[f][k]: that is sum_n p_nk(f))
// Same tile pipeline as k_lse (LDS-DMA, double buffer, one tile of prefetch); no exponential, no table in LDS.
template <int NB, int NF, bool WIDE>
__global__ void __launch_bounds__(64 * lse_waves(NB))
k_psweep(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ cmul,
         const double* __restrict__ cw, double* __restrict__ rinv0, double* __restrict__ rinv1,
         double* __restrict__ psum_part, const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {
        if (ctl[CTL_DONE] != 0) return;
        const int s = ctl[CTL_SLOT];
        rinv1 = rinv0 + (int64_t)((s + 2) % 3) * slot_stride;
        rinv0 = rinv0 + (int64_t)((s + 1) % 3) * slot_stride;
    }
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the 16 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + wave * (2 * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);

    double c[NF][NB], acc[NF][NB];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            c[f][I] = cmul[f * ROWS + 16 * I + ks];
            acc[f][I] = 0.0;
        }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int I = 0; I < NB; ++I) settle(c[f][I]);
    {   // rows whose multipliers are zero for every candidate (states without samples, padding: their rows of P are zero too)
        double cany[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) cany[I] = NF == 2 ? fabs(c[0][I]) + fabs(c[NF - 1][I]) : c[0][I];
        rows.live = live_piece_mask<NB>(cany, 0.0);
    }
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;
    const int gq = ks & 3;               // this lane keeps the reciprocal of sample 4 gq + ns for the store below
    const int fq = (NF == 2 && (ks & 4)) ? 1 : 0;

    int64_t t = gw;
    int cur = 0;
    if (t < ntiles) {
        stage_tile<ROWS, true, 0, 1>(P, ld, t * TS, buf, lane, so, rows);
        stage_vec16<true>(cw, t * TS, buf + U_BYTES, lane);
    }
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        const char* wslot = cbuf + U_BYTES;
        const int64_t tn = t + W;
        if (tn < ntiles) {
            char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
            stage_tile<ROWS, true, 0, 2>(P, ld, tn * TS, nbuf, lane, so, rows);
            stage_vec16<true>(cw, tn * TS, nbuf + U_BYTES, lane);
            constexpr int NEVEN = (ROWS / 8 + 1) / 2 + 1;  // even pieces + the weight slot
            // vmcnt counts stores too: [even(t)][odd(t)][rinv store of tile t - W][even(tn)], and tile t is needed now
            if (t != gw)
                wait_vm<NEVEN + 1>();
            else
                wait_vm<NEVEN>();
        } else {
            wait_vm<0>();
        }
        double w[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) w[g] = *reinterpret_cast<const double*>(wslot + (4 * g + ns) * 8);
        double ra[NF], rb[NF], keep = 0.0;
        psweep_two_groups<NB, NF>(cbuf, pos[0], pos[1], c, acc, w[0], w[1], ra, rb);
        if (gq == 0) keep = ra[fq];
        if (gq == 1) keep = rb[fq];
        __builtin_amdgcn_sched_barrier(0);
        if (tn < ntiles) stage_tile<ROWS, true, 1, 2>(P, ld, tn * TS, buf + (cur ^ 1) * TILE_BYTES, lane, so, rows);
        __builtin_amdgcn_sched_barrier(0);
        psweep_two_groups<NB, NF>(cbuf, pos[2], pos[3], c, acc, w[2], w[3], ra, rb);
        if (gq == 2) keep = ra[fq];
        if (gq == 3) keep = rb[fq];
        {
            const int64_t n = t * TS + 4 * gq + ns;
            double* out = fq ? rinv1 : rinv0;
            // (exactly ONE store instruction per tile and wave -- sample 0 of every tile exists, so it is never skipped --
            // which the vmcnt bookkeeping at the loop top relies on)
            if (n < N && ks < 4 * NF) out[n] = keep;
        }
        cur ^= 1;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(gw * NF + f) * ROWS + 16 * I + lane] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused sweep of the device-resident loop in P mode: ONE pass over the resident probability matrix per iteration.
// For the two candidates (multipliers cmul[0] = f_sci, cmul[1] = f_nr relative to the anchor) it does what k_psweep does --
// normalisers 1 / s_n into the slot vectors, per-state sums -- and, from the SAME tile in LDS, accumulates the Gram
// matrix of the Newton-Raphson candidate on the matrix cores, G'_nr = sum_n (P_n / s_n^nr)(P_n / s_n^nr)^T.  If the loop
// then accepts f_nr (it nearly always does: mbar_solvers.py:607) the next iteration's Hessian is already there and the
// separate Gram sweep is skipped; otherwise that sweep runs (k_select decides, CTL_NEEDGRAM).
// One wave per SIMD owns the register file (288 accumulator registers), so nothing hides behind another wave: every
// instruction that is not under an executing matrix instruction costs its issue slot, fp64 VALU work shares the matrix pipe,
// and an LDS-DMA instruction stalls the wave for tens of cycles.  The tile loop is therefore laid out by hand
// (profiles/r2_fused_sweep_anatomy.txt):
//   * the normalisers are 4x4x4 matrix instructions on a second read of the tile (32 x 16 cycles instead of 64 FMAs,
//     128 DPP moves, 32 adds and eight reciprocals), computed one tile AHEAD between the Gram blocks of groups 2 and 3;
//   * Gram operands are fetched one group ahead, the multiplier operands in two batches, all behind issued matrix work;
//   * the tile after next is requested piece by piece between the Gram blocks of group 3 (the buffer is free then);
//   * the 8 NB per-state accumulations stay where they are written (the compiler would sink them to the loop end and keep
//     all four groups' operands alive), and the loop body is ONE basic block (the register allocator handles the pinned
//     accumulators only then).
// wsq: sqrt of the per-sample multiplicities (= cw itself for plain 0 / 1 weights).
// ---------------------------------------------------------------------------------------------
// Experiment, off: the diagonal 16 x 16 blocks of the full panel as three v_mfma_f64_4x4x4_4b_f64 each (48 matrix-pipe cycles
// instead of 64; parity-green) measured 1 % SLOWER than the 16x16x4 blocks on the same box
// (profiles/r2_fused_sweep_anatomy.txt, section 5).
#ifndef MBAR_FUSED_DIAG4
#define MBAR_FUSED_DIAG4 0
#endif
// Schedule of the 2 NB 4x4x4 steps a group carries (k_fused): how many have been issued once row I of the group's Gram blocks
// is out.  Narrow panels: two per row, the rest after the last-but-one row.  NB >= 4: the first NB (one operand batch) two per
// row, the second batch spread over the rows that remain before the last.
template <int NB>
__host__ __device__ constexpr int fused_steps_done(int I) {
    if (I < 0) return 0;
    if (I >= NB - 2) return 2 * NB;
    if (NB < 4) return 2 * (I + 1) < 2 * NB ? 2 * (I + 1) : 2 * NB;
    const int r1 = (NB + 1) / 2 - 1;  // row that completes the first batch
    if (I <= r1) return 2 * (I + 1) < NB ? 2 * (I + 1) : NB;
    const int rows = NB - 2 - r1;     // rows r1 + 1 .. NB - 2 share the second batch
    return NB + (NB * (I - r1) + rows - 1) / rows;
}
#ifndef MBAR_FUSED_PSUM1_FROM_GRAM
#define MBAR_FUSED_PSUM1_FROM_GRAM 1
#endif
// from this many blocks of 16 states on, the fused sweep leaves the second candidate's per-state sums to k_select (see k_fused)
constexpr int FUSED_PSUM1_FROM_GRAM_NB = MBAR_FUSED_PSUM1_FROM_GRAM ? 8 : 99;
template <int NB, bool WIDE>
__global__ void __launch_bounds__(256, 1)
k_fused(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ cmul,
        const double* __restrict__ cw, const double* __restrict__ wsq, double* __restrict__ rinv0,
        double* __restrict__ rinv1, double* __restrict__ gram_part, double* __restrict__ psum_part,
        const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {
        if (ctl[CTL_DONE] != 0) return;
        const int s = ctl[CTL_SLOT];
        rinv1 = rinv0 + (int64_t)((s + 2) % 3) * slot_stride;
        rinv0 = rinv0 + (int64_t)((s + 1) % 3) * slot_stride;
    }
    constexpr int ROWS = NB * 16;
    constexpr int NDMA = ROWS / 8 + 1;            // tile rows + one piece for the two weight vectors
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 1024;    // (the weights' piece is a full-wave LDS-DMA too: no exec-masked branch)
    constexpr int NBLK = NB * (NB + 1) / 2;
    constexpr bool PINNED = NBLK > GRAM_AGPR_BLOCKS;
    // Full panel: the per-state sums of the SECOND candidate are not accumulated here -- the rows of p sum to one, so they are
    // sum_j c_j G'_kj of the Gram matrix this sweep accumulates for that very candidate, and k_select takes them from the
    // reduced blocks (FUSED_PSUM1_FROM_GRAM_NB): 8 NB fp64 instructions less per tile on a pipe the matrix instructions share.
    constexpr bool ACC1 = NB < FUSED_PSUM1_FROM_GRAM_NB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + wave * (2 * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * nwv + wave;
    const int64_t W = (int64_t)gridDim.x * nwv;
    RowIdentity rows{0};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);

    double acc[2][NB];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[f][I] = 0.0;
    {   // rows whose multipliers are zero for both candidates (states without samples, padding: their rows of P are zero too)
        double cany[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) cany[I] = fabs(cmul[16 * I + ks]) + fabs(cmul[ROWS + 16 * I + ks]);
        if constexpr (NB <= 2) rows.live = live_piece_mask<NB>(cany, 0.0);  // (narrow panels only: see k_gram)
    }
    // The normalisers s_n = sum_k P_kn c_k of both candidates come from the matrix pipe as well: v_mfma_f64_4x4x4_4b
    // contracts over lane bits 4-5 (measured lane map, profiles/r2_mfma4x4_probe.txt: A lane = i + 4 b + 16 k, B lane =
    // j + 4 b + 16 k, D lane = j + 4 b + 16 i), so with the tile read a SECOND time as A(sample = lane & 15, state = 4 step +
    // (lane >> 4)) and the multipliers as B(candidate = lane & 3, same state) 32 instructions of 16 cycles leave
    // s[sample (lane >> 4) + 4 ((lane >> 2) & 3)][candidate lane & 3] in one register -- in place of 64 FMAs, 128 DPP moves,
    // 32 adds and eight reciprocals.  The multiplier operand is a 4 KB table behind the wave buffers.
    constexpr int NSTEP = ROWS / 4;
    {
        double* ctab = reinterpret_cast<double*>(smem + nwv * (2 * TILE_BYTES));
        for (int e = threadIdx.x; e < NSTEP * 16; e += blockDim.x) ctab[e] = (e & 3) < 2 ? cmul[(e & 3) * ROWS + (e >> 2)] : 0.0;
        __syncthreads();
    }
    int apos[4];  // second-layout read: row 4 step + k holds sample n at position (n + (row & 14)) & 15
#pragma unroll
    for (int q = 0; q < 4; ++q) apos[q] = ns * (TS * 8) + (((lane & 15) + (ns & 2) + 4 * q) & 15) * 8;
    v4d G[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) G[b] = v4d{0.0, 0.0, 0.0, 0.0};
    // Diagonal blocks of the full panel as 4 x 4 sub-blocks: with A = B = the Gram operand, v_mfma_f64_4x4x4_4b_f64 gives the four
    // sub-blocks (b, b) of a 16 x 16 block at once; with B rotated by 4 (8) lanes inside each 16-lane row the sub-blocks
    // (b, b -+ 1) ((b, b + 2)): three instructions of 16 cycles cover the block (the wrapped sub-blocks are transposes of
    // wanted ones) where the 16x16x4 instruction spends 64 cycles, half of them below the diagonal.
    constexpr bool DIAG4 = PINNED && (MBAR_FUSED_DIAG4 != 0);
    double Dg[DIAG4 ? NB : 1][3];
#pragma unroll
    for (int I = 0; I < (DIAG4 ? NB : 1); ++I) Dg[I][0] = Dg[I][1] = Dg[I][2] = 0.0;

    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;
    const int sq = ns + 4 * ((lane >> 2) & 3);  // the lane's sample and candidate in the layout the 4x4x4 blocks leave
    const int fq = lane & 3;

    // multiplicities and their roots behind the tile: even 128-byte rows of the piece take cw, odd rows wsq (rows 2-7 repeat them)
    const char* wsrc = reinterpret_cast<const char*>(((lane >> 3) & 1) ? wsq : cw) + (lane & 7) * 16;
    auto stage_w = [&](int64_t tile, char* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + tile * (TS * 8)),
                                         (__attribute__((address_space(3))) void*)(dst + U_BYTES), 16, 0, 0);
    };
    auto stage = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, true, 0, 1>(P, ld, tile * TS, dst, lane, so, rows);
        stage_w(tile, dst);
    };
    // operands of group g in the Gram layout (state 16 I + ks, sample 4 g + ns) + the sample's multiplicity and its root
    auto read_group = [&](const char* tb, int g, double (&x)[NB], double& wg, double& swg) {
        wg = *reinterpret_cast<const double*>(tb + U_BYTES + (4 * g + ns) * 8);
        swg = *reinterpret_cast<const double*>(tb + U_BYTES + TS * 8 + (4 * g + ns) * 8);
#pragma unroll
        for (int I = 0; I < NB; ++I) x[I] = *reinterpret_cast<const double*>(tb + I * (16 * TS * 8) + rd_base + pos[g]);
    };
    uint32_t cop_off = (uint32_t)(nwv * (2 * TILE_BYTES) + (lane >> 4) * 32 + (lane & 3) * 8);
    auto read_step = [&](const char* tb, int st, double& a, double& b) {
        a = *reinterpret_cast<const double*>(tb + st * (4 * TS * 8) + apos[st & 3]);
        b = *reinterpret_cast<const double*>(smem + cop_off + st * 128);
    };
    // (first: the accumulator starts from the inline constant 0 -- a register zeroed by a VALU move right in front of an asm
    // matrix instruction, where the hazard recogniser cannot see it, gave wrong sums)
    auto mfma4 = [&](double& d, double a, double b, bool first) {
        if constexpr (PINNED) {
            if (first)
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
        } else {
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, first ? 0.0 : d, 0, 0, 0);
        }
    };
    auto store_rinv = [&](int64_t tile, double r) {
        // exactly ONE store instruction per tile and wave (sample 0 of every tile exists): the vmcnt bookkeeping needs it
        const int64_t n = tile * TS + sq;
        double* out = fq ? rinv1 : rinv0;
        if (n < N && fq < 2) out[n] = r;
    };
    // Software pipeline over the wave's tiles t_0, t_1, ... (two LDS buffers): while the Gram blocks of tile t_i issue, the
    // 4x4x4 blocks of tile t_{i+1}'s normalisers are slipped in between them (their LDS operands requested a row of blocks
    // earlier), the Gram operands are fetched one group ahead, and tile t_{i+2} is requested into t_i's buffer as soon as
    // its last group has been read -- no LDS round trip and no HBM latency is left exposed in the loop.
    //   vmcnt, in issue order, at the top of iteration i: [tile t_{i+1}: NDMA] [store of tile t_i's reciprocals: 1]
    int64_t t = gw;
    int cur = 0;
    double rcur = 0.0;
    double uv[2][NB], w[2], sw[2];
    if (t < ntiles) {
        stage(t, buf);
        if (t + W < ntiles) {
            stage(t + W, buf + TILE_BYTES);
            wait_vm<NDMA>();
        } else {
            wait_vm<0>();
        }
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int st = 0; st < NSTEP; st += 2) {
            double a0, b0, a1, b1;
            read_step(buf, st, a0, b0);
            read_step(buf, st + 1, a1, b1);
            sa = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, sb, 0, 0, 0);
        }
        // (a padded sample has an all-zero column: keep its reciprocal finite, its multiplicity is 0)
        rcur = recip_fast(fmax(sa + sb, 1e-300));
        store_rinv(t, rcur);
        read_group(buf, 0, uv[0], w[0], sw[0]);
    }
#if defined(MBAR_EXPERIMENT_FUSED_WAITS)  // instrumented build (profiles/r2_fused_sweep_anatomy.txt): cycles per tile and in the waits
    long long dbg_vm = 0, dbg_n = 0, dbg_b2 = 0;
    const long long dbg_t0 = clock64();
#endif
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
        const int64_t tn = t + W, tnn = t + 2 * W;
        const int64_t tstage = tnn < ntiles ? tnn : t;
        // (the multiplier table never changes: without this the compiler keeps all of it in 8 NB registers)
        asm volatile("" : "+v"(cop_off));
        // (four accumulators in rotation: the asm 4x4x4 blocks are invisible to the hazard recogniser, and a dependent one
        // needs four wait states after its predecessor)
        double sacc[4];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int gc = g & 1, gn = gc ^ 1;
#if defined(MBAR_EXPERIMENT_FUSED_WAITS)
            if (g == 2) {
                const long long c0 = clock64();
                wait_vm<1>();
                const long long c1 = clock64();
                const long long c2 = clock64();
                dbg_vm += (c1 - c0) - (c2 - c1);
                dbg_n += 1;
            }
#else
            if (g == 2) wait_vm<1>();  // tile t_{i+1}, requested three quarters of an iteration ago
#endif
            double r0, r1;
            switch (g) {
                case 0: r0 = row16_bcast<0>(rcur); r1 = row16_bcast<1>(rcur); break;
                case 1: r0 = row16_bcast<4>(rcur); r1 = row16_bcast<5>(rcur); break;
                case 2: r0 = row16_bcast<8>(rcur); r1 = row16_bcast<9>(rcur); break;
                default: r0 = row16_bcast<12>(rcur); r1 = row16_bcast<13>(rcur); break;
            }
            const double q0 = w[gc] * r0, q1 = w[gc] * r1;
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                acc[0][I] = fma(uv[gc][I], q0, acc[0][I]);
                if constexpr (ACC1) acc[1][I] = fma(uv[gc][I], q1, acc[1][I]);
                // (pinned here: left alone, the compiler sinks all 8 NB updates to the end of the iteration and keeps
                // the operands of all four groups alive for them)
                settle(acc[0][I]);
                if constexpr (ACC1) settle(acc[1][I]);
            }
            // (a padded sample needs no mask: its multiplicity and the root of it are stored as zeros)
            const double rin = r1 * sw[gc];  // operand of the Newton-Raphson candidate's Gram matrix
            double p[NB];
#pragma unroll
            for (int I = 0; I < NB; ++I) p[I] = uv[gc][I] * rin;
            // Operands of the next group and of the 4x4x4 steps this group carries (the next tile's normalisers ride on this
            // tile's groups 2 and 3: 2 NB steps each, their operands fetched in two batches of NB for wide panels), the weights'
            // piece of the tile after next: with pinned accumulators these are issued BEHIND the group's first Gram blocks
            // (request_next / request_steps below), otherwise here and the compiler places them.
            constexpr int B1 = NB >= 4 ? NB : 2 * NB;
            double opa[B1], opb[B1];
            auto request_next = [&]() {
                if (g < GROUPS - 1) {
                    read_group(cbuf, g + 1, uv[gn], w[gn], sw[gn]);
                } else {
                    // (every read of this tile was issued a group ago and has been consumed: its buffer can take the tile
                    // after next -- past the end this tile is simply requested again and never looked at, so that the loop
                    // body stays ONE basic block: the register allocator handles the 288 pinned accumulator registers only then)
                    read_group(nbuf, 0, uv[gn], w[gn], sw[gn]);
                }
            };
            auto request_steps = [&]() {
                if (g >= 2) {
#pragma unroll
                    for (int q = 0; q < B1; ++q) read_step(nbuf, (g - 2) * 2 * NB + q, opa[q], opb[q]);
                }
            };
            if constexpr (!PINNED) {
                request_next();
                request_steps();
            }
            auto mfma = [&](int b, double x, double y) {
                if constexpr (PINNED) {
                    if (DIAG4 || b < GRAM_AGPR_BLOCKS)  // (without the diagonal blocks all 28 fit the AGPRs)
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(G[b]) : "v"(x), "v"(y));
                    else
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(G[b]) : "v"(x), "v"(y));
                } else {
                    G[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, G[b], 0, 0, 0);
                }
            };
            // rotated copies of the diagonal operands, three in flight: panel 0's here, panel I + 1's behind the first 16x16x4
            // block of row I (the last panel's one row earlier, so that a VALU result never meets an asm matrix instruction
            // without matrix work in between)
            double rot[3][2];
            auto rotate = [&](int I) {
                rot[I % 3][0] = dpp_move<0x124>(p[I]);  // row_ror:4
                rot[I % 3][1] = dpp_move<0x128>(p[I]);  // row_ror:8
            };
            auto diag4 = [&](double& d, double x, double y) {
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y));
            };
            if constexpr (DIAG4) rotate(0);
            if constexpr (PINNED) {  // (asm MFMAs are opaque to the scheduler and the hazard recogniser: see k_gram)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            constexpr int NM = DIAG4 ? NBLK - NB : NBLK;  // 16x16x4 instructions per group
            int b = 0, nm = 0;
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                if (NB == 1 && g >= 2) {
                    mfma4(sacc[2 * (g & 1)], opa[0], opb[0], true);
                    mfma4(sacc[2 * (g & 1) + 1], opa[1], opb[1], true);
                }
                if constexpr (DIAG4) {
                    diag4(Dg[I][0], p[I], p[I]);
                    diag4(Dg[I][1], p[I], rot[I % 3][0]);
                    diag4(Dg[I][2], p[I], rot[I % 3][1]);
                }
#pragma unroll
                for (int J = I; J < NB; ++J) {
                    if (DIAG4 && J == I) {
                        ++b;
                        continue;
                    }
                    mfma(b, p[I], p[J]);
                    if constexpr (PINNED) {
                        if (nm == 0 || nm == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (nm == 0) request_next(); else request_steps();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (DIAG4) {
                        const int nxt = (J == I + 1 && I + 1 < NB - 1) ? I + 1 : ((I == NB - 3 && J == NB - 1) ? NB - 1 : -1);
                        if (nxt > 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            rotate(nxt);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (g == GROUPS - 1) {  // pieces [nm NDMA / NM, (nm + 1) NDMA / NM) of the tile after next; the last = weights
#pragma unroll
                        for (int j = nm * NDMA / NM; j < (nm + 1) * NDMA / NM; ++j) {
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                            if (j < ROWS / 8)
                                stage_piece<true>(P + rows(8 * j) * ld + rows.cols(j, tstage * TS), so.off[j & 1], cbuf + j * 1024, lane);
                            else
                                stage_w(tstage, cbuf);
                            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    ++b;
                    ++nm;
                }
                // 4x4x4 steps of the next tile after every row of blocks but the last (so that the accumulators are long
                // complete when the VALU reads them): fused_steps_done(I) of the group's 2 NB steps are issued by row I
                if (g >= 2 && NB > 1 && I < NB - 1) {
#pragma unroll
                    for (int q = fused_steps_done<NB>(I - 1); q < fused_steps_done<NB>(I); ++q)
                        mfma4(sacc[q & 3], opa[q % B1], opb[q % B1], g == 2 && q < 4);
                    if (B1 < 2 * NB && fused_steps_done<NB>(I - 1) < B1 && fused_steps_done<NB>(I) >= B1) {
#pragma unroll
                        for (int q = 0; q < B1; ++q) read_step(nbuf, (g - 2) * 2 * NB + B1 + q, opa[q], opb[q]);
                    }
#if defined(MBAR_EXPERIMENT_FUSED_WAITS)
                    if (B1 < 2 * NB && fused_steps_done<NB>(I - 2) < B1 && fused_steps_done<NB>(I - 1) >= B1) {
                        // (the point where the second batch is first used: time the wait for it)
                        const long long c0 = clock64();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const long long c1 = clock64();
                        const long long c2 = clock64();
                        dbg_b2 += (c1 - c0) - (c2 - c1);
                    }
#endif
                }
            }
            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
        }
        // (the wait states between the last 4x4x4 block and the VALU reading its result; tied to the accumulators so that the
        // scheduler cannot move the additions in front of it)
        if constexpr (PINNED)
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]));
        rcur = recip_fast(fmax((sacc[0] + sacc[1]) + (sacc[2] + sacc[3]), 1e-300));
        store_rinv(tn, rcur);  // (past the last tile every lane is beyond N)
        cur ^= 1;
    }
#if defined(MBAR_EXPERIMENT_FUSED_WAITS)
    if (lane == 0 && (gw % 341) == 0)
        printf("DBG wave %ld: tiles %lld total %lld cycles (%lld per tile), vm wait %lld per tile, batch-2 lgkm wait %lld per tile (2 per tile)\n",
               (long)gw, dbg_n, clock64() - dbg_t0, (clock64() - dbg_t0) / (dbg_n ? dbg_n : 1), dbg_vm / (dbg_n ? dbg_n : 1),
               dbg_b2 / (dbg_n ? dbg_n : 1));
#endif
    if constexpr (PINNED) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v = acc[f][I];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) psum_part[(gw * 2 + f) * ROWS + 16 * I + lane] = v;
        }
    }
    {
        int b = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J, ++b) {
                if (DIAG4 && J == I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) gram_part[((gw * NBLK + b) * 4 + r) * 64 + lane] = G[b][r];
            }
    }
    if constexpr (DIAG4) {
        // the 4 x 4 sub-blocks into the 16 x 16 record the 16x16x4 instruction would have left (element (r, c) at r 16 + c):
        // lane j + 4 b + 16 i holds (4 b + i, 4 b' + j), b' = the block the rotated operand came from -- asked of the same
        // DPP controls, so the direction of the rotation is not assumed.  (b, b) and (b, b + 2) land once each, (b, b -+ 1) also
        // transposed: all 16 sub-blocks of the record are written.
        const int jj = lane & 3, bb = (lane >> 2) & 3, ii = lane >> 4;
        const int b1 = __builtin_amdgcn_update_dpp(0, bb, 0x124, 0xF, 0xF, true);
        const int b2 = __builtin_amdgcn_update_dpp(0, bb, 0x128, 0xF, 0xF, true);
        const int row = 4 * bb + ii;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double* rec = gram_part + (gw * NBLK + (I * NB - (I * (I - 1)) / 2)) * 256;
            rec[row * 16 + 4 * bb + jj] = Dg[I][0];
            rec[row * 16 + 4 * b1 + jj] = Dg[I][1];
            rec[(4 * b1 + jj) * 16 + row] = Dg[I][1];
            rec[row * 16 + 4 * b2 + jj] = Dg[I][2];
        }
    }
}

// Same reduction as k_reduce for TWO partial-record arrays with the same number of records in one launch
// (blocks [0, gxA) work on A, the rest on B; identical summation order).
__global__ void __launch_bounds__(256)
k_reduce2(const double* __restrict__ partA, int64_t countA, const double* __restrict__ partB, int64_t countB,
          int64_t nparts, int64_t chunk, double* __restrict__ outA, double* __restrict__ outB) {
    const int64_t gxA = (countA + 255) / 256;
    const bool isB = (int64_t)blockIdx.x >= gxA;
    const double* part = isB ? partB : partA;
    const int64_t count = isB ? countB : countA;
    double* out = isB ? outB : outA;
    const int64_t i = ((int64_t)blockIdx.x - (isB ? gxA : 0)) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t p0 = (int64_t)blockIdx.y * chunk;
    const int64_t p1 = p0 + chunk < nparts ? p0 + chunk : nparts;
    double s = 0.0;
#pragma unroll 8
    for (int64_t p = p0; p < p1; ++p) s += part[p * count + i];
    out[(int64_t)blockIdx.y * count + i] = s;
}

// ---------------------------------------------------------------------------------------------
// Device-resident adaptive iteration (mbar_solvers.py:575-640): the K x K work between the two sweeps.
// ---------------------------------------------------------------------------------------------
// Element (ki, kj) of the reduced Gram panel.  Only the upper triangle is read (like the host-side unpack).
__device__ __forceinline__ double gram_elem(const double* __restrict__ g, int nb, int ki, int kj) {
    if (ki > kj) {
        const int t = ki;
        ki = kj;
        kj = t;
    }
    const int I = ki >> 4, J = kj >> 4;
    const int b = I * nb - (I * (I - 1)) / 2 + (J - I);
    return g[(int64_t)b * 256 + (ki & 15) * 16 + (kj & 15)];
}

// fixed-order sums / maxima over a workgroup of 256 threads
__device__ __forceinline__ double block256_sum(double v, double* red /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block256_max(double v, double* red /*[4]*/) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Both candidates and the inputs of the candidate sweep from the Newton direction xs (LDS: xs[i] for the i-th sampled state,
// xs[0] = 0 -- the gauge), shared by the register Gauss-Jordan solve (k_newton, up to 127 unknowns) and the blocked Cholesky
// solve (k_chol_*, up to 255).  `bad`: the elimination met a pivot that counts as zero.  All threads of the workgroup call it.
template <int NT>
__device__ __forceinline__ void newton_tail(const AdaptArgs& q, const double* xs, bool bad, const double* s_f, const double* s_ps,
                                            const double* s_nk, const double* s_ln, const double* s_a0, const int* smp,
                                            const int* pos, int tid) {
    const double gamma = q.prm[0];
    const int first = smp[0];
    const double shift = s_f[first] - log(s_ps[first] / s_nk[first]);
    int flags = bad ? 1 : 0;
    // Fused sweep: the Gram matrix it accumulates is that of the SECOND multiplier row.  While self-consistent steps are
    // forced (:607, sci_iter < min_sc_iter) that row is the self-consistent candidate's -- the one that WILL be accepted --
    // so that the speculation is never thrown away (the reference's default min_sc_iter = 2 cost two extra sweeps before).
    const bool swap = q.fused && q.ctl[CTL_SCI] < (int)q.prm[2];
    const int o_sci = swap ? q.Kp : 0, o_nr = swap ? 0 : q.Kp;
    if (tid == 0) q.ctl[CTL_SPEC] = swap ? 0 : 1;
    for (int k = tid; k < q.Kp; k += NT) {
        const bool sampled = k < q.K && s_nk[k] > 0.0;
        const double fk = s_f[k];
        double fs = fk, fn = fk, a0 = -INFINITY, rt = 1.0;
        if (sampled) {
            fn = fk - gamma * xs[pos[k]];                        // :584
            fs = (fk - log(s_ps[k] / s_nk[k])) - shift;          // :587-588
            a0 = fs + s_ln[k];
            const double a1 = fn + s_ln[k];
            if (q.pmode) {
                // multipliers of both candidates relative to the build point of P; too far from it (250 kT: the
                // flushed tail of P would start to matter near e^700) hands back, the host rebuilds P at the current f
                const double d0 = a0 - s_a0[k], d1 = a1 - s_a0[k];
                if (!(fabs(d0) < 250.0) || !(fabs(d1) < 250.0)) flags |= 2;
                a0 = exp(d0);
                rt = exp(d1);
            } else {
                const double d = a1 - a0;
                rt = exp(d);
                if (!(fabs(d) < 300.0)) flags |= 2;
            }
            if (!isfinite(fs) || !isfinite(fn)) flags |= 4;
        } else if (q.pmode) {
            a0 = 0.0;
            rt = 0.0;
        }
        q.cand[k] = fs;
        q.cand[q.Kp + k] = fn;
        q.ratio[k] = rt;
        q.aden[o_sci + k] = a0;
        q.aden[o_nr + k] = rt;
    }
    flags = __syncthreads_or(flags);
    if (flags != 0 && tid == 0) {
        q.ctl[CTL_REASON] = (flags & 1) ? 1 : ((flags & 4) ? 3 : 2);
        q.ctl[CTL_DONE] = 2;
    }
}

// Newton direction + both candidates, ONE workgroup of T x T threads with an R x R tile each (up to R T - 1 unknowns:
// 8 x 8 threads x 4 x 4 -> 31, 16 x 16 x 4 x 4 -> 63, 16 x 16 x 8 x 8 -> 127).
//   H = diag(psum) - G on the sampled states, g = psum - N_k (:581, :284-292); gauge x[first] = 0, so the system is the
//   (m-1) x (m-1) SPD block of H -- the same vector as lstsq(H, g) minus its first component (:582-583).
// The augmented matrix [A | b] lives in REGISTERS, an R x R tile per thread in a CYCLIC layout (thread (ty, tx): rows
// ty + T r, columns tx + T c; column R T - 1 holds b).  A step is a latency chain LDS write -> barrier -> LDS read -> rcp ->
// FMA; few waves matter more than few FMAs per thread (127 unknowns: 4 waves with 8 x 8 tiles 80 us, 16 waves with 4 x 4
// tiles 95 us).  Replacing the barrier by per-wave flag words in LDS, so that the next pivot column is published before the
// rest of the tile is updated, was slower still (112 us: the polling loop costs more than the barrier).  Gauss-Jordan without pivoting (A is SPD; the pivots are the
// squares of the Cholesky diagonal, so "pivot <= 0" is exactly the Cholesky breakdown test of the host path): step j
// needs only column j, which its owners publish through a double-buffered LDS vector -- row j of the live block is the
// same vector by symmetry -- so a step is one barrier, 2 R + 1 LDS reads and at most R R FMAs per thread, and there are no
// triangular solves: x_i = b_i / pivot_i at the end.  Pivots are taken TWO per barrier (both columns are published as they
// stand and every thread reconstructs what the second step would have read): half the latency chains for ~10 % more
// arithmetic -- 127 unknowns 68 -> 65 us (the 8 x 8 tiles are arithmetic-bound by then), 39 / 63 unknowns ~-30 %.  The whole kernel is bound by the fp64 issue rate of ONE compute
// unit, so it is written for instruction count:
//   * columns left of the pivot are never read again; they are left stale (whole tile columns c < j / T: skipped
//     statically, the step loop is unrolled over j / T) or take garbage, and the pivots are kept in their own vector;
//   * b_j travels in slot R T - 1 of the column vector (row R T - 1 is always padding: its multiplier is then garbage, which
//     only ever touches that row), so the b column needs no special case;
//   * the pivot row is excluded by zeroing ONE multiplier under a compare, not by a select per row.
// A non-positive pivot, candidates more than 300 kT apart (the fused two-candidate sweep shares one shift) or a
// non-finite candidate hand the solve back to the host loop (CTL_DONE = 2).
// Outputs: cand = (f_sci, f_nr), ratio = exp(aden_nr - aden_sci), aden = (aden_sci, ratio) for the sweep.
template <int T, int R>
__device__ __forceinline__ void newton_body(const AdaptArgs& q) {  // (T * T threads; the pointers of q may be LDS or global)
    constexpr int NC = R * T, NT = T * T;
    __shared__ double colbuf[2][2][NC];  // [parity of the step][column j, column j + 1]
    __shared__ double pv[NC], rh[NC], xs[NC + 1];
    __shared__ double s_f[128], s_ps[128], s_nk[128], s_ln[128];  // per-state vectors (Kp <= 128)
    __shared__ double s_cc[128], s_a0[128];                       // P mode: current multipliers, build point
    __shared__ int smp[NC + 1], pos[128];                          // sampled list (m <= NC) and its inverse
    if (q.ctl[CTL_DONE] != 0) return;
    const int tid = threadIdx.x, ty = tid / T, tx = tid % T;
    const int M = q.m - 1, nb = q.Kp / 16;
    // Everything the kernel indexes indirectly goes through LDS first: a dependent global load costs ~1 us, and the
    // tile set-up below would otherwise chain two of them in front of each of its 16 Gram loads.
    for (int k = tid; k < q.Kp; k += NT) {
        s_f[k] = k < q.K ? q.f[k] : 0.0;
        s_ps[k] = q.psum[k];
        s_nk[k] = q.Nk[k];
        s_ln[k] = q.lnNk[k];
        s_cc[k] = q.pmode ? (q.fused ? q.cgram[k] : q.ccur[k]) : 1.0;  // the multipliers the Gram sweep left out
        s_a0[k] = q.pmode ? q.a0[k] : 0.0;
        pos[k] = 0;
    }
    for (int i = tid; i < q.m; i += NT) smp[i] = q.sampled[i];
    __syncthreads();
    for (int i = tid; i < q.m; i += NT) pos[smp[i]] = i;

    int ki[R], kj[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = ty + T * r, k = tx + T * r;
        ki[r] = i < M ? smp[i + 1] : 0;
        kj[r] = k < M ? smp[k + 1] : 0;
    }
    double A[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)  // R x R independent loads (always a valid address; masked below)
#pragma unroll
        for (int c = 0; c < R; ++c) A[r][c] = -gram_elem(q.gram_red, nb, ki[r], kj[c]);
    if (q.pmode) {  // the P-mode Gram sweep leaves the two per-state factors exp(a - a0) to be applied here
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < R; ++c) A[r][c] *= s_cc[ki[r]] * s_cc[kj[c]];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int i = ty + T * r, k = tx + T * c;
            double v = A[r][c];
            if (i < M) {
                if (k < M) {
                    if (i == k) v += s_ps[ki[r]];
                } else {
                    v = (k == NC - 1) ? s_ps[ki[r]] - s_nk[ki[r]] : 0.0;
                }
            } else {
                v = (i == k && k != NC - 1) ? 1.0 : 0.0;  // padding rows: identity, never a pivot, multiplier 0
            }
            A[r][c] = v;
        }
    }
    // pivots below eps * M * (largest per-state sum, which bounds the diagonal of H) count as zero like the singular values
    // numpy.linalg.lstsq drops (:582): the host path then takes the pseudo-inverse
    double pmax = 0.0;
    for (int i = 0; i < q.m; ++i) pmax = fmax(pmax, s_ps[smp[i]]);
    const double piv_thr = pmax * 2.220446049250313e-16 * (double)(M > 0 ? M : 1);
    bool bad = false;
#pragma unroll
    for (int jc = 0; jc < R; ++jc) {
        const int jend = M < T * (jc + 1) ? M : T * (jc + 1);
        int j = T * jc;
        // Two pivots per barrier (the step is a latency chain, not arithmetic): columns j and j + 1 are published as they
        // stand, every thread forms the multiplier l = A[j+1][j] / p1 of row j + 1, the second pivot p2 = A[j+1][j+1] - l A[j+1][j]
        // and, for its rows and columns, what the second elimination step would have read:
        //   column j + 1 after step j: c2_i = A[i][j+1] - m1_i A[j][j+1],   row j + 1 after step j: r2_k = A[j+1][k] - l A[j][k]
        // and then applies both rank-1 updates at once.  Pivot rows: m1_j = 0, m2_{j+1} = 0 (row j is still cleared of its
        // (j + 1) entry by the second pivot, row j + 1 of its j entry by the first).
        for (; j + 1 < jend; j += 2) {
            const int jt = j - T * jc;
            double* ca = colbuf[(j >> 1) & 1][0];
            double* cb = colbuf[(j >> 1) & 1][1];
            if (tx == jt || tx == jt + 1) {  // owners of columns j and j + 1
                double* cx = tx == jt ? ca : cb;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < R - 1 || ty != T - 1) cx[ty + T * r] = A[r][jc];  // (slot NC-1 belongs to b)
            }
            if (tx == T - 1) {
                if (ty == jt) ca[NC - 1] = A[jc][R - 1];      // b_j
                if (ty == jt + 1) cb[NC - 1] = A[jc][R - 1];  // b_{j+1}
            }
            __syncthreads();
            const double p1 = ca[j], a12 = ca[j + 1], a22 = cb[j + 1];
            const double inv1 = recip_fast(p1);
            const double l = a12 * inv1;
            const double p2 = fma(-l, a12, a22);
            const double inv2 = recip_fast(p2);
            if (tid == 0) {
                pv[j] = p1;
                pv[j + 1] = p2;
            }
            if (!(p1 > piv_thr) || !isfinite(p1) || !(p2 > piv_thr) || !isfinite(p2)) bad = true;  // the same in every thread
            double m1[R], m2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double x1 = ca[ty + T * r], x2 = cb[ty + T * r];
                m1[r] = x1 * inv1;
                if (ty == jt && r == jc) m1[r] = 0.0;  // pivot row j
                m2[r] = fma(-m1[r], a12, x2) * inv2;
                if (ty == jt + 1 && r == jc) m2[r] = 0.0;  // pivot row j + 1
            }
#pragma unroll
            for (int c = jc; c < R; ++c) {
                const double r1 = ca[tx + T * c];
                const double r2 = fma(-l, r1, cb[tx + T * c]);
#pragma unroll
                for (int r = 0; r < R; ++r) A[r][c] = fma(-m2[r], r2, fma(-m1[r], r1, A[r][c]));
            }
        }
        for (; j < jend; ++j) {  // (an odd pivot left over in this tile column)
            const int jt = j - T * jc;
            double* cb = colbuf[(j >> 1) & 1][0];
            if (tx == jt) {  // owners of column j
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < R - 1 || ty != T - 1) cb[ty + T * r] = A[r][jc];  // (slot NC-1 belongs to b_j)
            }
            if (ty == jt && tx == T - 1) cb[NC - 1] = A[jc][R - 1];  // b_j
            __syncthreads();
            const double piv = cb[j];
            if (tid == 0) pv[j] = piv;
            if (!(piv > piv_thr) || !isfinite(piv)) bad = true;  // the same value in every thread
            const double inv = recip_fast(piv);
            double mr[R];
#pragma unroll
            for (int r = 0; r < R; ++r) mr[r] = cb[ty + T * r] * inv;
            if (ty == jt) mr[jc] = 0.0;  // the pivot row itself
#pragma unroll
            for (int c = jc; c < R; ++c) {
                const double rv = cb[tx + T * c];
#pragma unroll
                for (int r = 0; r < R; ++r) A[r][c] = fma(-mr[r], rv, A[r][c]);
            }
        }
    }
    if (tx == T - 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) rh[ty + T * r] = A[r][R - 1];
    }
    __syncthreads();
    if (tid == 0) xs[0] = 0.0;
    if (tid < M) xs[tid + 1] = rh[tid] / pv[tid];
    __syncthreads();

    newton_tail<NT>(q, xs, bad, s_f, s_ps, s_nk, s_ln, s_a0, smp, pos, tid);
}
template <int T, int R>
__global__ void __launch_bounds__(T * T)
k_newton(AdaptArgs q) {
    newton_body<T, R>(q);
}

// ---------------------------------------------------------------------------------------------
// The same K x K step for 128 .. 255 unknowns (129 .. 256 states): the register Gauss-Jordan solve above holds 127 unknowns
// in one workgroup's registers and no more, so here the gauge-fixed Newton system is solved by a BLOCKED right-looking
// Cholesky factorisation of the matrix in device memory (512 KB: it lives in L2), one pair of small kernels per block column
// of CB = 32 -- kernel boundaries are the grid barriers, everything is enqueued ahead like the rest of the iteration:
//   k_chol_setup   A = H[1:, 1:] (lower triangle) from the reduced Gram blocks, with b = g[1:] appended as ROW M: the
//                  factorisation then leaves y = L^-1 b in that row, i.e. the forward substitution rides along;
//   k_chol_panel   (one workgroup) Cholesky of the 32 x 32 diagonal block by one wave (a lane per row, the finished column
//                  broadcast through LDS), then every row below solves against it (a thread per row, incl. row M);
//   k_chol_update  (one workgroup per 32 x 32 tile of the trailing lower triangle, incl. row M) A_ik -= L_i L_k^T;
//   k_chol_finish  (one workgroup) back substitution L^T x = y in blocks of 32, then the candidates (newton_tail).
// A pivot that counts as zero (the threshold of k_newton / the host path) hands the solve back (CTL_DONE = 2).
// ~18 launches of 2-5 us for 255 unknowns: ~0.1 ms against >= 4 ms of sweeps at these state counts (the host-driven loop paid
// two synchronisations, a 0.5 MB download and a 0.57 ms host factorisation per iteration).
// ---------------------------------------------------------------------------------------------
constexpr int CHOL_NP = 256;  // row pitch of the workspace (unknowns + the appended right-hand-side row <= 256)
constexpr int CB = 32;
__global__ void __launch_bounds__(256)
k_chol_setup(AdaptArgs q, double* __restrict__ Aw, double* __restrict__ thr_out) {
    __shared__ double red[4];
    if (q.ctl[CTL_DONE] != 0) return;
    const int M = q.m - 1, nb = q.Kp / 16;
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4), k = blockIdx.x * 16 + (threadIdx.x & 15);
    if (i > M || k > i || k >= M) {
        // (the whole workspace is defined: the panel kernel reads and writes full 32-column runs of its rows)
        if (i < CHOL_NP && k < CHOL_NP) Aw[i * CHOL_NP + k] = (i == k) ? 1.0 : 0.0;
    } else {
        const int kk = q.sampled[k + 1];
        double v;
        if (i < M) {
            const int ki = q.sampled[i + 1];
            v = -gram_elem(q.gram_red, nb, ki, kk);
            if (q.pmode) {
                const double* cc = q.fused ? q.cgram : q.ccur;  // the multipliers the Gram sweep left out
                v *= cc[ki] * cc[kk];
            }
            if (i == k) v += q.psum[ki];
        } else {
            v = q.psum[kk] - q.Nk[kk];  // row M: the gradient (:284-292)
        }
        Aw[i * CHOL_NP + k] = v;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {  // (uniform per workgroup)
        // pivots below eps * M * (largest per-state sum, which bounds the diagonal of H) count as zero (see k_newton)
        const int t = threadIdx.x;
        double pm = 0.0;
        for (int s = t; s < q.m; s += 256) pm = fmax(pm, q.psum[q.sampled[s]]);
        pm = block256_max(pm, red);
        if (t == 0) thr_out[0] = pm * 2.220446049250313e-16 * (double)(M > 0 ? M : 1);
    }
}

// 1 / sqrt(d) for d > 0: hardware estimate + two Newton steps (the sqrt and the divide each expand to ~30 instructions)
__device__ __forceinline__ double rsqrt_fast(double d) {
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(fma(-d * r, r, 1.0), 0.5, 1.0);
    r = r * fma(fma(-d * r, r, 1.0), 0.5, 1.0);
    return r;
}

__global__ void __launch_bounds__(256)
k_chol_panel(AdaptArgs q, double* __restrict__ Aw, const double* __restrict__ thr_in, int j0) {
    __shared__ double D[CB][CB + 1];   // the factor of the diagonal block (lower triangle), row-major
    __shared__ double colv[CB], rdiag[CB];
    __shared__ int s_bad;
    if (q.ctl[CTL_DONE] != 0) return;
    const int M = q.m - 1, tid = threadIdx.x;
    const int nc = M - j0 < CB ? M - j0 : CB;  // columns of this panel (the last one may be short: padded with the identity)
    const double thr = thr_in[0];
    if (tid == 0) s_bad = 0;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c2 = e % CB;
        D[r][c2] = (r < nc && c2 <= r) ? Aw[(j0 + r) * CHOL_NP + j0 + c2] : (r == c2 ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {  // one wave: lane r owns row r of the block (the upper lanes repeat rows 0 .. 31: same values, same addresses;
        //                a store under `tid < CB` inside the unrolled loop keeps the row array out of registers)
        const int r = tid & (CB - 1);
        double row[CB];
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) row[c2] = D[r][c2];
        bool bad = false;
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) {
            // column c2 is final for rows >= c2 once the updates of columns < c2 are in: publish it
            colv[r] = row[c2];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double d = colv[c2];
            if (c2 < nc && (!(d > thr) || !isfinite(d))) bad = true;
            const double inv = rsqrt_fast(d);       // 1 / L[c2][c2]
            const double t = row[c2] * (inv * inv);  // L[r][c2] / L[c2][c2]
            // trailing entries of this row: row[k] -= L[r][c2] L[k][c2] = t colv[k]
#pragma unroll
            for (int k2 = c2 + 1; k2 < CB; ++k2) row[k2] = fma(-t, colv[k2], row[k2]);
            row[c2] *= inv;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) D[r][c2] = c2 <= r ? row[c2] : 0.0;
        rdiag[r] = 1.0 / D[r][r];
        if (bad && tid == 0) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) {
            q.ctl[CTL_REASON] = 1;
            q.ctl[CTL_DONE] = 2;
        }
        return;
    }
    // the factor of the diagonal block back to the workspace
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c2 = e % CB;
        if (r < nc && c2 <= r) Aw[(j0 + r) * CHOL_NP + j0 + c2] = D[r][c2];
    }
    // rows below the block (up to and including the right-hand-side row M): L_i = A_i D^-T, a thread per row
    // (full 32-column runs, also for a short last panel: the workspace is defined everywhere, the block is identity-padded, and
    // what lands beyond column M of the right-hand-side row is never read -- conditional loads here cost 2.7 KB of scratch)
    const int i = j0 + nc + tid;
    if (i <= M) {
        double a[CB];
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) a[c2] = Aw[i * CHOL_NP + j0 + c2];
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) {
            // (fenced: left alone, the scheduler hoists all 528 LDS reads of the unrolled solve to the top and spills)
            __builtin_amdgcn_sched_barrier(0);
            double v = a[c2];
#pragma unroll
            for (int k2 = 0; k2 < c2; ++k2) v = fma(-a[k2], D[c2][k2], v);
            a[c2] = v * rdiag[c2];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < CB; ++c2) Aw[i * CHOL_NP + j0 + c2] = a[c2];
    }
}

// trailing update after the panel at column j0: tile (bi, bk) of 32 x 32, bk <= bi, rows / columns from j0 + CB on
__global__ void __launch_bounds__(256)
k_chol_update(AdaptArgs q, double* __restrict__ Aw, int j0) {
    __shared__ double Li[CB][CB + 1], Lk[CB][CB + 1];
    if (q.ctl[CTL_DONE] != 0) return;
    const int M = q.m - 1, tid = threadIdx.x;
    // linear tile index -> (bi, bk), bk <= bi
    int bi = 0, rem = blockIdx.x;
    while (rem > bi) { rem -= bi + 1; ++bi; }
    const int bk = rem;
    const int r0 = j0 + CB + bi * CB, c0 = j0 + CB + bk * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c2 = e % CB;
        Li[r][c2] = (r0 + r <= M) ? Aw[(r0 + r) * CHOL_NP + j0 + c2] : 0.0;
        Lk[r][c2] = (c0 + r <= M) ? Aw[(c0 + r) * CHOL_NP + j0 + c2] : 0.0;
    }
    __syncthreads();
    const int r = tid >> 3, cg = (tid & 7) * 4;  // a thread: one row, four columns
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int k2 = 0; k2 < CB; ++k2) {
        const double l = Li[r][k2];
        s0 = fma(l, Lk[cg][k2], s0);
        s1 = fma(l, Lk[cg + 1][k2], s1);
        s2 = fma(l, Lk[cg + 2][k2], s2);
        s3 = fma(l, Lk[cg + 3][k2], s3);
    }
    const int gi = r0 + r;
    if (gi <= M) {
        const double sv[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gk = c0 + cg + e;
            if (gk <= gi && gk < M) Aw[gi * CHOL_NP + gk] -= sv[e];
        }
    }
}

__global__ void __launch_bounds__(256)
k_chol_finish(AdaptArgs q, const double* __restrict__ Aw) {
    __shared__ double y[CHOL_NP], xs[CHOL_NP + 1], Lbb[CB][CB + 1];
    __shared__ double s_f[256], s_ps[256], s_nk[256], s_ln[256], s_a0[256];
    __shared__ int smp[CHOL_NP + 1], pos[256];
    if (q.ctl[CTL_DONE] != 0) return;
    const int M = q.m - 1, tid = threadIdx.x;
    for (int k = tid; k < q.Kp; k += 256) {
        s_f[k] = k < q.K ? q.f[k] : 0.0;
        s_ps[k] = q.psum[k];
        s_nk[k] = q.Nk[k];
        s_ln[k] = q.lnNk[k];
        s_a0[k] = q.pmode ? q.a0[k] : 0.0;
        pos[k] = 0;
    }
    for (int i = tid; i < q.m; i += 256) smp[i] = q.sampled[i];
    for (int i = tid; i < CHOL_NP; i += 256) y[i] = i < M ? Aw[M * CHOL_NP + i] : 0.0;  // y = L^-1 b
    __syncthreads();
    for (int i = tid; i < q.m; i += 256) pos[smp[i]] = i;
    // L^T x = y from the last block of 32 upwards (right-looking: a solved block is folded into every unknown above it at once)
    const int nblk = (M + CB - 1) / CB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int c0 = b * CB, nc = M - c0 < CB ? M - c0 : CB;
        for (int e = tid; e < CB * CB; e += 256) {  // the block's own triangle (identity-padded)
            const int r = e / CB, c2 = e % CB;
            Lbb[r][c2] = (r < nc && c2 <= r) ? Aw[(c0 + r) * CHOL_NP + c0 + c2] : (r == c2 ? 1.0 : 0.0);
        }
        __syncthreads();
        if (tid < 64) {  // one wave: lane c holds unknown c0 + c
            const int c = tid & (CB - 1);
            double yv = y[c0 + c];  // (zero beyond M)
            const double rd = 1.0 / Lbb[c][c];
#pragma unroll
            for (int c2 = CB - 1; c2 >= 0; --c2) {
                const double x = __shfl(yv, c2) * __shfl(rd, c2);
                if (c == c2) yv = x;
                if (c < c2) yv = fma(-Lbb[c2][c], x, yv);
            }
            if (tid < CB) y[c0 + c] = yv;
        }
        __syncthreads();
        {
            const int c = tid;  // (c0 <= 224: one thread per remaining unknown)
            if (c < c0) {
                double acc = 0.0;
#pragma unroll 8
                for (int r = 0; r < CB; ++r)
                    if (r < nc) acc = fma(Aw[(c0 + r) * CHOL_NP + c], y[c0 + r], acc);
                y[c] -= acc;
            }
        }
        __syncthreads();
    }
    if (tid == 0) xs[0] = 0.0;
    for (int i = tid; i < M; i += 256) xs[i + 1] = y[i];
    __syncthreads();
    newton_tail<256>(q, xs, false, s_f, s_ps, s_nk, s_ln, s_a0, smp, pos, tid);
}

// Choice between the candidates and convergence test, one workgroup of 256 threads (one state per thread).  The two
// gradient norms are fixed-order tree sums (deterministic; the host loop adds the same terms serially, so a round-off
// tie between the candidates may fall differently there), the convergence measures are maxima.
__device__ __forceinline__ void select_body(const AdaptArgs& q) {  // (256 threads; the pointers of q may be LDS or global)
    __shared__ double red[4];
    int* ctl = q.ctl;
    if (ctl[CTL_DONE] != 0) return;
    const int tid = threadIdx.x, Kp = q.Kp;
    const double tol = q.prm[1];
    const int min_sc = (int)q.prm[2];
    const bool check = q.prm[3] != 0.0;
    const int first = q.sampled[0];
    const bool in = tid < Kp;
    const double nk = in ? q.Nk[tid] : 0.0;
    const bool sampled = in && tid < q.K && nk > 0.0;
    // the sweeps accumulate UNSCALED per-state sums: times the candidate's per-state constant = its psum
    // (classic: ratio c_k of the second candidate only; P mode: exp(a - a0) of both, kept in aden).  Index 0 = the
    // self-consistent candidate, 1 = Newton-Raphson; the fused sweep may have been handed them in swapped order (CTL_SPEC).
    const bool swap = q.fused && ctl[CTL_SPEC] == 0;
    const int o_sci = swap ? Kp : 0, o_nr = swap ? 0 : Kp;
    const double m0 = (in && q.pmode) ? q.aden[o_sci + tid] : 1.0;
    const double m1 = in ? (q.pmode ? q.aden[o_nr + tid] : q.ratio[tid]) : 0.0;
    double raw0 = in ? q.lse_red[o_sci + tid] : 0.0, raw1 = in ? q.lse_red[o_nr + tid] : 0.0;
    if (q.fused && Kp == 128 && FUSED_PSUM1_FROM_GRAM_NB <= 8) {  // (k_fused<8> only: narrower panels and k_fused_quad accumulate both rows)
        // the fused sweep of a full panel left the unscaled sums of its SECOND multiplier row c to be taken from the Gram matrix
        // it accumulated for that candidate: sum_n w_n P_kn / s_n = sum_j c_j G'_kj (rows of p sum to one)
        __shared__ double s_c[128], s_half[128];
        if (tid < Kp) s_c[tid] = q.aden[Kp + tid];
        __syncthreads();
        // (two threads per state, half of the columns each: the loads are what this costs)
        const int k = tid & 127, h = tid >> 7, nb = Kp / 16;
        double acc = 0.0;
#pragma unroll 8
        for (int j = h * 64; j < h * 64 + 64; ++j) acc = fma(s_c[j], gram_elem(q.gram_red, nb, k, j), acc);
        if (h == 1) s_half[k] = acc;
        __syncthreads();
        if (h == 0) acc += s_half[k];
        if (swap) raw0 = acc; else raw1 = acc;
    }
    const double ps0 = raw0 * m0;
    const double ps1 = raw1 * m1;
    const double fo = in ? q.f[tid] : 0.0, fs = in ? q.cand[tid] : 0.0, fn = in ? q.cand[Kp + tid] : 0.0;
    const double lnk = in ? q.lnNk[tid] : 0.0;
    const double ga = sampled ? ps0 - nk : 0.0, gb = sampled ? ps1 - nk : 0.0;
    const double gs = block256_sum(ga * ga, red);
    const double gn = block256_sum(gb * gb, red);
    // :607 (every thread holds the same sums); a NaN Newton gradient loses against a finite self-consistent one (host loop)
    const int ch = (gs < gn || (gn != gn && gs == gs) || ctl[CTL_SCI] < min_sc) ? 0 : 1;
    const double fnew = ch == 0 ? fs : fn;
    if (in) {
        q.f[tid] = fnew;
        q.psum[tid] = ch == 0 ? ps0 : ps1;
        q.anum[tid] = sampled ? fnew + lnk : -INFINITY;
        if (q.pmode) q.ccur[tid] = ch == 0 ? m0 : m1;
    }
    // convergence measures over the sampled states except the gauge state (:627-633); NaN: see the host loop
    const bool counts = sampled && tid != first;
    const double small = tol < 1e-8 ? tol : 1e-8;
    const double div = fabs(fnew) < small ? 1.0 : fabs(fnew);
    const double d1 = counts ? fabs(fnew - fo) / div : 0.0;
    const double d2 = counts ? fabs(fs - fn) / div : 0.0;
    const double nan_seen = block256_max((d1 != d1) ? 1.0 : 0.0, red);
    double max_delta = block256_max(d1 != d1 ? 0.0 : d1, red);
    const double max_diff = block256_max(d2 != d2 ? 0.0 : d2, red);
    if (nan_seen > 0.0) max_delta = NAN;
    // Fused sweep: the Gram matrix of the Newton-Raphson candidate is already there.  It serves the next iteration when
    // that candidate was accepted -- or when the two candidates coincide to 1e-10 (at the fixed point the choice is
    // round-off noise; the Hessian of one is the Hessian of the other far below any tolerance it is used at).
    const int spec = swap ? 0 : 1;  // the candidate the sweep speculated on
    const bool reuse = q.fused && (ch == spec || max_diff <= 1e-10);
    if (q.fused && in) q.cgram[tid] = (reuse ? spec : ch) == 0 ? m0 : m1;
    if (tid == 0) {
        const int it = ctl[CTL_ITER];
        if (it < q.hist_cap) {
            q.hist[4 * (int64_t)it + 0] = ch;
            q.hist[4 * (int64_t)it + 1] = sqrt(gs);
            q.hist[4 * (int64_t)it + 2] = sqrt(gn);
            q.hist[4 * (int64_t)it + 3] = max_delta;
        }
        q.state[0] = max_delta;
        const bool stop = check && (max_delta != max_delta || (max_delta < tol && max_diff < sqrt(tol)));  // :636
        ctl[CTL_ITER] = it + 1;
        if (ch == 0) ctl[CTL_SCI] += 1; else ctl[CTL_NR] += 1;
        // (the sweep wrote the reciprocals of its first multiplier row to slot + 1, of its second to slot + 2)
        ctl[CTL_SLOT] = (ctl[CTL_SLOT] + ((ch == 0) != swap ? 1 : 2)) % 3;
        if (q.fused) {
            ctl[CTL_NEEDGRAM] = reuse ? 0 : 1;
            if (!reuse) ctl[CTL_GRAMSWEEPS] += 1;
        }
        if (stop)
            ctl[CTL_DONE] = 1;
        else if (q.fused && !reuse)
            ctl[CTL_DONE] = 3;  // pause: the host enqueues the Gram sweep of the accepted candidate (same flags on every rank)
    }
}
__global__ void __launch_bounds__(256)
k_select(AdaptArgs q) {
    select_body(q);
}
// Fused loop: the selection of iteration i and the Newton solve of iteration i + 1 in ONE launch (a kernel boundary costs ~5 us;
// at the sizes pymbar is mostly used at that is a tenth of an iteration).  A stop or pause flag raised by the selection makes the
// solve return at once.
template <int R>
__global__ void __launch_bounds__(256)
k_select_newton(AdaptArgs q) {
    select_body(q);
    __syncthreads();
    newton_body<16, R>(q);
}

__global__ void __launch_bounds__(256)
k_loop_reduce(LoopSrc src, int64_t count, int op, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double v = src.p[0][i];
    for (int r = 1; r < src.n; ++r) v = op == 0 ? v + src.p[r][i] : fmax(v, src.p[r][i]);
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------
// Small problems: the whole device-resident loop in ONE launch.
// At the sizes pymbar is mostly run at (tens of states, 1e4 - 1e6 samples: a sweep of 5-20 us) an iteration of five launches
// is the fixed cost of its launches -- 3-4 us each before any work -- plus a thousand partial records written and re-read.
// Here a persistent grid of ONE workgroup per compute unit (at most) loops over the iterations itself:
//   [K x K Newton solve + candidates: every workgroup, redundantly, on its OWN copy of the solver state in LDS -- identical
//    inputs, identical bits, no broadcast]  [fused sweep over the workgroup's tiles: both candidates' normalisers and per-state
//    sums + the Gram matrix of the second one on the matrix cores]  [the four waves' records folded through LDS: ONE record per
//    workgroup]  [grid barrier]  [the records reduced in a fixed order, four threads per entry spread over the grid]
//   [grid barrier]  [selection + convergence test: every workgroup, redundantly]
// Records and reduced values are double-buffered by the parity of the iteration, which is what lets two barriers per iteration
// suffice.  The loop stops like the multi-launch one: converged, handed back, paused for a separate Gram sweep (the host takes
// over from the state workgroup 0 writes back), or out of iterations.  A barrier that is not met within ~0.5 s raises a flag
// and every workgroup leaves (the host then reports an error instead of hanging the device).
// The sweep is the plain one (compiler-scheduled matrix instructions, NB <= 5: at most 15 blocks) -- at these sizes it is not
// what the iteration costs.
// ---------------------------------------------------------------------------------------------
// Grid barrier of the persistent kernel.  What the workgroups exchange through device memory (records, reduced values) is
// written with agent-scope stores (write-through: visible to the other XCDs' L2 without a cache write-back) and read after an
// agent-scope acquire (an invalidate) -- a full __threadfence() here writes back the whole L2 per workgroup and cost 20 us.
__device__ __forceinline__ void store_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target) {
    __syncthreads();  // (every thread's stores have been waited for)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21) || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (invalidate: what the other workgroups wrote is read from memory)
    return __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

__host__ __device__ constexpr int small_ring_depth(int nb) { return nb == 1 ? 8 : (nb == 2 ? 6 : (nb == 3 ? 5 : 3)); }
template <int NB>
__global__ void __launch_bounds__(256, 1)
k_solve_small(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16, NBLK = NB * (NB + 1) / 2;
    constexpr int REC_L = 2 * ROWS, REC_G = NBLK * 256, E = REC_L + REC_G;
    constexpr int NSTG = ROWS / 8 + 2;                // LDS-DMA instructions per tile: the rows + the multiplicities + their roots
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 2 * TS * 8;  // + the tile's 16 multiplicities and their roots
    constexpr int DEPTH = small_ring_depth(NB);
    constexpr int R = NB <= 4 ? 4 : 8;                // register tile of the Newton solve (16 x 16 threads): 63 / 127 unknowns
    __shared__ double s_f[ROWS], s_psum[ROWS], s_cand[2 * ROWS], s_ratio[ROWS], s_aden[2 * ROWS], s_anum[ROWS], s_ccur[ROWS],
        s_cgram[ROWS], s_state[2];
    __shared__ int s_ctl[CTL_WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = lane & 15, ns = lane >> 4;
    const int G = gridDim.x;
    const AdaptArgs& qg = a.q;
    for (int k = tid; k < ROWS; k += 256) {
        s_f[k] = qg.f[k];
        s_psum[k] = qg.psum[k];
        s_anum[k] = qg.anum[k];
        s_ccur[k] = qg.ccur[k];
        s_cgram[k] = qg.cgram[k];
        s_ratio[k] = 1.0;
        s_cand[k] = s_cand[ROWS + k] = 0.0;
        s_aden[k] = s_aden[ROWS + k] = 0.0;
    }
    if (tid < CTL_WORDS) s_ctl[tid] = qg.ctl[tid];
    if (tid == 0) s_state[0] = qg.state[0];
    __syncthreads();
    AdaptArgs ql = a.q;  // this workgroup's view: the mutable state lives in its LDS
    ql.f = s_f;
    ql.psum = s_psum;
    ql.cand = s_cand;
    ql.ratio = s_ratio;
    ql.aden = s_aden;
    ql.anum = s_anum;
    ql.ccur = s_ccur;
    ql.cgram = s_cgram;
    ql.ctl = s_ctl;
    ql.state = s_state;
    if (blockIdx.x != 0) ql.hist_cap = 0;  // (the history rows are written once, by workgroup 0)

    char* buf = smem + wave * (DEPTH * TILE_BYTES);
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t W = (int64_t)G * 4;
    RowIdentity rows{0};
    const StageOffsets so = make_stage_offsets(a.ld, lane);
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;
    auto stage = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, true, 0, 1>(a.P, a.ld, tile * TS, dst, lane, so, rows);
        stage_vec16<true>(a.cw, tile * TS, dst + U_BYTES, lane);
        stage_vec16<true>(a.wsq, tile * TS, dst + U_BYTES + TS * 8, lane);
    };

    unsigned nbar = 0;
    int par = 0;
    bool ok = true;
#ifdef MBAR_EXPERIMENT_SMALL_TIMING  // instrumented build: where an iteration of the persistent loop spends its time (workgroup 0)
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    int nit = 0;
#define SMALL_TICK(i) do { const long long tn_ = wall_clock64(); tph[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define SMALL_TICK(i) do { } while (0)
#endif
    for (int it = 0; it < a.max_iters && ok; ++it) {
        if (s_ctl[CTL_DONE] != 0) break;
        SMALL_TICK(5);
        newton_body<16, R>(ql);
        __syncthreads();
        SMALL_TICK(0);
        if (s_ctl[CTL_DONE] != 0) break;  // (handed back: Newton system not positive definite, candidate out of the window)
        // ---- fused sweep over this wave's tiles
        double c0[NB], c1[NB], acc0[NB], acc1[NB];
        v4d Gm[NBLK];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            c0[I] = s_aden[16 * I + ks];
            c1[I] = s_aden[ROWS + 16 * I + ks];
            acc0[I] = acc1[I] = 0.0;
        }
#pragma unroll
        for (int b = 0; b < NBLK; ++b) Gm[b] = v4d{0.0, 0.0, 0.0, 0.0};
        __syncthreads();  // (the fold region of the previous iteration is the tile buffers' memory)
        // Ring of DEPTH tile buffers per wave: a wave has a handful of tiles (a few hundred KB of matrix per compute unit), so
        // what a tile costs is the latency of its LDS-DMA -- all of a wave's first DEPTH tiles are requested at once.  No store in
        // this loop (vmcnt counts the LDS-DMA pieces only, in order): the reciprocals 1 / s_n are not written -- only a separate
        // Gram sweep after a pause reads them, and the host has them recomputed then.
        const int64_t ntw = gw < a.ntiles ? (a.ntiles - gw + W - 1) / W : 0;  // tiles of this wave
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (j < ntw) stage(gw + j * W, buf + j * TILE_BYTES);
        int cur = 0;
        for (int64_t j = 0; j < ntw; ++j) {
            char* cbuf = buf + cur * TILE_BYTES;
            if (j + DEPTH - 1 < ntw)
                wait_vm<(DEPTH - 1) * NSTG>();  // exactly DEPTH - 1 younger tiles are in flight
            else
                wait_vm<0>();
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                double x[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I) x[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g]);
                const double w = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
                const double sw = *reinterpret_cast<const double*>(cbuf + U_BYTES + TS * 8 + (4 * g + ns) * 8);
                double d0 = dot_sum<NB>(x, c0), d1 = dot_sum<NB>(x, c1);
                row16_sum2(d0, d1);
                // (a padded sample has an all-zero column: keep its reciprocal finite, its multiplicity is 0)
                const double r0 = recip_fast(fmax(d0, 1e-300)), r1 = recip_fast(fmax(d1, 1e-300));
                const double q0 = w * r0, q1 = w * r1, rin = r1 * sw;
                double p[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    acc0[I] = fma(x[I], q0, acc0[I]);
                    acc1[I] = fma(x[I], q1, acc1[I]);
                    p[I] = x[I] * rin;
                }
                int b = 0;
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int J = I; J < NB; ++J, ++b) Gm[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(p[I], p[J], Gm[b], 0, 0, 0);
            }
            // (every LDS read of this buffer has been consumed by the arithmetic above: it takes tile j + DEPTH)
            if (j + DEPTH < ntw) stage(gw + (j + DEPTH) * W, cbuf);
            cur = cur + 1 == DEPTH ? 0 : cur + 1;
        }
        // ---- fold the four waves' records through LDS: one record per workgroup
        __syncthreads();  // (all tile buffers are dead)
        SMALL_TICK(1);
        double* frec = reinterpret_cast<double*>(smem) + (size_t)wave * E;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            double v0 = acc0[I], v1 = acc1[I];
            v0 += __shfl_xor(v0, 16);
            v0 += __shfl_xor(v0, 32);
            v1 += __shfl_xor(v1, 16);
            v1 += __shfl_xor(v1, 32);
            if (lane < 16) {
                frec[16 * I + lane] = v0;
                frec[ROWS + 16 * I + lane] = v1;
            }
        }
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) frec[REC_L + (b * 4 + r) * 64 + lane] = Gm[b][r];
        __syncthreads();
        {
            const double* f0 = reinterpret_cast<const double*>(smem);
            double* out = a.rec + ((size_t)par * G + blockIdx.x) * E;
            for (int e = tid; e < E; e += 256) store_agent(out + e, (f0[e] + f0[E + e]) + (f0[2 * E + e] + f0[3 * E + e]));
        }
        nbar += 1;
        ok = grid_barrier(a.bar, nbar * (unsigned)G);
        SMALL_TICK(2);
        if (!ok) break;
        // ---- reduction over the workgroups, fixed order: sixteen threads per entry, each with its records' loads all in flight
        {
            const double* rb = a.rec + (size_t)par * G * E;
            double* red = a.red + (size_t)par * E;
            for (int64_t e16 = (int64_t)blockIdx.x * 256 + tid; e16 < (int64_t)16 * E; e16 += (int64_t)G * 256) {
                const int e = (int)(e16 >> 4), part = (int)(e16 & 15);
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = part + 16 * i;
                    v[i] = r < G ? rb[(size_t)r * E + e] : 0.0;
                }
                double sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                sum += ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
                sum += __shfl_xor(sum, 1);
                sum += __shfl_xor(sum, 2);
                sum += __shfl_xor(sum, 4);
                sum += __shfl_xor(sum, 8);
                if (part == 0) store_agent(red + e, sum);
            }
        }
        nbar += 1;
        ok = grid_barrier(a.bar, nbar * (unsigned)G);
        SMALL_TICK(3);
        if (!ok) break;
        ql.lse_red = a.red + (size_t)par * E;
        ql.gram_red = a.red + (size_t)par * E + REC_L;
        select_body(ql);
        __syncthreads();
        SMALL_TICK(4);
        par ^= 1;
#ifdef MBAR_EXPERIMENT_SMALL_TIMING
        ++nit;
#endif
    }
#ifdef MBAR_EXPERIMENT_SMALL_TIMING
    if (blockIdx.x == 0 && tid == 0 && nit > 0)
        printf("DBG small: %d iterations; per iteration (x10 ns): newton %lld, sweep %lld, fold+barrier1 %lld, reduce+barrier2 %lld, select %lld, top %lld\n",
               nit, tph[0] / nit, tph[1] / nit, tph[2] / nit, tph[3] / nit, tph[4] / nit, tph[5] / nit);
#endif
    // ---- the state back to device memory (workgroup 0; the others hold the same bits)
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int k = tid; k < ROWS; k += 256) {
            qg.f[k] = s_f[k];
            qg.psum[k] = s_psum[k];
            qg.cand[k] = s_cand[k];
            qg.cand[ROWS + k] = s_cand[ROWS + k];
            qg.ratio[k] = s_ratio[k];
            qg.aden[k] = s_aden[k];
            qg.aden[ROWS + k] = s_aden[ROWS + k];
            qg.anum[k] = s_anum[k];
            qg.ccur[k] = s_ccur[k];
            qg.cgram[k] = s_cgram[k];
        }
        if (tid < CTL_WORDS) qg.ctl[tid] = s_ctl[tid];
        if (tid == 0) qg.state[0] = s_state[0];
    }
}

// Fused loop, resumed after a pause (CTL_DONE = 3): the host enqueues this in front of the accepted candidate's Gram sweep.
__global__ void k_ctl_resume(int* ctl) {
    if (threadIdx.x == 0 && ctl[CTL_DONE] == 3) {
        ctl[CTL_DONE] = 0;
        ctl[CTL_NEEDGRAM] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static int blocks_per_cu_for(size_t lds_bytes) {
    int b = (int)((160 * 1024) / lds_bytes);
    if (b < 1) b = 1;
    if (b > 4) b = 4;
    return b;
}

LaunchGeom lse_geometry(int nb, int nf, int num_cu, int64_t ntiles, int64_t grid_override, int variant) {
    LaunchGeom g;
    const size_t tile = (size_t)nb * 16 * TS * 8 + TS * 8;  // u tile + its 16 sample weights
    const bool small_ok = (variant & 0x10) != 0;  // set by the caller when the context qualifies (pitch, staging)
    const bool wide_ok = (variant & 0x20) != 0;   // likewise for the single-buffer wide-panel kernel
    variant &= 0xf;
    g.variant = (nb >= 6 && variant == 0 && !(nb > 8 && nf == 2)) ? 0 : 1;  // (wide two-candidate pairs would spill)
    if (small_ok && nf == 1 && nb <= 2) {  // few states (a third block of 16 would spill the per-lane state arrays): one sample per lane, 64-sample tiles, 8 waves x 1 buffer
        g.variant = 4;
        g.waves = 8;
        const size_t tile64 = (size_t)nb * 16 * TSS * 8 + TSS * 8;
        g.lds_bytes = (size_t)g.waves * tile64 + EXP_TABLE_BYTES;
        const int64_t nt64 = (ntiles * TS + TSS - 1) / TSS;
        int64_t want = (nt64 + g.waves - 1) / g.waves;
        int64_t cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
        if (grid_override > 0) cap = grid_override;
        if (want < 1) want = 1;
        g.blocks = (int)(want < cap ? want : cap);
        g.nwaves = g.blocks;  // partial records: this kernel folds its 8 waves and writes one per workgroup
        g.psum_records = g.nwaves;
        return g;
    }
    if (wide_ok && nb > 8) {  // 129..256 states: one tile buffer per wave, four waves per CU
        g.variant = 5;
        g.waves = 4;
        g.lds_bytes = (size_t)g.waves * tile + EXP_TABLE_BYTES;
        int64_t want = (ntiles + g.waves - 1) / g.waves;
        int64_t cap5 = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
        if (grid_override > 0) cap5 = grid_override;
        if (want < 1) want = 1;
        g.blocks = (int)(want < cap5 ? want : cap5);
        g.nwaves = g.blocks * g.waves;
        g.psum_records = g.nwaves;
        return g;
    }
    if (variant >= 2 && nb >= 5 && nb <= 8) g.variant = variant == 2 ? 2 : 3;
    int64_t cap;
    if (g.variant >= 2) {  // early refill: 8 waves x 1 tile buffer (2) or 4 waves x 2 buffers (3)
        g.waves = g.variant == 2 ? 8 : 4;
        g.lds_bytes = (size_t)8 * tile + EXP_TABLE_BYTES;
        int64_t want = (ntiles + g.waves - 1) / g.waves;
        cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
        if (grid_override > 0) cap = grid_override;
        if (want < 1) want = 1;
        g.blocks = (int)(want < cap ? want : cap);
        g.nwaves = g.blocks * g.waves;
    } else if (g.variant == 0) {  // paired: STREAMS tile streams x 2 waves
        const int streams = nb <= 8 ? 4 : 2;
        g.waves = 2 * streams;
        g.lds_bytes = (size_t)streams * 2 * tile + EXP_TABLE_BYTES;
        int64_t want = (ntiles + streams - 1) / streams;
        cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
        if (grid_override > 0) cap = grid_override;
        if (want < 1) want = 1;
        g.blocks = (int)(want < cap ? want : cap);
        g.nwaves = g.blocks * streams * 2;  // one partial record per wave
    } else {
        g.waves = lse_waves(nb);
        g.lds_bytes = (size_t)g.waves * 2 * tile + EXP_TABLE_BYTES;
        int64_t want = (ntiles + g.waves - 1) / g.waves;
        cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
        if (grid_override > 0) cap = grid_override;
        if (want < 1) want = 1;
        g.blocks = (int)(want < cap ? want : cap);
        g.nwaves = g.blocks * g.waves;
    }
    g.psum_records = g.nwaves;
    return g;
}

LaunchGeom gram_geometry(int tile_rows, bool diag, int num_cu, int64_t ntiles, int64_t grid_override, int nb8_variant) {
    LaunchGeom g;
    g.waves = 4;
    g.variant = -1;
    const size_t tile = (size_t)tile_rows * TS * 8 + TS * 8;  // u tile + its 16 logden values
    const bool one_buffer = tile_rows > 128 || (diag && tile_rows <= 80);  // must match NBUF in k_gram
    g.lds_bytes = (size_t)4 * (one_buffer ? 1 : 2) * tile + EXP_TABLE_BYTES;
    if (diag && tile_rows == 128 && nb8_variant == 2) {
        g.variant = 2;
    } else if (diag && tile_rows == 128) {
        g.variant = nb8_variant == 1 ? 1 : 0;
        g.waves = 8;
        if (g.variant == 0) g.lds_bytes = (size_t)4 * (tile + (size_t)GROUPS * 8 * 64 * 8) + EXP_TABLE_BYTES;  // single u buffer + operand buffer
    }
    int64_t want = (ntiles + 3) / 4;
    // Wide panels: the accumulators own the register file, one workgroup (one wave per SIMD) per CU.  Narrow diagonal
    // panels need few registers and little LDS, so several workgroups share a CU (register occupancy of k_gram<NB,NB>:
    // 8 / 6 / 4 / 3 / 2 waves per SIMD for NB = 1..5).
    int64_t cap = num_cu;
    if (diag && tile_rows <= 80) {
        static const int occ[6] = {1, 8, 6, 4, 3, 2};
        const int by_lds = blocks_per_cu_for(g.lds_bytes);
        const int by_reg = occ[tile_rows / 16];
        cap = (int64_t)num_cu * (by_lds < by_reg ? by_lds : by_reg);
        // (... once a wave has ~32 tiles to work on: every wave writes a partial record -- see fused_geometry)
        int64_t by_work = (ntiles + 127) / 128;
        if (by_work < num_cu) by_work = num_cu;
        if (cap > by_work) cap = by_work;
    }
    if (grid_override > 0) cap = grid_override;
    if (want < 1) want = 1;
    g.blocks = (int)(want < cap ? want : cap);
    g.nwaves = g.blocks * 4;  // per wave (k_gram) or per tile stream (paired kernels)
    g.psum_records = g.variant == 0 ? 2 * g.nwaves : g.nwaves;
    return g;
}

static bool stage_offsets_wide(int64_t ld) { return (uint64_t)ld * 56u + 128u >= (1ull << 32); }

template <typename Kern>
static hipError_t launch_kernel_lse(Kern kern, hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                    const double* aden, const double* cw, double* l0, double* l1, const double* dn,
                                    double* psum_part, double* obj_part, const LoopCtl& lc) {
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TS - 1) / TS;
    if (lc.ev_start && lc.ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, u, ld, N,
                              ntiles, aden, cw, l0, l1, dn, psum_part, obj_part, lc.ctl, lc.slot_stride);
    else
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, l0,
                           l1, dn, psum_part, obj_part, lc.ctl, lc.slot_stride);
    return hipGetLastError();
}

template <int NB, int NF, bool DMA>
static hipError_t launch_lse_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                               const double* aden, const double* cw, double* l0, double* l1, const double* dn,
                               double* psum_part, double* obj_part, const LoopCtl& lc) {
    if (stage_offsets_wide(ld))
        return launch_kernel_lse(k_lse<NB, NF, DMA, true>, s, g, u, ld, N, aden, cw, l0, l1, dn, psum_part, obj_part, lc);
    return launch_kernel_lse(k_lse<NB, NF, DMA, false>, s, g, u, ld, N, aden, cw, l0, l1, dn, psum_part, obj_part, lc);
}

template <int NB, int NF, bool DMA>
static hipError_t launch_lse_pair_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                    const double* aden, const double* cw, double* l0, double* l1, const double* dn,
                                    double* psum_part, double* obj_part) {
    auto kern = k_lse_pair<NB, NF, DMA>;
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, l0,
                       l1, dn, psum_part, obj_part);
    return hipGetLastError();
}

template <int NB, int NF, int NBUF>
static hipError_t launch_lse_early_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                  const double* aden, const double* cw, double* l0, double* l1, const double* dn,
                                  double* psum_part, double* obj_part) {
    auto kern = k_lse_early<NB, NF, NBUF>;
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, l0,
                       l1, dn, psum_part, obj_part);
    return hipGetLastError();
}

template <int NB>
static hipError_t launch_lse_nb(hipStream_t s, int nf, bool dma, const LaunchGeom& g, const double* u,
                                int64_t ld, int64_t N, const double* aden, const double* cw, double* l0, double* l1,
                                const double* dn, double* pp, double* op, const LoopCtl& lc) {
    if (lc.ctl && g.variant != 1) return hipErrorInvalidValue;  // only the default sweep honours the control words
    if constexpr (NB >= 5 && NB <= 8) {
        if (g.variant >= 2) {
            if (!dma) return hipErrorInvalidValue;  // (the caller selects these variants only with LDS-DMA staging)
            if (g.variant == 2)
                return nf == 1 ? launch_lse_early_t<NB, 1, 1>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op)
                               : launch_lse_early_t<NB, 2, 1>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op);
            return nf == 1 ? launch_lse_early_t<NB, 1, 2>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op)
                           : launch_lse_early_t<NB, 2, 2>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op);
        }
    }
    if constexpr (NB >= 6) {
        if (g.variant == 0) {
            if (nf == 1)
                return dma ? launch_lse_pair_t<NB, 1, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op)
                           : launch_lse_pair_t<NB, 1, false>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op);
            return dma ? launch_lse_pair_t<NB, 2, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op)
                       : launch_lse_pair_t<NB, 2, false>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op);
        }
    }
    if (nf == 1)
        return dma ? launch_lse_t<NB, 1, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc)
                   : launch_lse_t<NB, 1, false>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
    return dma ? launch_lse_t<NB, 2, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc)
               : launch_lse_t<NB, 2, false>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
}

template <int NB>
static hipError_t launch_lse_small_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                     const double* aden, const double* cw, double* l0, const double* dn,
                                     double* psum_part, double* obj_part) {
    auto kern = k_lse_small<NB>;
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TSS - 1) / TSS;
    hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, l0, dn,
                       psum_part, obj_part);
    return hipGetLastError();
}

hipError_t launch_lse(hipStream_t s, int nb, int nf, bool dma, const LaunchGeom& g, const double* u,
                      int64_t ld, int64_t N, const double* aden, const double* cw, double* l0, double* l1,
                      const double* dn, double* pp, double* op, const LoopCtl& lc) {
    if (lc.ctl && g.variant == 4) return hipErrorInvalidValue;
    if (g.variant == 5) {  // (geometry chose the single-buffer wide-panel kernel: nb = 12 or 16, LDS-DMA staging)
        if (!dma) return hipErrorInvalidValue;
        auto go = [&](auto kern) -> hipError_t {
            if (g.lds_bytes > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
                if (e != hipSuccess) return e;
            }
            const int64_t ntiles = (N + TS - 1) / TS;
            if (lc.ev_start && lc.ev_stop)
                hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, u, ld, N,
                                      ntiles, aden, cw, l0, l1, dn, pp, op, lc.ctl, lc.slot_stride);
            else
                hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, l0, l1,
                                   dn, pp, op, lc.ctl, lc.slot_stride);
            return hipGetLastError();
        };
        if (nb == 12) return nf == 1 ? go(k_lse_wide<12, 1>) : go(k_lse_wide<12, 2>);
        if (nb == 16) return nf == 1 ? go(k_lse_wide<16, 1>) : go(k_lse_wide<16, 2>);
        return hipErrorInvalidValue;
    }
    if (g.variant == 4) {  // (geometry chose the few-state kernel: nf == 1, nb <= 3, LDS-DMA staging, pitch % 64 == 0)
        if (nf != 1 || !dma || (ld % TSS) != 0) return hipErrorInvalidValue;
        switch (nb) {
            case 1: return launch_lse_small_t<1>(s, g, u, ld, N, aden, cw, l0, dn, pp, op);
            case 2: return launch_lse_small_t<2>(s, g, u, ld, N, aden, cw, l0, dn, pp, op);
            default: return hipErrorInvalidValue;
        }
    }
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_lse_nb<NB_>(s, nf, dma, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7)
        MBAR_CASE(8) MBAR_CASE(12) MBAR_CASE(16)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

template <int NBI, int NBJ, bool DIAG, bool DMA, bool CLAMP = true, bool PMODE = false>
static hipError_t launch_gram_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                const double* ai, const double* aj, const double* logden, int64_t ri,
                                int64_t rj, double* gp, double* pp, const LoopCtl& lc = LoopCtl()) {
    auto launch = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        if (lc.ev_start && lc.ev_stop)
            hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, u, ld,
                                  N, ntiles, ai, aj, logden, ri, rj, gp, pp, lc.ctl, lc.slot_stride, lc.cond_needgram ? 1 : 0);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, ai, aj,
                               logden, ri, rj, gp, pp, lc.ctl, lc.slot_stride, lc.cond_needgram ? 1 : 0);
        return hipGetLastError();
    };
    return stage_offsets_wide(ld) ? launch(k_gram<NBI, NBJ, DIAG, DMA, true, CLAMP, PMODE>)
                                  : launch(k_gram<NBI, NBJ, DIAG, DMA, false, CLAMP, PMODE>);
}

template <int NB, bool DMA>
static hipError_t launch_gram_pair_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                     const double* anum, const double* logden, int64_t row0, double* gp,
                                     double* pp) {
    auto kern = k_gram_pair<NB, DMA>;
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(512), g.lds_bytes, s, u, ld, N, ntiles, anum, logden, row0, gp, pp);
    return hipGetLastError();
}

template <int NB, bool DMA>
static hipError_t launch_gram_xchg_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                     const double* anum, const double* logden, int64_t row0, double* gp,
                                     double* pp) {
    auto kern = k_gram_xchg<NB, DMA>;
    if (g.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (N + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(512), g.lds_bytes, s, u, ld, N, ntiles, anum, logden, row0, gp, pp);
    return hipGetLastError();
}

hipError_t launch_gram_diag(hipStream_t s, int nb, bool dma, const LaunchGeom& g, const double* u, int64_t ld,
                            int64_t N, const double* anum, const double* logden, int64_t row0, double* gp,
                            double* pp, const LoopCtl& lc) {
    if (lc.pmode) {  // resident probability matrix: `u` = P, `logden` = the reciprocals 1 / s_n; LDS-DMA staging only
        if (!dma || (nb == 8 && g.variant != 2)) return hipErrorInvalidValue;
        switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_gram_t<NB_, NB_, true, true, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
            MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
            default: return hipErrorInvalidValue;
        }
    }
    if (nb == 8 && g.variant == 2) {  // one wave per SIMD owns all 36 blocks (accumulator classes pinned by hand)
        if (lc.unclamped && dma) return launch_gram_t<8, 8, true, true, false>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
        return dma ? launch_gram_t<8, 8, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc)
                   : launch_gram_t<8, 8, true, false>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
    }
    if (nb == 8) {  // paired-wave variants: g.nwaves counts tile streams (4 per workgroup)
        if (lc.ctl) return hipErrorInvalidValue;  // (only k_gram honours the control words)
        if (g.variant == 0)
            return dma ? launch_gram_xchg_t<8, true>(s, g, u, ld, N, anum, logden, row0, gp, pp)
                       : launch_gram_xchg_t<8, false>(s, g, u, ld, N, anum, logden, row0, gp, pp);
        return dma ? launch_gram_pair_t<8, true>(s, g, u, ld, N, anum, logden, row0, gp, pp)
                   : launch_gram_pair_t<8, false>(s, g, u, ld, N, anum, logden, row0, gp, pp);
    }
    switch (nb) {
#define MBAR_CASE(NB_)                                                                                  \
    case NB_:                                                                                           \
        return dma ? launch_gram_t<NB_, NB_, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc) \
                   : launch_gram_t<NB_, NB_, true, false>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

LaunchGeom gram_quad_geometry(int nbt, int num_cu, int64_t ntiles, int64_t grid_override) {
    LaunchGeom g;
    g.waves = 4;
    g.variant = 6;
    g.lds_bytes = (size_t)EXP_TABLE_BYTES + (size_t)2 * ((size_t)nbt * 16 * TS * 8 + 4 * 1024);
    int64_t cap = grid_override > 0 ? grid_override : num_cu;
    int64_t want = ntiles < 1 ? 1 : ntiles;
    g.blocks = (int)(want < cap ? want : cap);
    g.nwaves = g.blocks;  // one partial record per workgroup
    g.psum_records = g.nwaves;
    return g;
}
template <int NBT, bool PMODE, bool STOREP = false>
static hipError_t launch_gram_quad_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                     const double* anum, const double* logden, double* gp, const LoopCtl& lc, double* Pout = nullptr) {
    auto launch = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        if (lc.ev_start && lc.ev_stop)
            hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, u, ld, N, ntiles, anum,
                                  logden, gp, lc.ctl, lc.slot_stride, lc.cond_needgram ? 1 : 0, Pout);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, u, ld, N, ntiles, anum, logden, gp, lc.ctl,
                               lc.slot_stride, lc.cond_needgram ? 1 : 0, Pout);
        return hipGetLastError();
    };
    if (g.live_blocks > 0 && g.live_blocks <= NBT - 2)  // (the rows of the last two blocks are padding: the trimmed build)
        return stage_offsets_wide(ld) ? launch(k_gram_quad<NBT, true, PMODE, STOREP, NBT - 2>) : launch(k_gram_quad<NBT, false, PMODE, STOREP, NBT - 2>);
    return stage_offsets_wide(ld) ? launch(k_gram_quad<NBT, true, PMODE, STOREP>) : launch(k_gram_quad<NBT, false, PMODE, STOREP>);
}
hipError_t launch_gram_quad(hipStream_t s, int nbt, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                            const double* anum, const double* logden, double* gram_part, const LoopCtl& lc, double* Pout) {
    if (Pout) {  // classic operands, also written out as the probability matrix
        if (lc.pmode) return hipErrorInvalidValue;
        if (nbt == 12) return launch_gram_quad_t<12, false, true>(s, g, u, ld, N, anum, logden, gram_part, lc, Pout);
        if (nbt == 16) return launch_gram_quad_t<16, false, true>(s, g, u, ld, N, anum, logden, gram_part, lc, Pout);
        return hipErrorInvalidValue;
    }
    if (nbt == 12)
        return lc.pmode ? launch_gram_quad_t<12, true>(s, g, u, ld, N, anum, logden, gram_part, lc)
                        : launch_gram_quad_t<12, false>(s, g, u, ld, N, anum, logden, gram_part, lc);
    if (nbt == 16)
        return lc.pmode ? launch_gram_quad_t<16, true>(s, g, u, ld, N, anum, logden, gram_part, lc)
                        : launch_gram_quad_t<16, false>(s, g, u, ld, N, anum, logden, gram_part, lc);
    return hipErrorInvalidValue;
}

hipError_t launch_gram_off(hipStream_t s, int nbj, bool dma, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                           const double* ai, const double* aj, const double* logden, int64_t ri, int64_t rj,
                           double* gp) {
    if (nbj == 8)  // 64 x 128 rectangle: 32 blocks, pinned accumulator classes, one 192-row tile buffer per wave
        return dma ? launch_gram_t<4, 8, false, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr)
                   : launch_gram_t<4, 8, false, false>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
    return dma ? launch_gram_t<4, 4, false, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr)
               : launch_gram_t<4, 4, false, false>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
}

static int stream_blocks(int num_cu, int64_t N) {
    int64_t want = (N + 255) / 256;
    int64_t cap = (int64_t)num_cu * 8;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// 257 .. 512 states in one read: rows = allocated row count (a multiple of 64); returns the number of partial records.
hipError_t launch_lse_split(hipStream_t s, int num_cu, int nf, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                            const double* cw, double* logden, double* logden1, const double* dn, double* psum_part, double* obj_part,
                            int* blocks_out) {
    const int nbw = (int)((rows + 127) / 128);
    if (nbw < 1 || nbw > 4 || nf < 1 || nf > 2) return hipErrorInvalidValue;
    const int64_t ntiles = (N + TS - 1) / TS;
    const size_t tile = (size_t)nbw * 16 * TS * 8 + TS * 8;
    const size_t lds = EXP_TABLE_BYTES + (size_t)(1 + nf) * 8 * TS * 8 + (size_t)8 * 2 * tile;
    const int blocks = (int)(ntiles < num_cu ? (ntiles < 1 ? 1 : ntiles) : num_cu);
    *blocks_out = blocks;
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, s, u, ld, N, ntiles, rows, aden, cw, logden, logden1, dn, psum_part,
                           obj_part);
        return hipGetLastError();
    };
    if (nf == 1) {
        switch (nbw) {
            case 1: return go(k_lse_split<1, 1>);
            case 2: return go(k_lse_split<2, 1>);
            case 3: return go(k_lse_split<3, 1>);
            default: return go(k_lse_split<4, 1>);
        }
    }
    switch (nbw) {
        case 1: return go(k_lse_split<1, 2>);
        case 2: return go(k_lse_split<2, 2>);
        case 3: return go(k_lse_split<3, 2>);
        default: return go(k_lse_split<4, 2>);
    }
}

hipError_t launch_lse_generic(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t K,
                              const double* aden, const double* cw, double* logden, const double* dn,
                              double* obj_part, int* blocks_out) {
    const int blocks = stream_blocks(num_cu, N);
    *blocks_out = blocks;
    hipLaunchKernelGGL(k_lse_generic, dim3(blocks), dim3(256), 0, s, u, ld, N, K, aden, cw, logden, dn, obj_part);
    return hipGetLastError();
}

hipError_t launch_colsum_generic(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t K,
                                 const double* anum, const double* cw, const double* logden, double* psum_part,
                                 int* blocks_out) {
    int blocks = stream_blocks(num_cu, N);
    if (blocks > 512) blocks = 512;
    *blocks_out = blocks;
    hipLaunchKernelGGL(k_colsum_generic, dim3(blocks, (unsigned)((K + 7) / 8)), dim3(256), 0, s, u, ld, N, K,
                       anum, cw, logden, psum_part);
    return hipGetLastError();
}

hipError_t launch_reduce(hipStream_t s, const double* part, int64_t nparts, int64_t count, double* scratch,
                         double* out) {
    const unsigned gx = (unsigned)((count + 255) / 256);
    if (nparts <= 32) {
        hipLaunchKernelGGL(k_reduce, dim3(gx, 1), dim3(256), 0, s, part, nparts, count, nparts, out);
        return hipGetLastError();
    }
    const int64_t chunk = 32;
    const int64_t n1 = (nparts + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_reduce, dim3(gx, (unsigned)n1), dim3(256), 0, s, part, nparts, count, chunk, scratch);
    hipLaunchKernelGGL(k_reduce, dim3(gx, 1), dim3(256), 0, s, (const double*)scratch, n1, count, n1, out);
    return hipGetLastError();
}

// Sample ranges of the per-state log-space reduction: enough waves (state-row groups x ranges) to fill the chip a few
// times over, each with a long run of tiles.
static int64_t lognum_tiles_per_chunk(int64_t N, int64_t K) {
    const int64_t ntiles = (N + LN_TILE - 1) / LN_TILE;
    const int64_t nsb = (K + LN_ROWS - 1) / LN_ROWS;
    int64_t target = 16384 / nsb;
    if (target < 1) target = 1;
    int64_t tpc = (ntiles + target - 1) / target;
    return tpc < 1 ? 1 : tpc;
}
int64_t lognum_chunks(int64_t N, int64_t K) {
    const int64_t ntiles = (N + LN_TILE - 1) / LN_TILE;
    const int64_t tpc = lognum_tiles_per_chunk(N, K);
    return (ntiles + tpc - 1) / tpc;
}

hipError_t launch_lognum(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K, const double* anum,
                         const double* logden, double* pmax, double* psum, int64_t nchunks) {
    const int64_t nsb = (K + LN_ROWS - 1) / LN_ROWS;
    const int64_t waves = nsb * nchunks;
    hipLaunchKernelGGL(k_lognum, dim3((unsigned)((waves + 3) / 4)), dim3(256), EXP_TABLE_BYTES, s, u, ld, N, K, anum, logden, pmax,
                       psum, nchunks, lognum_tiles_per_chunk(N, K));
    return hipGetLastError();
}

hipError_t launch_lognum_merge(hipStream_t s, const double* pmax, const double* psum, int64_t K, int64_t nchunks,
                               double* out_max, double* out_sum) {
    hipLaunchKernelGGL(k_lognum_merge, dim3((unsigned)K), dim3(256), 0, s, pmax, psum, nchunks, out_max, out_sum);
    return hipGetLastError();
}

hipError_t launch_logw(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K, const double* f,
                       const double* logden, double* out, int64_t ld_out, bool exponentiate) {
    int64_t bx = (N + 255) / 256;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    if (exponentiate)
        hipLaunchKernelGGL(k_logw<true>, dim3((unsigned)bx, (unsigned)K), dim3(256), 0, s, u, ld, N, f, logden, out, ld_out);
    else
        hipLaunchKernelGGL(k_logw<false>, dim3((unsigned)bx, (unsigned)K), dim3(256), 0, s, u, ld, N, f, logden, out, ld_out);
    return hipGetLastError();
}

hipError_t launch_check_u(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K, int* flags) {
    int64_t bx = (N + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_check_u, dim3((unsigned)bx, (unsigned)K), dim3(256), 0, s, u, ld, N, flags);
    return hipGetLastError();
}

hipError_t launch_shift_logden(hipStream_t s, const double* logden, const double* cw, double alpha, int64_t N,
                               double* out, const LoopCtl& lc) {
    int64_t bx = (N + 255) / 256;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_shift_logden, dim3((unsigned)bx), dim3(256), 0, s, logden, cw, alpha, N, out, lc.ctl, lc.slot_stride);
    return hipGetLastError();
}

hipError_t launch_generate_harmonic(hipStream_t s, double* u, int64_t ld, int64_t N, int64_t K, uint64_t seed,
                                    const double* O_k, const double* K_k, const int64_t* cumN,
                                    int64_t n_global0) {
    int64_t bx = (N + 255) / 256;
    if (bx > 4096) bx = 4096;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_generate_harmonic, dim3((unsigned)bx), dim3(256), 0, s, u, ld, N, K, seed, O_k, K_k, cumN,
                       n_global0);
    return hipGetLastError();
}

// rows[i][n] = (label[n] == i) ? v[n] : +inf  for i < nrows: one "state" per histogram bin whose only samples are the
// bin's own (a +inf reduced potential is weight zero).  grid.y = row.
__global__ void __launch_bounds__(256)
k_fill_masked_rows(double* __restrict__ rows, int64_t ld, int64_t n, const double* __restrict__ v,
                   const int* __restrict__ label) {
    const int i = blockIdx.y;
    double* row = rows + (int64_t)i * ld;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        row[k] = label[k] == i ? v[k] : INFINITY;
}
hipError_t launch_fill_masked_rows(hipStream_t s, double* rows, int64_t ld, int64_t n, int64_t nrows, const double* v,
                                   const int* label) {
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(k_fill_masked_rows, dim3((unsigned)(want < 2048 ? (want < 1 ? 1 : want) : 2048), (unsigned)nrows), dim3(256),
                       0, s, rows, ld, n, v, label);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_row_sub(double* __restrict__ row, const double* __restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) row[i] -= v[i];
}
hipError_t launch_row_sub(hipStream_t s, double* row, const double* v, int64_t n) {
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(k_row_sub, dim3((unsigned)(want < 4096 ? (want < 1 ? 1 : want) : 4096)), dim3(256), 0, s, row, v, n);
    return hipGetLastError();
}

// dst[r][i] = src[r][i] - v[i] for r < nrows (rows `ld` apart; dst == src: in place)
__global__ void __launch_bounds__(256) k_rows_sub(double* __restrict__ dst, const double* __restrict__ src, int64_t ld, int64_t nrows,
                                                  const double* __restrict__ v, int64_t n) {
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            dst[r * ld + i] = src[r * ld + i] - v[i];
}
// dst[r][i] = src[r][i] - dst[r][i]  (the observable rows arrive as log A and leave as u - log A)
__global__ void __launch_bounds__(256) k_rows_rsub(double* __restrict__ dst, const double* __restrict__ src, int64_t ld, int64_t nrows,
                                                   int64_t n) {
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            dst[r * ld + i] = src[r * ld + i] - dst[r * ld + i];
}
// Observables into log space on the device (mbar.py:858-867 shifts every observable to be positive, :886-903 takes its log):
//   level 1: part[r][blockIdx.x] = min over a slice of row r;
//   level 2: shift_r = min_r - |4 eps min_r| (so that the smallest shifted value is a positive number of relative size 4 eps, or
//            zero when the minimum is zero -- the reference's choice), row r <- log(row r - shift_r) in place, shift_r handed back.
__global__ void __launch_bounds__(256) k_rows_min_partial(const double* __restrict__ base, int64_t ld, int64_t nrows, int64_t n,
                                                          double* __restrict__ part) {
    __shared__ double red[4];
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y) {
        double m = INFINITY;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            m = fmin(m, base[r * ld + i]);
        m = -block256_max(-m, red);
        if (threadIdx.x == 0) part[r * gridDim.x + blockIdx.x] = m;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_rows_logshift(double* __restrict__ base, int64_t ld, int64_t nrows, int64_t n,
                                                       const double* __restrict__ part, int nparts, double* __restrict__ shift_out) {
    __shared__ double red[4];
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y) {
        double m = INFINITY;
        for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = fmin(m, part[r * nparts + i]);
        m = -block256_max(-m, red);
        const double shift = m - fabs(8.881784197001252e-16 * m);  // 4 eps (mbar.py:827-832)
        if (blockIdx.x == 0 && threadIdx.x == 0) shift_out[r] = shift;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            base[r * ld + i] = log(base[r * ld + i] - shift);
        __syncthreads();
    }
}
hipError_t launch_rows_logshift(hipStream_t s, double* base, int64_t ld, int64_t nrows, int64_t n, double* part, double* shift_out) {
    const int64_t want = (n + 2047) / 2048;
    const unsigned gx = (unsigned)(want < 256 ? (want < 1 ? 1 : want) : 256);
    const unsigned gy = (unsigned)(nrows < 1024 ? (nrows < 1 ? 1 : nrows) : 1024);
    hipLaunchKernelGGL(k_rows_min_partial, dim3(gx, gy), dim3(256), 0, s, base, ld, nrows, n, part);
    hipLaunchKernelGGL(k_rows_logshift, dim3(gx, gy), dim3(256), 0, s, base, ld, nrows, n, part, (int)gx, shift_out);
    return hipGetLastError();
}
hipError_t launch_rows_rsub(hipStream_t s, double* dst, const double* src, int64_t ld, int64_t nrows, int64_t n) {
    const int64_t want = (n + 255) / 256;
    const unsigned gx = (unsigned)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
    const unsigned gy = (unsigned)(nrows < 1024 ? (nrows < 1 ? 1 : nrows) : 1024);
    hipLaunchKernelGGL(k_rows_rsub, dim3(gx, gy), dim3(256), 0, s, dst, src, ld, nrows, n);
    return hipGetLastError();
}
hipError_t launch_rows_sub(hipStream_t s, double* dst, const double* src, int64_t ld, int64_t nrows, const double* v, int64_t n) {
    const int64_t want = (n + 255) / 256;
    const unsigned gx = (unsigned)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
    const unsigned gy = (unsigned)(nrows < 1024 ? (nrows < 1 ? 1 : nrows) : 1024);
    hipLaunchKernelGGL(k_rows_sub, dim3(gx, gy), dim3(256), 0, s, dst, src, ld, nrows, v, n);
    return hipGetLastError();
}

hipError_t launch_sci_update(hipStream_t s, const double* part, int64_t nparts, int64_t rows, const double* Nk,
                             const double* lnNk, int64_t K, int64_t Kp, int first_state, double tol, double* f,
                             double* aden, double* f_hist, double* delta_out) {
    const SciArgs q{Nk, lnNk, K, Kp, first_state, tol, f, aden, f_hist, delta_out};
    hipLaunchKernelGGL(k_sci_update, dim3(1), dim3(256), (size_t)(Kp + 256) * sizeof(double), s, part, nparts, rows, q);
    return hipGetLastError();
}

// first level only of the two-level reduction: out[c][i] = sum over the c-th chunk of 32 records; returns #chunks
hipError_t launch_reduce_level1(hipStream_t s, const double* part, int64_t nparts, int64_t count, double* out,
                                int64_t* nchunks) {
    const int64_t chunk = 32;
    const int64_t n1 = (nparts + chunk - 1) / chunk;
    const unsigned gx = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(k_reduce, dim3(gx, (unsigned)n1), dim3(256), 0, s, part, nparts, count, chunk, out);
    *nchunks = n1;
    return hipGetLastError();
}

hipError_t launch_mfma_peak(hipStream_t s, int blocks, int iters, double* sink) {
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, iters, sink);
    return hipGetLastError();
}

hipError_t launch_reduce2(hipStream_t s, const double* partA, int64_t countA, const double* partB, int64_t countB,
                          int64_t nparts, double* scratch, double* outA, double* outB) {
    const unsigned gx = (unsigned)((countA + 255) / 256 + (countB + 255) / 256);
    if (nparts <= 32) {
        hipLaunchKernelGGL(k_reduce2, dim3(gx, 1), dim3(256), 0, s, partA, countA, partB, countB, nparts, nparts, outA, outB);
        return hipGetLastError();
    }
    const int64_t chunk = 32;
    const int64_t n1 = (nparts + chunk - 1) / chunk;
    double* sA = scratch;
    double* sB = scratch + n1 * countA;
    hipLaunchKernelGGL(k_reduce2, dim3(gx, (unsigned)n1), dim3(256), 0, s, partA, countA, partB, countB, nparts, chunk, sA, sB);
    hipLaunchKernelGGL(k_reduce2, dim3(gx, 1), dim3(256), 0, s, (const double*)sA, countA, (const double*)sB, countB, n1, n1,
                       outA, outB);
    return hipGetLastError();
}

hipError_t launch_newton(hipStream_t s, const AdaptArgs& a) {
    const int M = a.m - 1;
    if (M > 127 || a.Kp > 128) return hipErrorInvalidValue;
    if (M <= 31)
        hipLaunchKernelGGL((k_newton<8, 4>), dim3(1), dim3(64), 0, s, a);
    else if (M <= 63)
        hipLaunchKernelGGL((k_newton<16, 4>), dim3(1), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_newton<16, 8>), dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_loop_reduce(hipStream_t s, const LoopSrc& src, int64_t count, int op, double* out) {
    if (count < 1 || src.n < 1 || src.n > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_loop_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, count, op, out);
    return hipGetLastError();
}

hipError_t launch_newton_chol(hipStream_t s, const AdaptArgs& a, double* work) {
    const int M = a.m - 1;
    if (M < 0 || M > CHOL_NP - 1 || a.Kp > 256) return hipErrorInvalidValue;
    double* Aw = work;
    double* thr = work + (size_t)CHOL_NP * CHOL_NP;
    hipLaunchKernelGGL(k_chol_setup, dim3(CHOL_NP / 16, CHOL_NP / 16), dim3(256), 0, s, a, Aw, thr);
    for (int j0 = 0; j0 < M; j0 += CB) {
        hipLaunchKernelGGL(k_chol_panel, dim3(1), dim3(256), 0, s, a, Aw, (const double*)thr, j0);
        const int rows_below = M + 1 - (j0 + CB);  // incl. the right-hand-side row
        if (rows_below > 0) {
            const int nt = (rows_below + CB - 1) / CB;
            hipLaunchKernelGGL(k_chol_update, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, s, a, Aw, j0);
        }
    }
    hipLaunchKernelGGL(k_chol_finish, dim3(1), dim3(256), 0, s, a, (const double*)Aw);
    return hipGetLastError();
}

size_t solve_small_record_doubles(int nb) { return (size_t)2 * nb * 16 + (size_t)nb * (nb + 1) / 2 * 256; }
int solve_small_grid(int num_cu, int64_t ntiles, int64_t grid_override) {
    int64_t g = (ntiles + 3) / 4;  // at least one tile per wave
    if (g > num_cu) g = num_cu;    // one workgroup per compute unit: the grid barriers need every workgroup resident
    if (g > 256) g = 256;          // (the in-kernel reduction sums at most 16 x 16 records per entry)
    if (grid_override > 0 && grid_override < g) g = grid_override;
    return (int)(g < 1 ? 1 : g);
}
template <int NB>
static hipError_t launch_solve_small_nb(hipStream_t s, int grid, const SmallArgs& a) {
    const size_t tile = (size_t)NB * 16 * TS * 8 + 2 * TS * 8;
    size_t lds = (size_t)4 * small_ring_depth(NB) * tile;
    const size_t fold = (size_t)4 * solve_small_record_doubles(NB) * sizeof(double);
    if (fold > lds) lds = fold;
    auto kern = k_solve_small<NB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_solve_small(hipStream_t s, int nb, int grid, const SmallArgs& a) {
    switch (nb) {
        case 1: return launch_solve_small_nb<1>(s, grid, a);
        case 2: return launch_solve_small_nb<2>(s, grid, a);
        case 3: return launch_solve_small_nb<3>(s, grid, a);
        case 4: return launch_solve_small_nb<4>(s, grid, a);
        case 5: return launch_solve_small_nb<5>(s, grid, a);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ctl_resume(hipStream_t s, int* ctl) {
    hipLaunchKernelGGL(k_ctl_resume, dim3(1), dim3(64), 0, s, ctl);
    return hipGetLastError();
}

hipError_t launch_select_newton(hipStream_t s, const AdaptArgs& a) {
    const int M = a.m - 1;
    if (M > 127 || a.Kp > 128) return hipErrorInvalidValue;
    if (M <= 63)
        hipLaunchKernelGGL((k_select_newton<4>), dim3(1), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_select_newton<8>), dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_select(hipStream_t s, const AdaptArgs& a) {
    if (a.Kp > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_select, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- P mode ------------------------------------------------------------------------------------------------------
LaunchGeom psweep_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override) {
    LaunchGeom g;
    const size_t tile = (size_t)nb * 16 * TS * 8 + TS * 8;
    g.variant = 1;
    g.waves = lse_waves(nb);
    g.lds_bytes = (size_t)g.waves * 2 * tile;  // no look-up tables: the sweep has no exponential
    int64_t want = (ntiles + g.waves - 1) / g.waves;
    int64_t cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
    if (grid_override > 0) cap = grid_override;
    if (want < 1) want = 1;
    g.blocks = (int)(want < cap ? want : cap);
    g.nwaves = g.blocks * g.waves;
    g.psum_records = g.nwaves;
    return g;
}

template <int NB>
static hipError_t launch_psweep_nb(hipStream_t s, int nf, const LaunchGeom& g, const double* P, int64_t ld, int64_t N,
                                   const double* cmul, const double* cw, double* rinv0, double* rinv1, double* pp,
                                   const LoopCtl& lc) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        if (lc.ev_start && lc.ev_stop)
            hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, P, ld, N,
                                  ntiles, cmul, cw, rinv0, rinv1, pp, lc.ctl, lc.slot_stride);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, P, ld, N, ntiles, cmul, cw, rinv0,
                               rinv1, pp, lc.ctl, lc.slot_stride);
        return hipGetLastError();
    };
    const bool wide = stage_offsets_wide(ld);
    if (nf == 1) return wide ? go(k_psweep<NB, 1, true>) : go(k_psweep<NB, 1, false>);
    return wide ? go(k_psweep<NB, 2, true>) : go(k_psweep<NB, 2, false>);
}

hipError_t launch_psweep(hipStream_t s, int nb, int nf, const LaunchGeom& g, const double* P, int64_t ld, int64_t N,
                         const double* cmul, const double* cw, double* rinv0, double* rinv1, double* pp, const LoopCtl& lc) {
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_psweep_nb<NB_>(s, nf, g, P, ld, N, cmul, cw, rinv0, rinv1, pp, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rinv_weighted(hipStream_t s, const double* rinv, const double* cw, int64_t N, double* out,
                                const LoopCtl& lc) {
    int64_t bx = (N + 255) / 256;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_rinv_weighted, dim3((unsigned)bx), dim3(256), 0, s, rinv, cw, N, out, lc.ctl, lc.slot_stride);
    return hipGetLastError();
}

// Fused build: single-candidate sweep at the anchor point + P + unit reciprocals.  Geometry of the classic sweep.
template <int NB>
static hipError_t launch_build_sweep_nb(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                        const double* aden, const double* cw, double* P, double* rinv_slot, double* pp) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, P, rinv_slot, pp);
        return hipGetLastError();
    };
    return stage_offsets_wide(ld) ? go(k_build_sweep<NB, true>) : go(k_build_sweep<NB, false>);
}
LaunchGeom build_sweep_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override) {
    return lse_geometry(nb, 1, num_cu, ntiles, grid_override, 1);  // default double-buffered sweep, tables in LDS
}
hipError_t launch_build_sweep(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                              const double* aden, const double* cw, double* P, double* rinv_slot, double* pp) {
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_build_sweep_nb<NB_>(s, g, u, ld, N, aden, cw, P, rinv_slot, pp);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

// Fused sweep (P mode): geometry of the full Gram panel -- one workgroup of four waves per CU, two tile buffers per wave.
LaunchGeom fused_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override) {
    LaunchGeom g;
    g.waves = 4;
    g.variant = 1;
    if (nb > 8) {  // 129 .. 256 states: one tile stream per workgroup, one partial record per workgroup (k_fused_quad)
        g.variant = 6;
        g.lds_bytes = (size_t)2 * ((size_t)nb * 16 * TS * 8 + 4 * 1024) + 1024;
        int64_t capq = grid_override > 0 ? grid_override : num_cu;
        int64_t wantq = ntiles < 1 ? 1 : ntiles;
        g.blocks = (int)(wantq < capq ? wantq : capq);
        g.nwaves = g.blocks;
        g.psum_records = g.nwaves;
        return g;
    }
    const size_t tile = (size_t)nb * 16 * TS * 8 + 1024;    // + one LDS-DMA piece for the two weight vectors
    g.lds_bytes = (size_t)4 * 2 * tile + (size_t)nb * 512;  // + the candidates' multipliers as a 4x4x4 MFMA operand
    int64_t want = (ntiles + 3) / 4;
    int64_t cap = num_cu;
    if (nb <= 5) {  // narrow panels: few accumulators, several workgroups per CU (cf. gram_geometry)
        static const int occ[6] = {1, 4, 4, 3, 2, 2};
        const int by_lds = blocks_per_cu_for(g.lds_bytes);
        cap = (int64_t)num_cu * (by_lds < occ[nb] ? by_lds : occ[nb]);
        // ... but every wave leaves a partial record (NB (NB + 1) / 2 blocks of 2 KB + the per-state sums) and pays a prologue:
        // a second workgroup per CU only once a wave has ~32 tiles to work on (config 5, K = 40, N = 95 000: 58 instead of
        // 64 us per iteration with one workgroup per CU; K = 32, N = 1e6 is fastest at two)
        int64_t by_work = (ntiles + 127) / 128;
        if (by_work < num_cu) by_work = num_cu;
        if (cap > by_work) cap = by_work;
    }
    if (grid_override > 0) cap = grid_override;
    if (want < 1) want = 1;
    g.blocks = (int)(want < cap ? want : cap);
    g.nwaves = g.blocks * 4;
    g.psum_records = g.nwaves;
    return g;
}
template <int NB>
static hipError_t launch_fused_nb(hipStream_t s, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                                  const double* cw, const double* wsq, double* rinv0, double* gp, double* pp, const LoopCtl& lc) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        double* r1 = nullptr;
        if (lc.ev_start && lc.ev_stop)
            hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, P, ld, N, ntiles, cmul,
                                  cw, wsq, rinv0, r1, gp, pp, lc.ctl, lc.slot_stride);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, P, ld, N, ntiles, cmul, cw, wsq, rinv0, r1, gp, pp,
                               lc.ctl, lc.slot_stride);
        return hipGetLastError();
    };
    return stage_offsets_wide(ld) ? go(k_fused<NB, true>) : go(k_fused<NB, false>);
}
template <int NBT>
static hipError_t launch_fused_quad_t(hipStream_t s, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                                      const double* cw, const double* wsq, double* rinv0, double* gp, double* pp, const LoopCtl& lc) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        if (lc.ev_start && lc.ev_stop)
            hipExtLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, lc.ev_start, lc.ev_stop, 0, P, ld, N, ntiles, cmul,
                                  cw, wsq, rinv0, gp, pp, lc.ctl, lc.slot_stride);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, P, ld, N, ntiles, cmul, cw, wsq, rinv0, gp, pp, lc.ctl,
                               lc.slot_stride);
        return hipGetLastError();
    };
    if (g.live_blocks > 0 && g.live_blocks <= NBT - 2)
        return stage_offsets_wide(ld) ? go(k_fused_quad<NBT, true, NBT - 2>) : go(k_fused_quad<NBT, false, NBT - 2>);
    return stage_offsets_wide(ld) ? go(k_fused_quad<NBT, true>) : go(k_fused_quad<NBT, false>);
}
hipError_t launch_make_p(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                         const double* logden, double* P) {
    hipLaunchKernelGGL(k_make_p, dim3((unsigned)(num_cu * 8)), dim3(256), 0, s, u, ld, N, rows, aden, logden, P);
    return hipGetLastError();
}
hipError_t launch_fill(hipStream_t s, double* v, double value, int64_t n) {
    int64_t bx = (n + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0, s, v, value, n);
    return hipGetLastError();
}
hipError_t launch_sqrt_vec(hipStream_t s, double* dst, const double* src, int64_t n) {
    int64_t bx = (n + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(k_sqrt_vec, dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}
hipError_t launch_fused(hipStream_t s, int nb, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                        const double* cw, const double* wsq, double* rinv_base, double* gram_part, double* psum_part,
                        const LoopCtl& lc) {
    if (!lc.ctl) return hipErrorInvalidValue;  // (the slot vectors are addressed through the control words)
    if (nb == 12) return launch_fused_quad_t<12>(s, g, P, ld, N, cmul, cw, wsq, rinv_base, gram_part, psum_part, lc);
    if (nb == 16) return launch_fused_quad_t<16>(s, g, P, ld, N, cmul, cw, wsq, rinv_base, gram_part, psum_part, lc);
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_fused_nb<NB_>(s, g, P, ld, N, cmul, cw, wsq, rinv_base, gram_part, psum_part, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

// Build sweep + Gram matrix at the anchor: geometry of the fused sweep (same partial-record counts), tables in LDS.
LaunchGeom build_gram_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override) {
    // (same grid as the fused sweep, so that the partial-record counts agree; the kernel strides over tiles, so it does
    // not matter if the look-up tables leave room for one workgroup per CU less)
    LaunchGeom g = fused_geometry(nb, num_cu, ntiles, grid_override);
    g.lds_bytes = (size_t)4 * 2 * ((size_t)nb * 16 * TS * 8 + 2 * TS * 8) + EXP_TABLE_BYTES;  // (its own tile layout + the tables)
    return g;
}
template <int NB>
static hipError_t launch_build_gram_nb(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                       const double* aden, const double* cw, const double* wsq, double* P, double* rinv_slot,
                                       double* pp, double* gp) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, u, ld, N, ntiles, aden, cw, wsq, P, rinv_slot, pp, gp);
        return hipGetLastError();
    };
    return stage_offsets_wide(ld) ? go(k_build_gram<NB, true>) : go(k_build_gram<NB, false>);
}
hipError_t launch_build_gram(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                             const double* aden, const double* cw, const double* wsq, double* P, double* rinv_slot,
                             double* psum_part, double* gram_part) {
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_build_gram_nb<NB_>(s, g, u, ld, N, aden, cw, wsq, P, rinv_slot, psum_part, gram_part);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mbar
