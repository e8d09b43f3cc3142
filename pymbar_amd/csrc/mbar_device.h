// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path: device-side helpers shared by the kernel translation units
// (mbar_k_eval.hip, mbar_k_gram.hip, mbar_k_quad.hip, mbar_k_pmode.hip, mbar_k_fused.hip, mbar_k_solver.hip).
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
#ifndef MBAR_DEVICE_H
#define MBAR_DEVICE_H
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
// The value of lane (l ^ M): across the 16-lane rows through the LDS crossbar (ds_bpermute), inside a row by DPP -- row_ror:8,
// two bank-masked row shifts by 4, the quad permutes -- whose latency is an instruction's, not a round trip's.  A butterfly
// built from it adds the same pairs in the same order as one built from __shfl_xor.
template <int M>
__device__ __forceinline__ double lane_xor(double x) {
    if constexpr (M >= 16) {
        return __shfl_xor(x, M);
    } else if constexpr (M == 8) {
        return dpp_move<0x128>(x);
    } else if constexpr (M == 4) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        int tl = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xF, 0x5, false);  // quads 0, 2: lane l + 4 (row_shl:4)
        tl = __builtin_amdgcn_update_dpp(tl, lo, 0x114, 0xF, 0xA, false);      // quads 1, 3: lane l - 4 (row_shr:4)
        int th = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0x5, false);
        th = __builtin_amdgcn_update_dpp(th, hi, 0x114, 0xF, 0xA, false);
        return __hiloint2double(th, tl);
    } else if constexpr (M == 2) {
        return dpp_move<0x4E>(x);
    } else {
        return dpp_move<0xB1>(x);
    }
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
constexpr int EXP_TABLE_DMA_PER_WAVE = (EXP_TABLE_BYTES / 1024 + 7) / 8;  // requests per wave of exp_table_init_dma in an 8-wave workgroup
typedef __attribute__((address_space(3))) const double lds_cdouble;
// Every thread block copies the table to LDS offset 0 (its dynamic LDS starts there: the kernels have no static
// __shared__), so a look-up address is just the masked integer -- no base add.  Callers barrier afterwards.
__device__ __forceinline__ void exp_table_init(char* smem) {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    for (int i = threadIdx.x; i < (1 << EXP2_BITS); i += blockDim.x) reinterpret_cast<double*>(smem)[i] = EXP2_TABLE[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        reinterpret_cast<double*>(smem + EXP2_TABLE_BYTES)[i] = LOG_TABLE[i];
}
// The same copy as LDS-DMA requests (1 KB per instruction, 18 in all, shared out over the waves): asynchronous, nothing passes
// through registers, and -- issued BEFORE a kernel's first tile request -- it does not queue behind it.  The caller waits for
// its own requests (s_waitcnt vmcnt(0)) and passes a workgroup barrier before the first look-up.
__device__ __forceinline__ void exp_table_init_dma(char* smem) {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    constexpr int NCH = EXP_TABLE_BYTES / 1024, NCH_EXP = EXP2_TABLE_BYTES / 1024;
    // (every wave issues the same number of requests, EXP_TABLE_DMA_PER_WAVE for eight waves -- a chunk requested twice lands
    // twice with the same bytes -- so that a caller can count them in an s_waitcnt)
    const int per_wave = (NCH + nwv - 1) / nwv;
    for (int i = 0; i < per_wave; ++i) {
        const int c = (wave + i * nwv) % NCH;
        const char* src = c < NCH_EXP ? reinterpret_cast<const char*>(EXP2_TABLE) + c * 1024
                                      : reinterpret_cast<const char*>(LOG_TABLE) + (c - NCH_EXP) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
    }
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
// 2^(-w/S) for w >= 0 (round 6; the build sweep's exponential, where the argument is "column maximum minus entry" and can
// be formed non-negative EXACTLY): floor and fraction come straight from the argument -- n = -trunc(w) (one conversion with a
// negated source), z = fract(w) in [0, 1) (one instruction) -- instead of rint / subtract / convert, and no clamp is needed below:
//   2^(-w/S) = 2^(n/S) 2^(-z/S) = ldexp(T[n & (S-1)] * Q(z), n >> EXP2_BITS),  Q = degree-3 polynomial of 2^(-z/S) on [0, 1]
// (error 8.8e-18, tools/gen_exp2_poly.py).  Eight fp64 + three integer instructions (exp2s_batch: nine + three, + the clamp).  Huge
// finite w is safe (v_cvt_i32_f64 saturates, ldexp(.., -2^20) = 0); CLAMP = true also takes w = +inf (matrices with +inf entries).
constexpr double EXP2N_POLY[4] = {1.0, -0x1.62e42fefa3028p-12, 0x1.ebfbdf9648000p-25, -0x1.c69cea0000000p-38};
template <int N, bool CLAMP>
__device__ __forceinline__ void exp2s_neg_batch(double (&w)[N]) {
    double T[N];
    int q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t = CLAMP ? fmin(w[i], 2.0e9) : w[i];
        const int ni = (int)(-t);
        w[i] = __builtin_amdgcn_fract(t);
        q[i] = ni >> EXP2_BITS;
        T[i] = exp2_table_at(ni);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double p = EXP2N_POLY[3];
        p = fma(p, w[i], EXP2N_POLY[2]);
        p = fma(p, w[i], EXP2N_POLY[1]);
        w[i] = fma(p, w[i], EXP2N_POLY[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = ldexp(T[i] * w[i], q[i]);
}
template <int N, bool CLAMP>
__device__ __forceinline__ void exp2s_neg_batch2(double (&w0)[N], double (&w1)[N]) {  // two argument sets, one pipeline
    double T0[N], T1[N];
    int q0[N], q1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t0 = CLAMP ? fmin(w0[i], 2.0e9) : w0[i], t1 = CLAMP ? fmin(w1[i], 2.0e9) : w1[i];
        const int n0 = (int)(-t0), n1 = (int)(-t1);
        w0[i] = __builtin_amdgcn_fract(t0);
        w1[i] = __builtin_amdgcn_fract(t1);
        q0[i] = n0 >> EXP2_BITS;
        q1[i] = n1 >> EXP2_BITS;
        T0[i] = exp2_table_at(n0);
        T1[i] = exp2_table_at(n1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double p0 = EXP2N_POLY[3], p1 = EXP2N_POLY[3];
        p0 = fma(p0, w0[i], EXP2N_POLY[2]);
        p1 = fma(p1, w1[i], EXP2N_POLY[2]);
        p0 = fma(p0, w0[i], EXP2N_POLY[1]);
        p1 = fma(p1, w1[i], EXP2N_POLY[1]);
        w0[i] = fma(p0, w0[i], EXP2N_POLY[0]);
        w1[i] = fma(p1, w1[i], EXP2N_POLY[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        w0[i] = ldexp(T0[i] * w0[i], q0[i]);
        w1[i] = ldexp(T1[i] * w1[i], q1[i]);
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

// A global load the compiler does not know to be one: issued AHEAD of LDS-DMA requests whose number depends on control flow,
// its result would otherwise be waited for with a conservative s_waitcnt vmcnt(0) -- i.e. for the tile behind it as well.  The
// caller waits by hand (wait_vm_for: vmcnt(N) with N <= the requests issued after these loads) before the first use.  Only
// safe for loads issued before every compiler-tracked memory operation of the kernel (the counter retires in order).
__device__ __forceinline__ double load_untracked(const double* p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vm_for(double& a, double& b, double& c, double& d, double& e, double (&r)[16]) {
    asm volatile("s_waitcnt vmcnt(%21)"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),
                   "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N)
                 : "memory");
}
// Workgroup barrier for LDS traffic only: __syncthreads() carries a release fence that also waits for every outstanding global
// request (vmcnt(0)) -- i.e. for LDS-DMA tiles in flight that the code between two such barriers has no business waiting for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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

// ---- constants shared by several kernel families --------------------------------------------------------------------
constexpr int lse_waves(int nb) { return nb <= 2 ? 16 : (nb <= 4 ? 8 : (nb <= 8 ? 4 : 2)); }
constexpr int TSS = 64;  // samples per tile of the small-K kernel
constexpr int GRAM_AGPR_BLOCKS = 31;
// Finite stand-ins for the infinities of the operand exponent when the sweep runs without the exponential's clamp
constexpr double GRAM_NEG_HUGE = -2.9e303, LOGDEN_HUGE = 1e300;
constexpr int FUSED_PSUM1_FROM_GRAM_NB = 8;

static inline int blocks_per_cu_for(size_t lds_bytes) {
    int b = (int)((160 * 1024) / lds_bytes);
    if (b < 1) b = 1;
    if (b > 4) b = 4;
    return b;
}

static inline bool stage_offsets_wide(int64_t ld) { return (uint64_t)ld * 56u + 128u >= (1ull << 32); }

// (defined in mbar_k_quad.hip; launch_fused in mbar_k_fused.hip hands 129 .. 256 states to it)
hipError_t launch_fused_quad(hipStream_t s, int nbt, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                             const double* cw, const double* wsq, double* rinv0, double* gp, double* pp, const LoopCtl& lc);

}  // namespace mbar
#endif  // MBAR_DEVICE_H
