// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- reductions, small kernels, log-space per-state sums, log W, generator, and the K x K work of the device-resident loop (k_newton, k_chol_*, k_select).
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

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
// zero fill at HBM write speed (16 bytes per lane and store; hipMemsetAsync's fill kernel runs at ~1 TB/s: 20 ms for the 20 GB of an
// augmented matrix of the expectation family, 10 ms for config 3's matrix in front of its upload)
__global__ void __launch_bounds__(256) k_zero16(uint4* __restrict__ p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}
// dst = sqrt(src): the roots of the sample multiplicities for the matrix-core operands of the weighted sweeps
__global__ void __launch_bounds__(256) k_sqrt_vec(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = sqrt(src[i]);
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
    // Last iteration without its Gram matrix (CTL_LIGHT): the stop test of mbar_solvers.py:627-636 compares the ACCEPTED candidate
    // with the current f and the two candidates with each other.  If it holds for BOTH candidates, the coming iteration is the
    // last whichever wins, and the sweep only has to say which one does: flags bit 3 collects "some state fails it".
    const double tol = q.prm[1], tol_small = tol < 1e-8 ? tol : 1e-8, tol_diff = sqrt(tol);
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
            if (k != first) {
                const double ds = fabs(fs) < tol_small ? 1.0 : fabs(fs), dn = fabs(fn) < tol_small ? 1.0 : fabs(fn);
                const double gap = fabs(fs - fn);
                if (!(fabs(fs - fk) / ds < tol) || !(fabs(fn - fk) / dn < tol) || !(gap / ds < tol_diff) || !(gap / dn < tol_diff)) flags |= 8;
            }
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
    {   // bitwise OR over the workgroup (__syncthreads_or is a logical one)
        __shared__ int s_flags;
        if (tid == 0) s_flags = 0;
        __syncthreads();
        if (flags) atomicOr(&s_flags, flags);
        __syncthreads();
        flags = s_flags;
    }
    if ((flags & 7) != 0 && tid == 0) {
        q.ctl[CTL_REASON] = (flags & 1) ? 1 : ((flags & 4) ? 3 : 2);
        q.ctl[CTL_DONE] = 2;
    }
    if (tid == 0) q.ctl[CTL_LIGHT] = (q.light_ok && q.fused && flags == 0 && q.prm[3] != 0.0) ? 1 : 0;
}

// The elimination proper: publish the pivot columns, barrier, read them back, update -- a step's latency chain (LDS round trip,
// barrier, reciprocals, multipliers) and its fused multiply-adds one after the other.  Instrumented (MBAR_DEBUG_STAMPS, shader
// clocks): 1770 per two-pivot step at 127 unknowns (~900 of them the chain), 1300 at 39.  Measured dead end (round 4): running the
// NEXT step's chain under THIS step's arithmetic -- the owners of the next pivot columns bring those entries up to date first and
// publish them, barrier, everybody reads the next columns and starts the reciprocals, then the rest of the tile is updated;
// bit-identical to this form (every entry keeps its order of updates), but 2.2x SLOWER as compiled: the priority update is a
// latency chain of its own (LDS read, two FMAs, LDS write, barrier) and the in-order wave does not start the bulk FMAs while
// the reciprocals are in flight unless every stage is interleaved by hand.
template <int T, int R>
__device__ __forceinline__ bool eliminate_serial(double (&A)[R][R], double (&colbuf)[2][2][R * T], double* pv, int M, double piv_thr,
                                                 int tid, int ty, int tx) {
    constexpr int NC = R * T;
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
            // (the two reciprocals side by side instead of one after the other: 1 / p2 = p1 / (a22 p1 - a12^2); the pivot p2 itself
            // -- recorded, and compared with the threshold -- as before)
            const double det = fma(a22, p1, -a12 * a12);
            const double inv1 = recip_fast(p1);
            const double invdet = recip_fast(det);
            const double l = a12 * inv1;
            const double p2 = fma(-l, a12, a22);
            const double inv2 = p1 * invdet;
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
    return bad;
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
__device__ __forceinline__ void newton_body(const AdaptArgs& q, long long* st = nullptr) {  // (T * T threads; the pointers of q may be LDS or global)
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
    {   // (a maximum over the workgroup: the serial loop over the sampled states was ~2 us of dependent LDS reads)
        __shared__ double s_pmax[NT / 64];
        double v = 0.0;
        for (int i = tid; i < q.m; i += NT) v = fmax(v, s_ps[smp[i]]);
        v = wave_max(v);
        if ((tid & 63) == 0) s_pmax[tid >> 6] = v;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) pmax = fmax(pmax, s_pmax[w]);
    }
    const double piv_thr = pmax * 2.220446049250313e-16 * (double)(M > 0 ? M : 1);
    if (st && tid == 0) st[2] = clock64();
    const bool bad = eliminate_serial<T, R>(A, colbuf, pv, M, piv_thr, tid, ty, tx);
    if (st && tid == 0) st[3] = clock64();
    if (tx == T - 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) rh[ty + T * r] = A[r][R - 1];
    }
    __syncthreads();
    if (tid == 0) xs[0] = 0.0;
    if (tid < M) xs[tid + 1] = rh[tid] / pv[tid];
    __syncthreads();
    if (st && tid == 0) st[4] = clock64();

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
// gram_lds: SELECT_GRAM_LDS_DOUBLES doubles of dynamic LDS (or nullptr): the reduced Gram blocks are staged there with coalesced
// loads -- all 36 in flight -- before the matrix-vector product below reads them.
constexpr int SELECT_GRAM_PITCH = 17;  // a 16 x 16 block's rows are stored 17 doubles apart: the column walk of a row is conflict-free
constexpr int SELECT_GRAM_LDS_DOUBLES = 36 * 16 * SELECT_GRAM_PITCH;
__device__ __forceinline__ void select_body(const AdaptArgs& q, double* gram_lds = nullptr, long long* st = nullptr) {  // (256 threads; the pointers of q may be LDS or global)
    int* ctl = q.ctl;
    const int tid = threadIdx.x, Kp = q.Kp;
    // Every input is loaded up front and unconditionally (both halves of aden / lse_red, the choice between them made afterwards):
    // ONE round of memory latency instead of a chain of three, with the 36 loads of the staged Gram blocks queued behind them.
    const bool in = tid < Kp;
    const int done = ctl[CTL_DONE], spec_w = ctl[CTL_SPEC], light_w = ctl[CTL_LIGHT], sci_w = ctl[CTL_SCI];
    const double tol = q.prm[1], prm2 = q.prm[2], prm3 = q.prm[3];
    const int first = q.sampled[0];
    const double nk = in ? q.Nk[tid] : 0.0;
    const double ad0 = (in && q.pmode) ? q.aden[tid] : 1.0, ad1 = (in && q.pmode) ? q.aden[Kp + tid] : 1.0;
    const double rat = (in && !q.pmode) ? q.ratio[tid] : 0.0;
    const double ls0 = in ? q.lse_red[tid] : 0.0, ls1 = in ? q.lse_red[Kp + tid] : 0.0;
    const double fo = in ? q.f[tid] : 0.0, fs = in ? q.cand[tid] : 0.0, fn = in ? q.cand[Kp + tid] : 0.0;
    const double lnk = in ? q.lnNk[tid] : 0.0;
    constexpr int NBLK_STAGE = 36;
    double gstage[NBLK_STAGE];
    if (gram_lds) {
#pragma unroll
        for (int b = 0; b < NBLK_STAGE; ++b) gstage[b] = q.gram_red[b * 256 + tid];
    }
    if (done != 0) return;
    const int min_sc = (int)prm2;
    const bool check = prm3 != 0.0;
    const bool sampled = in && tid < q.K && nk > 0.0;
    // the sweeps accumulate UNSCALED per-state sums: times the candidate's per-state constant = its psum
    // (classic: ratio c_k of the second candidate only; P mode: exp(a - a0) of both, kept in aden).  Index 0 = the
    // self-consistent candidate, 1 = Newton-Raphson; the fused sweep may have been handed them in swapped order (CTL_SPEC).
    const bool swap = q.fused && spec_w == 0;
    const bool light = q.fused && q.light_ok && light_w != 0;  // this iteration's sweep was the plain one: no Gram matrix
    const double m0 = swap ? ad1 : ad0;
    const double m1 = in ? (q.pmode ? (swap ? ad0 : ad1) : rat) : 0.0;
    double raw0 = swap ? ls1 : ls0, raw1 = swap ? ls0 : ls1;
    if (st && tid == 0) st[8] = st[9] = st[10] = clock64();
    if (gram_lds) {
#pragma unroll
        for (int b = 0; b < NBLK_STAGE; ++b) gram_lds[b * (16 * SELECT_GRAM_PITCH) + (tid >> 4) * SELECT_GRAM_PITCH + (tid & 15)] = gstage[b];
    }
    if (q.fused && !light && Kp == 128 && FUSED_PSUM1_FROM_GRAM_NB <= 8) {  // (k_fused<8> only: narrower panels and k_fused_quad accumulate both rows)
        // the fused sweep of a full panel left the unscaled sums of its SECOND multiplier row c to be taken from the Gram matrix
        // it accumulated for that candidate: sum_n w_n P_kn / s_n = sum_j c_j G'_kj (rows of p sum to one)
        __shared__ double s_c[128], s_half[128];
        if (tid < Kp) s_c[tid] = q.aden[Kp + tid];
        __syncthreads();
        // (two threads per state, half of the columns each.  Straight from global memory this cost ~15 us -- 64 dependent-ish
        // loads per thread, neighbouring threads 128 bytes apart; staged through LDS first it is one round of loads)
        const int k = tid & 127, h = tid >> 7, nb = Kp / 16;
        double acc = 0.0;
        if (gram_lds) {
            if (st && tid == 0) st[9] = clock64();
            // (block by block: the block index and whether it is read transposed are the same for the 16 columns of a block, the
            // inner loop is 16 multiply-adds on a fixed stride; the terms are added in the same order as before -- j ascending)
            const int I = k >> 4, ki = k & 15;
#pragma unroll
            for (int Jq = 0; Jq < 4; ++Jq) {
                const int J = 4 * h + Jq;
                const int lo = I < J ? I : J, hi = I < J ? J : I;
                const int b = lo * nb - (lo * (lo - 1)) / 2 + (hi - lo);
                const double* gp = gram_lds + b * (16 * SELECT_GRAM_PITCH) + (I <= J ? ki * SELECT_GRAM_PITCH : ki);
                const int stride = I <= J ? 1 : SELECT_GRAM_PITCH;
#pragma unroll
                for (int kj = 0; kj < 16; ++kj) acc = fma(s_c[16 * J + kj], gp[kj * stride], acc);
            }
        } else {
#pragma unroll 8
            for (int j = h * 64; j < h * 64 + 64; ++j) acc = fma(s_c[j], gram_elem(q.gram_red, nb, k, j), acc);
        }
        if (h == 1) s_half[k] = acc;
        __syncthreads();
        if (h == 0) acc += s_half[k];
        if (swap) raw0 = acc; else raw1 = acc;
        if (st && tid == 0) st[10] = clock64();
    }
    const double ps0 = raw0 * m0;
    const double ps1 = raw1 * m1;
    const double ga = sampled ? ps0 - nk : 0.0, gb = sampled ? ps1 - nk : 0.0;
    // convergence measures over the sampled states except the gauge state (:627-633); NaN: see the host loop.  They depend on the
    // choice through f_new, so both variants go through the ONE exchange that also carries the two gradient norms (five block
    // reductions one after the other were ~6 k clocks of butterflies and barriers): sums in the order of block256_sum, maxima.
    const bool counts = sampled && tid != first;
    const double small = tol < 1e-8 ? tol : 1e-8;
    double rv[8];
    rv[0] = ga * ga;
    rv[1] = gb * gb;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const double fc = c2 == 0 ? fs : fn;
        const double div = fabs(fc) < small ? 1.0 : fabs(fc);
        const double d1 = counts ? fabs(fc - fo) / div : 0.0;
        const double d2 = counts ? fabs(fs - fn) / div : 0.0;
        rv[2 + 3 * c2] = (d1 != d1) ? 1.0 : 0.0;
        rv[3 + 3 * c2] = d1 != d1 ? 0.0 : d1;
        rv[4 + 3 * c2] = d2 != d2 ? 0.0 : d2;
    }
    auto butterfly = [&](auto mtag) {
        constexpr int m = decltype(mtag)::value;
        double o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = lane_xor<m>(rv[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) rv[i] = i < 2 ? rv[i] + o[i] : fmax(rv[i], o[i]);
    };
    butterfly(std::integral_constant<int, 32>());
    butterfly(std::integral_constant<int, 16>());
    butterfly(std::integral_constant<int, 8>());
    butterfly(std::integral_constant<int, 4>());
    butterfly(std::integral_constant<int, 2>());
    butterfly(std::integral_constant<int, 1>());
    __shared__ double red8[4][8];
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red8[tid >> 6][i] = rv[i];
    }
    __syncthreads();
    const double gs = (red8[0][0] + red8[1][0]) + (red8[2][0] + red8[3][0]);
    const double gn = (red8[0][1] + red8[1][1]) + (red8[2][1] + red8[3][1]);
    if (st && tid == 0) st[11] = clock64();
    // :607 (every thread holds the same sums); a NaN Newton gradient loses against a finite self-consistent one (host loop)
    const int ch = (gs < gn || (gn != gn && gs == gs) || sci_w < min_sc) ? 0 : 1;
    const double fnew = ch == 0 ? fs : fn;
    if (in) {
        q.f[tid] = fnew;
        q.psum[tid] = ch == 0 ? ps0 : ps1;
        q.anum[tid] = sampled ? fnew + lnk : -INFINITY;
        if (q.pmode) q.ccur[tid] = ch == 0 ? m0 : m1;
    }
    const int ro = 2 + 3 * ch;
    const double nan_seen = fmax(fmax(red8[0][ro], red8[1][ro]), fmax(red8[2][ro], red8[3][ro]));
    double max_delta = fmax(fmax(red8[0][ro + 1], red8[1][ro + 1]), fmax(red8[2][ro + 1], red8[3][ro + 1]));
    const double max_diff = fmax(fmax(red8[0][ro + 2], red8[1][ro + 2]), fmax(red8[2][ro + 2], red8[3][ro + 2]));
    if (nan_seen > 0.0) max_delta = NAN;
    if (st && tid == 0) st[12] = clock64();
    // Fused sweep: the Gram matrix of the Newton-Raphson candidate is already there.  It serves the next iteration when
    // that candidate was accepted -- or when the two candidates coincide to 1e-10 (at the fixed point the choice is
    // round-off noise; the Hessian of one is the Hessian of the other far below any tolerance it is used at).
    const int spec = swap ? 0 : 1;  // the candidate the sweep speculated on
    const bool reuse = q.fused && !light && (ch == spec || max_diff <= 1e-10);
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
        if (light) {
            ctl[CTL_LIGHTS] += 1;
            ctl[CTL_LIGHT] = 0;  // (the next Newton solve -- if there is one: this iteration was to be the last -- decides afresh)
        }
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
k_select(AdaptArgs q, int gram_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double select_dyn_lds[];
    select_body(q, gram_in_lds ? select_dyn_lds : nullptr);
}
// ---------------------------------------------------------------------------------------------
// The K x K Newton solve as a BLOCKED LDL^T factorisation on the fp64 matrix cores (round 6; up to 128 states, the default).
// The register Gauss-Jordan solve above (newton_body) does K^3 / 2 multiply-adds in 64 two-pivot steps with a barrier each and
// is bound by the vector issue rate of one compute unit (114 k clocks at 127 unknowns); here the work is K^3 / 6, almost all of
// it rank-16 updates of 16 x 16 blocks (four v_mfma_f64_16x16x4_f64 each), with TWO barriers per block column of 16 pivots
// instead of one per two pivots.  tools/newton_ldlt_model.py is the lane-level numpy statement of the same algorithm.
//   * Rows / columns are the states 0 .. Kp-1 in their natural order, so the 16 x 16 blocks are the blocks of the reduced Gram
//     record.  live = sampled and not the gauge state; every other row is an identity row (pivot 1, no coupling, x = 0).  Row 0
//     is never live (state 0 is the gauge state or unsampled): it carries the right-hand side g and RIDES ALONG -- pivots run
//     from the LAST row upwards (H = U D U^T), so row 0 is eliminated by every pivot and never is one.
//   * Block (i, j), i <= j, is held TRANSPOSED in the accumulator layout of the matrix instruction: lane (g, r), register t <->
//     row 16 i + r, column 16 j + 4 t + g.  In that layout the pivot row of the diagonal block and column p of a panel block
//     are operands AS THEY STAND (the 16 lanes of group g = p & 3 of register t = p >> 2; the other lane groups of the operand
//     are zero), a pivot is one rank-1 matrix instruction per block of the block column, and register t of a finished panel
//     block is the K-chunk {pivots 4 t + g} of the rank-16 update -- no transposition anywhere.
//   * Every wave eliminates the diagonal block for itself (bit-identical copies) and carries its share of the panel blocks
//     (block (i, j) lives in wave (i + j) & 3, slot i >> 2; diagonal masters in wave j & 3), so there is no exchange inside a
//     block column.  Per block column: frozen panel -> LDS (plain V, and W = -V / d), barrier, every wave updates its blocks of
//     the trailing matrix B_ij += V_j W_i^T, the master of the next diagonal block goes to LDS, barrier.
//   * x by forward substitution on the unit triangle W with x_0 = -1 (the ride-along row makes the right-hand side one more
//     column of the triangle): one thread per unknown, 16 sequential steps per diagonal block through v_readlane.
// Pivots are recorded and tested against the same threshold as before (a live pivot <= eps * M * max psum, or a non-finite one,
// hands the solve to the host loop); they are the pivots of the REVERSED elimination order, not of the ascending one.
// LDS: 44 blocks of 2 KB (7 plain panels, 28 scaled panels, 8 scaled diagonal blocks, 1 exchange) in the dynamic allocation,
// which the selection's staged Gram blocks (dead once the set-up has them in registers) share.
// ---------------------------------------------------------------------------------------------
constexpr int LDLT_BLK = 4 * 64;             // doubles of a 16 x 16 block in register order [t][lane]
constexpr int LDLT_V_OFF = 0;                // [7]  plain panel blocks (i, kb) of the current block column
constexpr int LDLT_W_OFF = 7 * LDLT_BLK;     // [28] scaled panel blocks, index kb (kb - 1) / 2 + i
constexpr int LDLT_DW_OFF = 35 * LDLT_BLK;   // [8]  scaled frozen diagonal blocks
constexpr int LDLT_DN_OFF = 43 * LDLT_BLK;   // [1]  the next diagonal block, from its master to every wave
constexpr int LDLT_LDS_DOUBLES = 44 * LDLT_BLK;
static_assert(LDLT_LDS_DOUBLES >= SELECT_GRAM_LDS_DOUBLES, "the staged Gram blocks share the allocation");

__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ v4d ldlt_read_block(const double* p, int lane) {
    v4d r;
#pragma unroll
    for (int t = 0; t < 4; ++t) r[t] = p[t * 64 + lane];
    return r;
}
__device__ __forceinline__ void ldlt_write_block(double* p, int lane, const v4d& v) {
#pragma unroll
    for (int t = 0; t < 4; ++t) p[t * 64 + lane] = v[t];
}
__device__ __forceinline__ void ldlt_rank16(v4d& acc, const v4d& a, const v4d& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
}

template <int NB>  // block rows the instantiation has registers for: 4 (up to 64 states) or 8 (up to 128)
__device__ __forceinline__ void newton_body_ldlt(const AdaptArgs& q, double* ws, const double* gram_lds, long long* st = nullptr) {  // 256 threads
    __shared__ double s_f[128], s_ps[128], s_nk[128], s_ln[128], s_cl[128], s_a0[128], s_b[128], s_dg[128], s_x[128], s_pv[128];
    __shared__ int s_live[128], s_first[1], s_pos[128];
    __shared__ double s_pmax[4];
    if (q.ctl[CTL_DONE] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = lane >> 4, R = lane & 15;
    const int nb = q.Kp / 16, M = q.m - 1;
    const int first = q.sampled[0];
    double pm = 0.0;
    if (tid < 128) {
        const int k = tid;
        const bool in = k < q.Kp;
        const double nk = in ? q.Nk[k] : 0.0, ps = in ? q.psum[k] : 0.0;
        const bool sampled = k < q.K && nk > 0.0;
        const bool live = sampled && k != first;
        s_f[k] = k < q.K ? q.f[k] : 0.0;
        s_ps[k] = ps;
        s_nk[k] = nk;
        s_ln[k] = in ? q.lnNk[k] : 0.0;
        const double cc = (in && q.pmode) ? (q.fused ? q.cgram[k] : q.ccur[k]) : 1.0;  // the multipliers the Gram sweep left out
        s_cl[k] = live ? cc : 0.0;
        s_a0[k] = (in && q.pmode) ? q.a0[k] : 0.0;
        s_live[k] = live ? 1 : 0;
        s_b[k] = live ? ps - nk : 0.0;   // g (:284-292), the ride-along row
        s_dg[k] = live ? ps : 1.0;       // diagonal of H (:395-411) / of an identity row
        s_pos[k] = k;                    // (newton_tail indexes the direction through the sampled list: here by state)
        s_pv[k] = 1.0;
        if (sampled) pm = ps;
    }
    if (tid == 0) s_first[0] = first;
    pm = wave_max(pm);
    if (lane == 0) s_pmax[wave] = pm;
    __syncthreads();
    // pivots below eps * M * (largest per-state sum, which bounds the diagonal of H) count as zero like the singular values
    // numpy.linalg.lstsq drops (:582): the host path then takes the pseudo-inverse
    const double piv_thr = fmax(fmax(s_pmax[0], s_pmax[1]), fmax(s_pmax[2], s_pmax[3])) * 2.220446049250313e-16 * (double)(M > 0 ? M : 1);

    // ---- set-up: H = diag(psum) - c c^T G on the live states, identity elsewhere, g in row / column 0
    auto load_block = [&](int i, int j) -> v4d {  // (i <= j < nb, wave-uniform)
        const int b = i * nb - (i * (i - 1)) / 2 + (j - i);
        const int kr = 16 * i + R;
        const double ncr = -s_cl[kr];  // (zero for a row that is not live: the product below then is the zero of an identity row)
        v4d a;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = 4 * t + G, kc = 16 * j + c;
            const double gv = gram_lds ? gram_lds[b * (16 * SELECT_GRAM_PITCH) + R * SELECT_GRAM_PITCH + c] : q.gram_red[b * 256 + R * 16 + c];
            a[t] = (ncr * s_cl[kc]) * gv;
        }
        if (i == j) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (R == 4 * t + G) a[t] += s_dg[kr];
        }
        if (i == 0) {  // row 0 (and column 0 of block (0, 0)): the right-hand side
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 4 * t + G, kc = 16 * j + c;
                if (R == 0 && kc != 0) a[t] = s_b[kc];
                if (j == 0 && c == 0 && R != 0) a[t] = s_b[kr];
            }
        }
        return a;
    };
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    v4d Pn[NB - 1][2], Dm[2];
#pragma unroll
    for (int j = 1; j < NB; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int i = 4 * s + ((wave - j) & 3);
            Pn[j - 1][s] = (j < nb && i < j) ? load_block(i, j) : zero4;
        }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = wave + 4 * s;
        Dm[s] = j < nb ? load_block(j, j) : zero4;
    }
    __syncthreads();  // (the staged Gram blocks are dead from here: ws may be written)
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (wave + 4 * s == nb - 1) ldlt_write_block(ws + LDLT_DN_OFF, lane, Dm[s]);
    __syncthreads();
    if (st && tid == 0) st[2] = clock64();

    // ---- elimination, block column by block column from the last
#pragma unroll
    for (int kb = NB - 1; kb >= 0; --kb) {
        if (kb < nb) {
            long long tk0 = 0;
            if (st && tid == 0) tk0 = clock64();
            v4d D = ldlt_read_block(ws + LDLT_DN_OFF, lane);
            const int i0 = (wave - kb) & 3, i1 = 4 + i0;  // this wave's panel blocks of the block column
            const bool ok0 = i0 < kb, ok1 = i1 < kb;
            v4d RV = zero4;
#pragma unroll
            for (int p = 15; p >= (kb == 0 ? 1 : 0); --p) {
                const int t = p >> 2, g = p & 3;
                const double d = readlane_f64(D[t], 16 * g + p);
                const double nr = -recip_fast(d);
                const bool gs = G == g;
                const double X = (gs && R < p) ? D[t] : 0.0;  // the pivot row, left of the pivot; zero in the other lane groups,
                const double Xn = X * nr;                     // which is what confines the rank-1 products to K index g
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(X, Xn, D, 0, 0, 0);
                if (kb > 0) {  // (register t of a panel block over the pivot: its lane group g is column p)
                    if (ok0) Pn[kb > 0 ? kb - 1 : 0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xn, Pn[kb > 0 ? kb - 1 : 0][0][t], Pn[kb > 0 ? kb - 1 : 0][0], 0, 0, 0);
                    if (ok1) Pn[kb > 0 ? kb - 1 : 0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xn, Pn[kb > 0 ? kb - 1 : 0][1][t], Pn[kb > 0 ? kb - 1 : 0][1], 0, 0, 0);
                }
                RV[t] = gs ? nr : RV[t];
            }
            if (st && tid == 0) {
                const long long tk1 = clock64();
                st[6] += tk1 - tk0;
                tk0 = tk1;
            }
            // the pivots: the diagonal of the block (entry [p][p] is final once pivot p + 1 is done)
            if (wave == (kb & 3)) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (R == 4 * t + G) s_pv[16 * kb + R] = D[t];
            }
            // the frozen block column: V (this block column's updates only) and W = -V / d (kept for the substitution)
            if (kb > 0) {
                if (ok0) {
                    ldlt_write_block(ws + LDLT_V_OFF + i0 * LDLT_BLK, lane, Pn[kb > 0 ? kb - 1 : 0][0]);
                    ldlt_write_block(ws + LDLT_W_OFF + (kb * (kb - 1) / 2 + i0) * LDLT_BLK, lane, Pn[kb > 0 ? kb - 1 : 0][0] * RV);
                }
                if (ok1) {
                    ldlt_write_block(ws + LDLT_V_OFF + i1 * LDLT_BLK, lane, Pn[kb > 0 ? kb - 1 : 0][1]);
                    ldlt_write_block(ws + LDLT_W_OFF + (kb * (kb - 1) / 2 + i1) * LDLT_BLK, lane, Pn[kb > 0 ? kb - 1 : 0][1] * RV);
                }
            }
            if (wave == (kb & 3)) {
                v4d dw = D * RV;
#pragma unroll
                for (int t = 0; t < 4; ++t) dw[t] = R < 4 * t + G ? dw[t] : 0.0;  // entry [c][r], r < c: the pivot row of pivot c
                ldlt_write_block(ws + LDLT_DW_OFF + kb * LDLT_BLK, lane, dw);
            }
            if (kb > 0) {
                __syncthreads();
                // trailing matrix: B_ij += V_j W_i^T for this wave's blocks, the next diagonal block first
#pragma unroll
                for (int j = kb - 1; j >= 0; --j) {
                    const int a0 = (wave - j) & 3, a1 = 4 + a0;
                    const bool own_d = (j & 3) == wave, t0 = j > 0 && a0 < j, t1 = j > 0 && a1 < j;
                    if (own_d || t0 || t1) {
                        const v4d Vj = ldlt_read_block(ws + LDLT_V_OFF + j * LDLT_BLK, lane);
                        if (own_d) {
                            const v4d Wj = ldlt_read_block(ws + LDLT_W_OFF + (kb * (kb - 1) / 2 + j) * LDLT_BLK, lane);
                            ldlt_rank16(Dm[j >> 2], Vj, Wj);
                            if (j == kb - 1) ldlt_write_block(ws + LDLT_DN_OFF, lane, Dm[j >> 2]);
                        }
                        if (t0) {
                            const v4d Wi = ldlt_read_block(ws + LDLT_W_OFF + (kb * (kb - 1) / 2 + a0) * LDLT_BLK, lane);
                            ldlt_rank16(Pn[j > 0 ? j - 1 : 0][0], Vj, Wi);
                        }
                        if (t1) {
                            const v4d Wi = ldlt_read_block(ws + LDLT_W_OFF + (kb * (kb - 1) / 2 + a1) * LDLT_BLK, lane);
                            ldlt_rank16(Pn[j > 0 ? j - 1 : 0][1], Vj, Wi);
                        }
                    }
                }
                __syncthreads();
            }
            if (st && tid == 0) st[7] += clock64() - tk0;
        }
    }
    if (st && tid == 0) st[3] = clock64();

    // ---- x_P = sum_{r < P} W[r][P] x_r with x_0 = -1.  Wave w holds the unknowns of block rows 4 w .. 4 w + 3, one per lane;
    // a source block b is 16 sequential steps in which x_r leaves lane (b, r) through v_readlane and EVERY lane of the wave adds its
    // own weight times it -- the triangle of the block itself for the lanes of group b (zero from the diagonal on), the panel block
    // over (b, G) for the groups after it, zero before it -- so the four blocks of a wave need no barrier among themselves.
    __syncthreads();  // (the last diagonal block of W)
#pragma unroll
    for (int w2 = 0; w2 < (NB + 3) / 4; ++w2) {
        if (wave == w2 && 4 * w2 < nb) {
            const int kbP = 4 * w2 + G;
            const int off = (R >> 2) * 64 + 16 * (R & 3);  // the 16 weights of unknown p = R within a block: register R >> 2, lane group R & 3
            double wt[4][16];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int kb = 4 * w2 + b;
                const double* wp = G == b ? ws + LDLT_DW_OFF + kb * LDLT_BLK + off : ws + LDLT_W_OFF + (kbP * (kbP - 1) / 2 + kb) * LDLT_BLK + off;
#pragma unroll
                for (int r = 0; r < 16; ++r) wt[b][r] = (G >= b && kbP < nb) ? wp[r] : 0.0;  // (block rows from nb on: LDS nobody wrote)
            }
            double xv = (w2 == 0 && lane == 0) ? -1.0 : 0.0;
            if (kbP < nb) {
                for (int kb = 0; kb < 4 * w2; ++kb) {  // the block rows of the waves before this one
                    const double* wp = ws + LDLT_W_OFF + (kbP * (kbP - 1) / 2 + kb) * LDLT_BLK + off;
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv = fma(wp[r], s_x[16 * kb + r], xv);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (4 * w2 + b < nb) {  // (a source block past the last one would feed whatever its lanes hold into 0 * x)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv = fma(wt[b][r], readlane_f64(xv, 16 * b + r), xv);
                }
            }
            s_x[64 * w2 + lane] = xv;
        }
        __syncthreads();
    }
    // the gauge component (and the ride-along -1 of row 0) are zero in the direction
    if (tid == 0) {
        s_x[0] = 0.0;
        s_x[first] = 0.0;
    }
    __syncthreads();
    bool bad = false;
    if (tid < 128 && tid < q.Kp && s_live[tid]) {
        const double pvv = s_pv[tid];
        bad = !(pvv > piv_thr) || !isfinite(pvv);
    }
    if (st && tid == 0) st[4] = clock64();
    newton_tail<256>(q, s_x, bad, s_f, s_ps, s_nk, s_ln, s_a0, s_first, s_pos, tid);
}
// (amdgpu_waves_per_eu(2): at most 256 registers, for which the compiler takes the matrix instructions' VGPR form -- with the whole
// 512 it keeps the accumulators in AGPRs and copies every block to VGPRs and back around each pivot's vector instructions)
template <int NB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_newton_ldlt(AdaptArgs q) {
    extern __shared__ __attribute__((aligned(16))) double select_dyn_lds[];
    newton_body_ldlt<NB>(q, select_dyn_lds, nullptr);
}

// Fused loop: the selection of iteration i and the Newton solve of iteration i + 1 in ONE launch (a kernel boundary costs ~5 us;
// at the sizes pymbar is mostly used at that is a tenth of an iteration).  A stop or pause flag raised by the selection makes the
// solve return at once.
template <int R>
__global__ void __launch_bounds__(256)
k_select_newton(AdaptArgs q, int gram_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double select_dyn_lds[];
    long long* st = q.stamps ? q.stamps + 16 * (q.ctl[CTL_ITER] & 63) : nullptr;  // (debug: one slot per iteration)
    if (st && threadIdx.x == 0) st[0] = clock64();
    select_body(q, gram_in_lds ? select_dyn_lds : nullptr);
    __syncthreads();
    if (st && threadIdx.x == 0) st[1] = clock64();
    newton_body<16, R>(q, st);
    if (st && threadIdx.x == 0) st[5] = clock64();
}
template <int NB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_select_newton_ldlt(AdaptArgs q, int gram_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double select_dyn_lds[];
    long long* st = q.stamps ? q.stamps + 16 * (q.ctl[CTL_ITER] & 63) : nullptr;  // (debug: one slot per iteration)
    if (st && threadIdx.x == 0) st[0] = clock64();
    select_body(q, gram_in_lds ? select_dyn_lds : nullptr, st);
    __syncthreads();
    if (st && threadIdx.x == 0) st[1] = clock64();
    newton_body_ldlt<NB>(q, select_dyn_lds, gram_in_lds ? select_dyn_lds : nullptr, st);
    if (st && threadIdx.x == 0) st[5] = clock64();
}

__global__ void __launch_bounds__(256)
k_loop_reduce(LoopSrc src, int64_t count, int op, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double v = src.p[0][i];
    for (int r = 1; r < src.n; ++r) v = op == 0 ? v + src.p[r][i] : fmax(v, src.p[r][i]);
    out[i] = v;
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
    const int64_t n = (ntiles + tpc - 1) / tpc;
    return n < 1 ? 1 : n;  // (an empty shard still launches: its record is (max = -inf, sum = 0))
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
// dst_r = state_r - log(obs_r - shift_r) with shift_r as in k_rows_logshift, from the same partial minima: observables that are rows of
// a resident matrix, each at its own state, written straight into the rows of an extension context (mbar_ctx_rows_obs_from)
__global__ void __launch_bounds__(256) k_rows_obs(double* __restrict__ dst, const double* __restrict__ obs, const double* __restrict__ state,
                                                  int64_t ld, int64_t nrows, int64_t n, const double* __restrict__ part, int nparts,
                                                  double* __restrict__ shift_out) {
    __shared__ double red[4];
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y) {
        double m = INFINITY;
        for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = fmin(m, part[r * nparts + i]);
        m = -block256_max(-m, red);
        const double shift = m - fabs(8.881784197001252e-16 * m);  // 4 eps (mbar.py:827-832)
        if (blockIdx.x == 0 && threadIdx.x == 0) shift_out[r] = shift;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            dst[r * ld + i] = state[r * ld + i] - log(obs[r * ld + i] - shift);
        __syncthreads();
    }
}
// have_min: part[r] already holds the minimum of observable row r (one entry per row: the caller kept it from an earlier call on
// the same resident rows) -- the pass over the rows that finds it is skipped
hipError_t launch_rows_obs(hipStream_t s, double* dst, const double* obs, const double* state, int64_t ld, int64_t nrows, int64_t n,
                           double* part, double* shift_out, bool have_min) {
    const int64_t want = (n + 2047) / 2048;
    const unsigned gx = (unsigned)(want < 256 ? (want < 1 ? 1 : want) : 256);
    const unsigned gy = (unsigned)(nrows < 1024 ? (nrows < 1 ? 1 : nrows) : 1024);
    if (!have_min) hipLaunchKernelGGL(k_rows_min_partial, dim3(gx, gy), dim3(256), 0, s, obs, ld, nrows, n, part);
    hipLaunchKernelGGL(k_rows_obs, dim3(gx, gy), dim3(256), 0, s, dst, obs, state, ld, nrows, n, (const double*)part, have_min ? 1 : (int)gx,
                       shift_out);
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
    if (a.newton_ldlt) {
        const size_t lds = (size_t)LDLT_LDS_DOUBLES * sizeof(double);
        auto go = [&](auto kern) -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, s, a);
            return hipGetLastError();
        };
        return a.Kp <= 64 ? go(k_newton_ldlt<4>) : go(k_newton_ldlt<8>);
    }
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

hipError_t launch_ctl_resume(hipStream_t s, int* ctl) {
    hipLaunchKernelGGL(k_ctl_resume, dim3(1), dim3(64), 0, s, ctl);
    return hipGetLastError();
}

hipError_t launch_select_newton(hipStream_t s, const AdaptArgs& a) {
    const int M = a.m - 1;
    if (M > 127 || a.Kp > 128) return hipErrorInvalidValue;
    // full panel in the fused loop: the selection takes the second candidate's per-state sums from the reduced Gram blocks, staged in LDS
    const int stage = (a.fused && a.Kp == 128) ? 1 : 0;
    const size_t lds = a.newton_ldlt ? (size_t)LDLT_LDS_DOUBLES * sizeof(double) : stage ? (size_t)SELECT_GRAM_LDS_DOUBLES * sizeof(double) : 0;
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, s, a, stage);
        return hipGetLastError();
    };
    if (a.newton_ldlt) return a.Kp <= 64 ? go(k_select_newton_ldlt<4>) : go(k_select_newton_ldlt<8>);
    if (M <= 63) return go(k_select_newton<4>);
    return go(k_select_newton<8>);
}

hipError_t launch_select(hipStream_t s, const AdaptArgs& a) {
    if (a.Kp > 256) return hipErrorInvalidValue;
    const int stage = (a.fused && a.Kp == 128) ? 1 : 0;
    const size_t lds = stage ? (size_t)SELECT_GRAM_LDS_DOUBLES * sizeof(double) : 0;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_select, dim3(1), dim3(256), lds, s, a, stage);
    return hipGetLastError();
}
hipError_t launch_make_p(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                         const double* logden, double* P) {
    hipLaunchKernelGGL(k_make_p, dim3((unsigned)(num_cu * 8)), dim3(256), 0, s, u, ld, N, rows, aden, logden, P);
    return hipGetLastError();
}
// cw[n] = exp(p v[n]), cwsq[n] = exp(p v[n] / 2) for n < N: per-sample weights A'^p from the staging vector log A' of an observable
// (mbar_ctx_weights_from_vec; the padding behind N keeps its zeros)
__global__ void __launch_bounds__(256)
k_weights_from_log(const double* __restrict__ v, double p, int64_t n, double* __restrict__ cw, double* __restrict__ cwsq,
                   int* __restrict__ overflow) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = p * v[i];
        const double e = exp(x);
        cw[i] = e;
        cwsq[i] = exp(0.5 * x);
        bad |= !(e <= 1.79e308);  // inf (the observable spans more than 1e308^(1/p)) or NaN
    }
    if (bad) atomicOr(overflow, 1);
}
hipError_t launch_weights_from_log(hipStream_t s, const double* v, double p, int64_t n, double* cw, double* cwsq, int* overflow) {
    int64_t bx = (n + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(k_weights_from_log, dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0, s, v, p, n, cw, cwsq, overflow);
    return hipGetLastError();
}
// One bootstrap replicate as draw counts (mbar.py:417-449: every state redraws its N_k samples from its own samples): slot j of the
// state whose run of positions contains j draws one of that run's positions; the sample there gets one more count.
__global__ void __launch_bounds__(256)
k_bootstrap_counts(uint64_t seed, int64_t replicate, const int64_t* __restrict__ cum, int64_t K, int64_t total,
                   const int64_t* __restrict__ order, int64_t n0, int64_t N, double* __restrict__ cw) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = K;  // the state k with cum[k] <= j < cum[k + 1] (empty states have empty runs)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (cum[mid] <= j) lo = mid; else hi = mid;
        }
        const int64_t start = cum[lo], nk = cum[lo + 1] - start;
        const int64_t pos = start + bootstrap_draw(seed, (uint64_t)replicate, (uint64_t)j, (uint64_t)nk);
        const int64_t sample = (order ? order[pos] : pos) - n0;
        if (sample >= 0 && sample < N) atomicAdd(cw + sample, 1.0);
    }
}
hipError_t launch_bootstrap_counts(hipStream_t s, uint64_t seed, int64_t replicate, const int64_t* cum, int64_t K, int64_t total,
                                   const int64_t* order, int64_t n0, int64_t N, double* cw) {
    int64_t bx = (total + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_bootstrap_counts, dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0, s, seed, replicate, cum, K, total, order, n0, N, cw);
    return hipGetLastError();
}
hipError_t launch_zero(hipStream_t s, void* p, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 15) != 0 || ((uintptr_t)p & 15) != 0 || bytes < (size_t)1 << 16) return hipMemsetAsync(p, 0, bytes, s);
    const size_t n16 = bytes / 16;
    size_t bx = (n16 + 255) / 256;
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(k_zero16, dim3((unsigned)bx), dim3(256), 0, s, (uint4*)p, n16);
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

}  // namespace mbar
