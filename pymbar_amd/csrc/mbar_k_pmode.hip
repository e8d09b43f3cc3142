// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- resident probability matrix: build sweeps (k_build_sweep, k_build_gram) and the evaluation sweep on it (k_psweep).
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

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

// out[n] = exp(ld0[n] - ld[n]) (* sqrt(cw[n]) when weighted): the reciprocal 1 / s_n of the P-mode Gram sweep for a point f whose
// log-denominators came from a sweep on u (the host-driven loop above 256 states): s_n = sum_k P_kn exp(a_k - a0_k) = exp(ld - ld0)
__global__ void __launch_bounds__(256)
k_rinv_from_logden(const double* __restrict__ ld0, const double* __restrict__ ldv, const double* __restrict__ cw, int weighted,
                   int64_t N, double* __restrict__ out) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        double r = exp(ld0[n] - ldv[n]);
        if (weighted) r *= sqrt(cw[n]);
        out[n] = r;
    }
}
hipError_t launch_rinv_from_logden(hipStream_t s, const double* ld0, const double* ldv, const double* cw, bool weighted, int64_t N,
                                   double* out) {
    int64_t bx = (N + 255) / 256;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_rinv_from_logden, dim3((unsigned)bx), dim3(256), 0, s, ld0, ldv, cw, weighted ? 1 : 0, N, out);
    return hipGetLastError();
}

// Build sweep of P mode: the single-candidate evaluation sweep at the anchor point a0 (log-sum-exp over states, per-state
// sums: the solver's initial gradient) that ALSO writes the normalised probabilities P_kn = e_kn / s_n.  They go back into
// the LDS tile in place of the energies they came from and leave with coalesced 16-byte stores that mirror the DMA
// pattern (8 lanes per 128-byte row), so the pass moves 8 K N bytes in and 8 K N out instead of the separate
// sweep + build (8 + 16).  The reciprocal slot of the anchor point is all ones.
// 16-byte store of two entries of P.  MBAR_P_STORE_NT (A/B build): with the non-temporal hint -- P is not read again before the
// next sweep, so its lines need not displace the tile stream in L2.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_p16(char* dst, const double2& v) {
#if defined(MBAR_P_STORE_NT)
    __builtin_nontemporal_store(v2d{v.x, v.y}, reinterpret_cast<v2d*>(dst));
#else
    *reinterpret_cast<double2*>(dst) = v;
#endif
}
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
            store_p16(dst, v);
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
// The kernel is bound by matrix + vector ISSUE on the pipe they share (9.2 k matrix cycles + the vector instructions of 512
// exponentials per tile and wave), so round 6 took vector instructions out of it -- per element 18 -> 14:
//   * the per-state sums are not accumulated: the rows of p sum to one, so sum_n c_n p_kn = sum_j G_kj of the Gram matrix this very
//     sweep accumulates (the host adds the rows of the reduced blocks: gram_row_sums);
//   * GENERAL = false (no sample multiplicities, no +inf entries in the matrix -- the first solve on a matrix): the operand IS the
//     probability (no multiplication by the root of the multiplicity; a sample beyond N gets the reciprocal 0 instead, which also
//     leaves an all-zero column in P's padding -- what the sweeps on P expect there), and the exponential needs no clamp;
//   * the exponential takes a non-negative argument "column maximum minus entry" (exp2s_neg_batch: fract + one conversion instead
//     of rint + subtract + convert).
template <int NB, bool WIDE, bool GENERAL>
__global__ void __launch_bounds__(256, 1)
k_build_gram(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ aden,
             const double* __restrict__ wsq, double* __restrict__ P, double* __restrict__ rinv_slot, double* __restrict__ gram_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int NDMA = ROWS / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + TS * 8;  // + the square roots of the tile's 16 sample multiplicities
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

    double aL[NB];  // a_k log2(e) S; states without samples / padding rows: a finite sentinel far below everything (their e is 0)
    {
        double a[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) a[I] = aden[16 * I + ks];
        if constexpr (NB <= 2) rows.live = live_piece_mask<NB>(a, -INFINITY);  // (narrow panels only: see k_gram)
#pragma unroll
        for (int I = 0; I < NB; ++I) aL[I] = a[I] > -INFINITY ? a[I] * LOG2E_S : -1.0e9;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) settle(aL[I]);
    v4d G[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) G[b] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = rd_base + ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage = [&](int64_t tile, char* dst) {
        stage_tile<ROWS, true, 0, 1>(u, ld, tile * TS, dst, lane, so, rows);
        stage_vec16<true>(wsq, tile * TS, dst + U_BYTES, lane);
    };
    int64_t t = gw;
    int cur = 0;
    if (t < ntiles) stage(t, buf);
    for (; t < ntiles; t += W) {
        char* cbuf = buf + cur * TILE_BYTES;
        const int64_t tn = t + W;
        if (tn < ntiles) {
            stage(tn, buf + (cur ^ 1) * TILE_BYTES);
            // vmcnt counts stores too, in issue order: [tile t: NDMA + 1][stores of tile t - W: NDMA + 1][tile tn: NDMA + 1]
            if (t != gw)
                wait_vm<2 * (NDMA + 1)>();
            else
                wait_vm<NDMA + 1>();
        } else {
            wait_vm<0>();
        }
        const int nvalid = (int)(N - t * TS < TS ? N - t * TS : TS);  // (wave-uniform: samples of this tile that exist)
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
        // Two groups of four samples side by side (round 6): a group's way from its LDS reads to its operands is a CHAIN -- tile
        // read, 16-lane maximum (DPP), table look-up, 16-lane sum (DPP), reciprocal -- whose latencies a lone wave cannot hide behind
        // anything but independent work of its own; the second group's chain is that work.  (Round 5 had no registers for it: 200
        // vector registers beside the 248 accumulator registers; the leaner exponential and the dropped per-state sums left 121.)
#pragma unroll
        for (int g = 0; g < GROUPS; g += 2) {
            double x0[NB], x1[NB];
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                x0[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g]);
                x1[I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                x0[I] = fma(x0[I], -LOG2E_S, aL[I]);   // (a_k - u_kn) log2(e) S
                x1[I] = fma(x1[I], -LOG2E_S, aL[I]);
            }
            double m0 = tree_max<NB>(x0), m1 = tree_max<NB>(x1);
            row16_max2(m0, m1);
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                x0[I] = m0 - x0[I];                      // >= 0, exactly 0 for the column's largest term
                x1[I] = m1 - x1[I];
            }
            exp2s_neg_batch2<NB, GENERAL>(x0, x1);
            double s0 = tree_sum<NB>(x0), s1 = tree_sum<NB>(x1);
            row16_sum2(s0, s1);
            double ri0 = recip_fast(s0), ri1 = recip_fast(s1);
            if (4 * g + ns >= nvalid) ri0 = 0.0;       // beyond N: an all-zero column of P, operand 0
            if (4 * g + 4 + ns >= nvalid) ri1 = 0.0;
            double p0[NB], p1[NB];
            if constexpr (GENERAL) {
                const double sw0 = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
                const double sw1 = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + 4 + ns) * 8);
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    x0[I] *= ri0;                          // P_kn
                    x1[I] *= ri1;
                    *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + pos[g]) = x0[I];
                    *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + pos[g + 1]) = x1[I];
                    p0[I] = x0[I] * sw0;                   // MFMA operand: the root of the multiplicity rides on both sides
                    p1[I] = x1[I] * sw1;
                }
            } else {
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    p0[I] = x0[I] * ri0;                   // P_kn = the MFMA operand
                    p1[I] = x1[I] * ri1;
                    *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + pos[g]) = p0[I];
                    *reinterpret_cast<double*>(cbuf + I * (16 * TS * 8) + pos[g + 1]) = p1[I];
                }
            }
            if constexpr (PINNED) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            {
                int b = 0;
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int J = I; J < NB; ++J) mfma(b++, p0[I], p0[J]);
            }
            {
                int b = 0;
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int J = I; J < NB; ++J) mfma(b++, p1[I], p1[J]);
            }
            if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
        }
        // the tile now holds P: out with it, 16 bytes per lane, 8 lanes per row (the LDS-DMA pattern backwards)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const double2 v = *reinterpret_cast<const double2*>(cbuf + j * 1024 + lane * 16);
            char* dst = reinterpret_cast<char*>(P + (int64_t)(8 * j) * ld + t * TS) + so.off[j & 1];
            store_p16(dst, v);
        }
        {
            const int64_t n = t * TS + lane;
            if (lane < TS && n < N) rinv_slot[n] = 1.0;
        }
        cur ^= 1;
    }
    if constexpr (PINNED) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
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
         double* __restrict__ psum_part, const int* __restrict__ ctl, int64_t slot_stride, int light_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {
        if (ctl[CTL_DONE] != 0) return;
        if (light_only && ctl[CTL_LIGHT] == 0) return;  // (fused loop: this sweep stands in for the fused one in the last iteration only)
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
// host-side launchers
// ---------------------------------------------------------------------------------------------

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
                                  ntiles, cmul, cw, rinv0, rinv1, pp, lc.ctl, lc.slot_stride, lc.light_only ? 1 : 0);
        else
            hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), g.lds_bytes, s, P, ld, N, ntiles, cmul, cw, rinv0,
                               rinv1, pp, lc.ctl, lc.slot_stride, lc.light_only ? 1 : 0);
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

// Build sweep + Gram matrix at the anchor: geometry of the fused sweep (same partial-record counts), tables in LDS.
LaunchGeom build_gram_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override) {
    // (same grid as the fused sweep, so that the partial-record counts agree; the kernel strides over tiles, so it does
    // not matter if the look-up tables leave room for one workgroup per CU less)
    LaunchGeom g = fused_geometry(nb, num_cu, ntiles, grid_override);
    g.lds_bytes = (size_t)4 * 2 * ((size_t)nb * 16 * TS * 8 + TS * 8) + EXP_TABLE_BYTES;  // (its own tile layout + the tables)
    return g;
}
template <int NB>
static hipError_t launch_build_gram_nb(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                       const double* aden, const double* wsq, bool general, double* P, double* rinv_slot, double* gp) {
    auto go = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, u, ld, N, ntiles, aden, wsq, P, rinv_slot, gp);
        return hipGetLastError();
    };
    if (general) return stage_offsets_wide(ld) ? go(k_build_gram<NB, true, true>) : go(k_build_gram<NB, false, true>);
    return stage_offsets_wide(ld) ? go(k_build_gram<NB, true, false>) : go(k_build_gram<NB, false, false>);
}
hipError_t launch_build_gram(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                             const double* aden, const double* wsq, bool general, double* P, double* rinv_slot, double* gram_part) {
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_build_gram_nb<NB_>(s, g, u, ld, N, aden, wsq, general, P, rinv_slot, gram_part);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mbar
