// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- fp64 MFMA Gram / Hessian sweep for up to 128 states per panel: k_gram.
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

// ---------------------------------------------------------------------------------------------
// Gram pass with known logden:  p = exp(anum_k - u_kn - logden_n)  (no cross-state dependency),
// acc[I][J] += p[I]^T p[J] on the fp64 matrix cores.  DIAG: one panel against itself, upper
// triangular blocks only; otherwise an NBI x NBJ rectangle between two panels.
// Output block b, register r, lane l  ->  element (row = (l >> 4) + 4 r, col = l & 15) of block b.
// ---------------------------------------------------------------------------------------------
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
// host-side launchers
// ---------------------------------------------------------------------------------------------

LaunchGeom gram_geometry(int tile_rows, bool diag, int num_cu, int64_t ntiles, int64_t grid_override) {
    LaunchGeom g;
    g.waves = 4;
    g.variant = -1;
    const size_t tile = (size_t)tile_rows * TS * 8 + TS * 8;  // u tile + its 16 logden values
    const bool one_buffer = tile_rows > 128 || (diag && tile_rows <= 80);  // must match NBUF in k_gram
    g.lds_bytes = (size_t)4 * (one_buffer ? 1 : 2) * tile + EXP_TABLE_BYTES;
    if (diag && tile_rows == 128) g.variant = 2;  // one wave per SIMD, pinned accumulator classes
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
    g.nwaves = g.blocks * 4;  // one partial record per wave
    g.psum_records = g.nwaves;
    return g;
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

hipError_t launch_gram_diag(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld,
                            int64_t N, const double* anum, const double* logden, int64_t row0, double* gp,
                            double* pp, const LoopCtl& lc) {
    if (lc.pmode) {  // resident probability matrix: `u` = P, `logden` = the reciprocals 1 / s_n
        switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_gram_t<NB_, NB_, true, true, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
            MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
            default: return hipErrorInvalidValue;
        }
    }
    if (nb == 8) {  // one wave per SIMD owns all 36 blocks (accumulator classes pinned by hand)
        if (lc.unclamped) return launch_gram_t<8, 8, true, true, false>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
        return launch_gram_t<8, 8, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
    }
    switch (nb) {
#define MBAR_CASE(NB_)                                                                                  \
    case NB_:                                                                                           \
        return launch_gram_t<NB_, NB_, true, true>(s, g, u, ld, N, anum, anum, logden, row0, row0, gp, pp, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gram_off(hipStream_t s, int nbj, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                           const double* ai, const double* aj, const double* logden, int64_t ri, int64_t rj,
                           double* gp, bool pmode) {
    if (pmode) {  // resident probability matrix: one multiplication per operand element instead of an exponential
        if (nbj == 8) return launch_gram_t<4, 8, false, true, true, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
        return launch_gram_t<4, 4, false, true, true, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
    }
    if (nbj == 8)  // 64 x 128 rectangle: 32 blocks, pinned accumulator classes, one 192-row tile buffer per wave
        return launch_gram_t<4, 8, false, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
    return launch_gram_t<4, 4, false, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
}

// 16 x 128 rectangle: ONE block row (the I panel: up to 16 states) against a full 128-state panel -- a few rows appended to a
// resident matrix against its states (mbar_gram_w_ext with the resident states' own W^T W supplied by the caller).  Every wave
// streams its own 144-row tiles; 8 blocks per k-step, so the sweep is bound by the operands' exponentials and by HBM, not by the
// matrix pipe.
hipError_t launch_gram_thin(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* ai,
                            const double* aj, const double* logden, int64_t ri, int64_t rj, double* gp) {
    return launch_gram_t<1, 8, false, true>(s, g, u, ld, N, ai, aj, logden, ri, rj, gp, nullptr);
}

}  // namespace mbar
