// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- one-read sweeps for 129 .. 256 states: k_gram_quad, k_fused_quad.
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

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
// Block of the last group behind which a wave waits for its LDS-DMA of the next tile and takes its rows into registers: the
// request went out behind the first blocks of group 0 (A/B: MBAR_QUAD_OWN_EARLY = behind block 1, three groups later)
#ifdef MBAR_QUAD_OWN_EARLY
#define QUAD_OWN_AT(n) 1
#else
#define QUAD_OWN_AT(n) ((n) - 4)
#endif
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
// SPLIT: the panel's rows come from TWO matrices of one row pitch -- rows [0, split_rows) from `u`, the rest from row_j0 on, counted
// in rows of `u` (an extension context's storage starts a whole number of row pitches away from its base matrix's:
// mbar_ctx_create_ext) -- so that rows appended to a resident matrix are swept with it without a copy of it.
template <int NBT, int WV, bool WIDE, bool PMODE, bool STOREP = false, int NBM = NBT, bool SPLIT = false>
__device__ __forceinline__ void gram_quad_body(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                               const double* __restrict__ anum, const double* __restrict__ logden,
                                               double* __restrict__ gram_part, char* smem, int lane, double* __restrict__ Pout = nullptr,
                                               int64_t split_rows = 0, int64_t row_j0 = 0) {
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
    auto rows = [&](int tr) -> int64_t { return (SPLIT && tr >= split_rows) ? row_j0 + (tr - split_rows) : (int64_t)tr; };
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

    // (The barriers of the tile loops in this file are lds_barrier(): LDS traffic only.  __syncthreads() also waits for every
    // outstanding global request -- here the reciprocal / probability-matrix stores a wave issued just before it, a full round trip
    // per tile that all four waves then stand in.)
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
        lds_barrier();  // every row of tile t holds operands; every wave is done with the other buffer
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
                        if (PREFETCH && g == GROUPS - 1 && mine == QUAD_OWN_AT(NMINE)) {  // the next tile was requested three groups ago: its rows into registers
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

template <int NBT, bool WIDE, int NBM = NBT>
__global__ void __launch_bounds__(256, 1)
k_gram_quad_split(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ anum,
                  const double* __restrict__ logden, double* __restrict__ gram_part, int64_t split_rows, int64_t row_j0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    exp_table_init(smem);
    __syncthreads();
    switch (wave) {
        case 0: gram_quad_body<NBT, 0, WIDE, false, false, NBM, true>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, nullptr, split_rows, row_j0); break;
        case 1: gram_quad_body<NBT, 1, WIDE, false, false, NBM, true>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, nullptr, split_rows, row_j0); break;
        case 2: gram_quad_body<NBT, 2, WIDE, false, false, NBM, true>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, nullptr, split_rows, row_j0); break;
        default: gram_quad_body<NBT, 3, WIDE, false, false, NBM, true>(u, ld, N, ntiles, anum, logden, gram_part, smem, lane, nullptr, split_rows, row_j0); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Rectangle between an NBI-block panel (64 / 128 states) and a 256-state panel on the resident probability matrix (P mode of the
// host-driven loop above 256 states, round 5): k_gram_quad's tile stream -- four waves, one shared tile of (NBI + 16) x 16 rows,
// each wave turns its quarter of the rows into operands P_kn / s_n in place, ONE barrier per tile -- and every wave issues the
// blocks (I, J) with J = WV mod 4: NBI + 4 operand reads for 4 NBI matrix instructions per k-step.  A 128 x 256 rectangle is
// 128 blocks for 384 staged rows (96 bytes per matrix instruction; the 64 x 128 rectangles of k_gram, one tile stream per wave,
// stage 192 rows for 32 blocks: 192 bytes, above the ~150 the matrix pipe can absorb at this HBM rate): above 256 states the Gram
// sweep reads the matrix 5.5 times at 1024 states instead of 11.5 (2.5 instead of 5.5 at 512).
// Record: one per workgroup, block b = I * 16 + J (what unpack_gram expects of a rectangle).
// ---------------------------------------------------------------------------------------------
template <int NBI, int NWV, int WV, bool WIDE>
__device__ __forceinline__ void gram_rect_body(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, int64_t row_i0,
                                               int64_t row_j0, const double* __restrict__ rinv, double* __restrict__ gram_part,
                                               char* smem, int lane) {
    // NWV waves (4: one per SIMD; 8: two per SIMD, 16 blocks each -- while one of a SIMD's two waves stands in the barrier or waits
    // for its first operands, the other's matrix instructions keep the pipe busy)
    constexpr int NBJ = 16, NBT = NBI + NBJ, ROWS = NBT * 16, NQ = NBT / NWV, QDMA = ROWS / NWV / 8, NJW = NBJ / NWV;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + NWV * 1024;  // + one copy of the tile's 16 reciprocals per wave
    constexpr int NBLK = NBI * NBJ, NMINE = NBLK / NWV, NP = NBI + NJW;
    static_assert(NBT % NWV == 0 && NBJ % NWV == 0 && NMINE >= QDMA + 2 && NMINE <= 32, "rectangle shape");
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem + EXP_TABLE_BYTES;
    const RowTwoPanels rows{row_i0, row_j0, NBI * 16};
    const StageOffsetsT<WIDE> so = make_stage_offsets<WIDE>(ld, lane);
    const int64_t G = gridDim.x;
    v4d acc[NMINE];
#pragma unroll
    for (int b = 0; b < NMINE; ++b) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;

    auto stage_piece_j = [&](int64_t tile, char* dst, int j) {
        stage_piece<true>(P + rows(8 * j) * ld + tile * TS, so.off[j & 1], dst + j * 1024, lane);
    };
    const char* lsrc = reinterpret_cast<const char*>(rinv) + (lane & 7) * 16;
    auto stage_l = [&](int64_t tile, char* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + tile * (TS * 8)),
                                         (__attribute__((address_space(3))) void*)(dst + U_BYTES + WV * 1024), 16, 0, 0);
    };
    auto stage = [&](int64_t tile, char* dst) {
#pragma unroll
        for (int j = WV * QDMA; j < (WV + 1) * QDMA; ++j) stage_piece_j(tile, dst, j);
        stage_l(tile, dst);
    };
    // operands of one group of four samples: the NBI row blocks of the short panel, then this wave's four column blocks
    auto read_group = [&](const char* tb, int g, double (&x)[NP]) {
#pragma unroll
        for (int I = 0; I < NBI; ++I) x[I] = *reinterpret_cast<const double*>(tb + I * (16 * TS * 8) + rd_base + pos[g]);
#pragma unroll
        for (int q = 0; q < NJW; ++q) x[NBI + q] = *reinterpret_cast<const double*>(tb + (NBI + WV + NWV * q) * (16 * TS * 8) + rd_base + pos[g]);
    };
    auto mfma = [&](int b, double x, double y) {
        if (b < GRAM_AGPR_BLOCKS)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(x), "v"(y));
        else
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(x), "v"(y));
    };
    double x[GROUPS * NQ], ldc[GROUPS];
    auto read_own = [&](const char* tb) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            ldc[g] = *reinterpret_cast<const double*>(tb + U_BYTES + WV * 1024 + (4 * g + ns) * 8);
#pragma unroll
            for (int i = 0; i < NQ; ++i) x[g * NQ + i] = *reinterpret_cast<const double*>(tb + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]);
        }
    };
    int64_t t = blockIdx.x;
    int cur = 0;
    if (t < ntiles) {
        stage(t, buf);
        wait_vm<0>();
        read_own(buf);
    }
#ifdef MBAR_RECT_STAMPS
    long long st_conv = 0, st_bar = 0, st_rd = 0, st_mf = 0, st_n = 0;
#define RECT_STAMP(v) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); v = clock64(); } while (0)
#else
#define RECT_STAMP(v) do { } while (0)
#endif
    for (; t < ntiles; t += G) {
        [[maybe_unused]] long long s0, s1, s2, s3, s4;
        RECT_STAMP(s0);
        char* cbuf = buf + cur * TILE_BYTES;
        char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
        const int64_t tnext = t + G < ntiles ? t + G : t;  // (past the end this tile is requested again and never looked at)
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const double rin = (t * TS + 4 * g + ns) < N ? ldc[g] : 0.0;  // 1 / s_n (times sqrt(c_n) when weighted); padded samples: 0
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                *reinterpret_cast<double*>(cbuf + (WV * NQ + i) * (16 * TS * 8) + rd_base + pos[g]) = x[g * NQ + i] * rin;
        }
        RECT_STAMP(s1);
        lds_barrier();  // every row of tile t holds operands; every wave is done with the other buffer
        RECT_STAMP(s2);
        double p[2][NP];
        read_group(cbuf, 0, p[0]);
        RECT_STAMP(s3);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7");
#pragma unroll
            for (int I = 0; I < NBI; ++I)
#pragma unroll
                for (int q = 0; q < NJW; ++q) {
                    const int mine = I * NJW + q;
                    mfma(mine, p[g & 1][I], p[g & 1][NBI + q]);
                    if (mine == 0 && g < GROUPS - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        read_group(cbuf, g + 1, p[(g + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (g == 0 && mine >= 1 && mine <= QDMA + 1) {  // one piece of the next tile behind each of the next blocks
                        __builtin_amdgcn_sched_barrier(0);
                        if (mine <= QDMA)
                            stage_piece_j(tnext, nbuf, WV * QDMA + mine - 1);
                        else
                            stage_l(tnext, nbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (g == GROUPS - 1 && mine == QUAD_OWN_AT(NMINE)) {  // the next tile was requested three groups ago: its rows into registers
                        __builtin_amdgcn_sched_barrier(0);
                        wait_vm<0>();
                        read_own(nbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        RECT_STAMP(s4);
#ifdef MBAR_RECT_STAMPS
        st_conv += s1 - s0; st_bar += s2 - s1; st_rd += s3 - s2; st_mf += s4 - s3; ++st_n;
#endif
        cur ^= 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // matrix result -> VALU read distance
#ifdef MBAR_RECT_STAMPS
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
        printf("[rect stamps] block %d wave %d: %lld tiles, per tile: operands %lld, barrier %lld, first operands %lld, blocks %lld clocks\n",
               (int)blockIdx.x, WV, st_n, st_conv / st_n, st_bar / st_n, st_rd / st_n, st_mf / st_n);
#endif
    double* rec = gram_part + (int64_t)blockIdx.x * NBLK * 256;
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int q = 0; q < NJW; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[((I * NBJ + WV + NWV * q) * 4 + r) * 64 + lane] = acc[I * NJW + q][r];
}

template <int NBI, int NWV, bool WIDE>
__global__ void __launch_bounds__(64 * NWV, 1)
k_gram_rect(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, int64_t row_i0, int64_t row_j0,
            const double* __restrict__ rinv, double* __restrict__ gram_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define MBAR_CASE(W_) case W_: gram_rect_body<NBI, NWV, W_, WIDE>(P, ld, N, ntiles, row_i0, row_j0, rinv, gram_part, smem, lane); break;
    if constexpr (NWV == 8) {
        switch (wave) { MBAR_CASE(0) MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) default: gram_rect_body<NBI, NWV, 7, WIDE>(P, ld, N, ntiles, row_i0, row_j0, rinv, gram_part, smem, lane); }
    } else {
        switch (wave) { MBAR_CASE(0) MBAR_CASE(1) MBAR_CASE(2) default: gram_rect_body<NBI, NWV, 3, WIDE>(P, ld, N, ntiles, row_i0, row_j0, rinv, gram_part, smem, lane); }
    }
#undef MBAR_CASE
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
        lds_barrier();  // (the four partial sums of every sample are in the table; every wave is done with the other buffer)
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
        lds_barrier();  // every row of tile t holds operands
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
                        if (g == GROUPS - 1 && mine == QUAD_OWN_AT(NMINE)) {
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

// The last iteration of a converging solve (CTL_LIGHT, mbar_internal.h): nobody will read the Gram matrix, so the sweep is the
// evaluation half of fused_quad_body alone -- normalisers, reciprocals and per-state sums of both candidates, same tile stream,
// same partial records -- and HBM-bound (one 24- / 32-KB tile in flight per CU) instead of matrix-bound.  A body of its own: the
// full one keeps its hand-placed instruction stream untouched.
template <int NBT, int WV, bool WIDE, int NBM = NBT>
__device__ __forceinline__ void fused_quad_light_body(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles,
                                                      const double* __restrict__ cmul, const double* __restrict__ cw,
                                                      const double* __restrict__ wsq, double* __restrict__ rinv0,
                                                      double* __restrict__ rinv1, double* __restrict__ psum_part, char* smem, int lane) {
    constexpr int ROWS = NBT * 16, NQ = NBT / 4, QDMA = ROWS / 4 / 8;
    constexpr int U_BYTES = ROWS * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + 4 * 1024;
    const int ks = lane & 15, ns = lane >> 4;
    char* buf = smem;
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
    const int rd_base = ks * (TS * 8);
    int pos[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) pos[g] = ((4 * g + ns + (ks & 14)) & 15) * 8;
    const char* wsrc = reinterpret_cast<const char*>(((lane >> 3) & 1) ? wsq : cw) + (lane & 7) * 16;
    auto stage = [&](int64_t tile, char* dst) {  // this wave's quarter of the rows + its copy of the multiplicities
#pragma unroll
        for (int j = WV * QDMA; j < (WV + 1) * QDMA; ++j)
            if (j < 2 * NBM) stage_piece<true>(P + rows(8 * j) * ld + tile * TS, so.off[j & 1], dst + j * 1024, lane);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + tile * (TS * 8)),
                                         (__attribute__((address_space(3))) void*)(dst + U_BYTES + WV * 1024), 16, 0, 0);
    };
    double x[GROUPS * NQ], w[GROUPS];
    auto read_own = [&](const char* tb) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            w[g] = *reinterpret_cast<const double*>(tb + U_BYTES + WV * 1024 + (4 * g + ns) * 8);
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
        char* nbuf = buf + (cur ^ 1) * TILE_BYTES;
        const int64_t tnext = t + G < ntiles ? t + G : t;
        stage(tnext, nbuf);  // (this wave's rows of the other buffer: read last at the end of the previous iteration, by this wave)
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
        lds_barrier();
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int sidx = 4 * g + ns;
            const double s0 = (xs[(0 * 2 + 0) * TS + sidx] + xs[(1 * 2 + 0) * TS + sidx]) + (xs[(2 * 2 + 0) * TS + sidx] + xs[(3 * 2 + 0) * TS + sidx]);
            const double s1 = (xs[(0 * 2 + 1) * TS + sidx] + xs[(1 * 2 + 1) * TS + sidx]) + (xs[(2 * 2 + 1) * TS + sidx] + xs[(3 * 2 + 1) * TS + sidx]);
            const double r0 = recip_fast(fmax(s0, 1e-300)), r1 = recip_fast(fmax(s1, 1e-300));
            const double q0 = w[g] * r0, q1 = w[g] * r1;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (WV * NQ + i >= NBM) continue;
                acc0[i] = fma(x[g * NQ + i], q0, acc0[i]);
                acc1[i] = fma(x[g * NQ + i], q1, acc1[i]);
            }
            if (g == WV) {
                const int64_t n = t * TS + sidx;
                if (n < N && ks < 2) (ks == 0 ? rinv0 : rinv1)[n] = ks == 0 ? r0 : r1;
            }
        }
        lds_barrier();  // (the table of partial normalisers is free again)
        wait_vm<0>();
        read_own(nbuf);
        cur ^= 1;
    }
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
    if (ctl[CTL_LIGHT] != 0) {  // the last iteration of a converging solve: no Gram matrix (its records are left alone)
        switch (wave) {
            case 0: fused_quad_light_body<NBT, 0, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, psum_part, smem, lane); break;
            case 1: fused_quad_light_body<NBT, 1, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, psum_part, smem, lane); break;
            case 2: fused_quad_light_body<NBT, 2, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, psum_part, smem, lane); break;
            default: fused_quad_light_body<NBT, 3, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, psum_part, smem, lane); break;
        }
        return;
    }
    switch (wave) {
        case 0: fused_quad_body<NBT, 0, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        case 1: fused_quad_body<NBT, 1, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        case 2: fused_quad_body<NBT, 2, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
        default: fused_quad_body<NBT, 3, WIDE, NBM>(P, ld, N, ntiles, cmul, cw, wsq, rinv0, rinv1, gram_part, psum_part, smem, lane); break;
    }
}


// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------

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
// One-read Gram sweep of a 192- / 256-row panel whose rows [split_rows, 16 nbt) live in another matrix of the same row pitch
// (row_j0: their first row, counted in rows of `u`; classic operands only).  live_blocks as in launch_gram_quad.
hipError_t launch_gram_quad_split(hipStream_t s, int nbt, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* anum,
                                  const double* logden, double* gp, int64_t split_rows, int64_t row_j0) {
    auto launch = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(256), g.lds_bytes, s, u, ld, N, ntiles, anum, logden, gp, split_rows, row_j0);
        return hipGetLastError();
    };
    if (split_rows < 8 || split_rows % 8 != 0 || split_rows >= 16 * nbt) return hipErrorInvalidValue;
    const bool wide = stage_offsets_wide(ld);
    const bool trim = g.live_blocks > 0 && g.live_blocks <= nbt - 2;
    if (nbt == 12) {
        if (trim) return wide ? launch(k_gram_quad_split<12, true, 10>) : launch(k_gram_quad_split<12, false, 10>);
        return wide ? launch(k_gram_quad_split<12, true>) : launch(k_gram_quad_split<12, false>);
    }
    if (nbt == 16) {
        if (trim) return wide ? launch(k_gram_quad_split<16, true, 14>) : launch(k_gram_quad_split<16, false, 14>);
        return wide ? launch(k_gram_quad_split<16, true>) : launch(k_gram_quad_split<16, false>);
    }
    return hipErrorInvalidValue;
}
// (the geometry is gram_quad_geometry(nbi + 16, ...): the same shared double buffer, one record per workgroup)
hipError_t launch_gram_rect(hipStream_t s, int nbi, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, int64_t ri, int64_t rj,
                            const double* rinv, double* gram_part) {
    auto launch = [&](auto kern) -> hipError_t {
        if (g.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TS - 1) / TS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(64 * g.waves), g.lds_bytes, s, P, ld, N, ntiles, ri, rj, rinv, gram_part);
        return hipGetLastError();
    };
    const bool wide = stage_offsets_wide(ld);
    if (nbi == 8 && g.waves == 8) return wide ? launch(k_gram_rect<8, 8, true>) : launch(k_gram_rect<8, 8, false>);
    if (nbi == 8) return wide ? launch(k_gram_rect<8, 4, true>) : launch(k_gram_rect<8, 4, false>);
    if (nbi == 4) return wide ? launch(k_gram_rect<4, 4, true>) : launch(k_gram_rect<4, 4, false>);
    return hipErrorInvalidValue;
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
hipError_t launch_fused_quad(hipStream_t s, int nbt, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                             const double* cw, const double* wsq, double* rinv0, double* gp, double* pp, const LoopCtl& lc) {
    if (nbt == 12) return launch_fused_quad_t<12>(s, g, P, ld, N, cmul, cw, wsq, rinv0, gp, pp, lc);
    if (nbt == 16) return launch_fused_quad_t<16>(s, g, P, ld, N, cmul, cw, wsq, rinv0, gp, pp, lc);
    return hipErrorInvalidValue;
}

}  // namespace mbar
