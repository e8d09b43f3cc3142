// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- the fused sweep of the device-resident adaptive loop: k_fused.
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

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
// (Measured dead end, twice: the diagonal 16 x 16 blocks of the full panel as three v_mfma_f64_4x4x4_4b_f64 each -- 48 matrix-pipe
// cycles instead of 64, half of the 16x16x4 block lies below the diagonal.  Round 2, with DPP-rotated copies of the scaled operand:
// 1 % SLOWER (profiles/r2_fused_sweep_anatomy.txt, section 5).  Round 4, with NO vector instruction for it -- the per-sample scale
// on the A side only, G' = sum_n (P_n w_n / s_n^2) P_n^T, every B operand the raw tile entry, the rotated ones read from LDS through a
// rotated lane map -- parity-green and within +-1 % of the 16x16x4 blocks in three alternations on one box
// (profiles/r4_ab_fused_diag_subblocks_from_lds.txt): a 4x4x4 instruction costs the wave more than its 16 pipe cycles.)
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
template <int NB, bool WIDE>
__global__ void __launch_bounds__(256, 1)
k_fused(const double* __restrict__ P, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ cmul,
        const double* __restrict__ cw, const double* __restrict__ wsq, double* __restrict__ rinv0,
        double* __restrict__ rinv1, double* __restrict__ gram_part, double* __restrict__ psum_part,
        const int* __restrict__ ctl, int64_t slot_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl) {
        if (ctl[CTL_DONE] != 0 || ctl[CTL_LIGHT] != 0) return;  // (CTL_LIGHT: the last iteration needs no Gram matrix, see k_psweep)
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
            if (g == 2) wait_vm<1>();  // tile t_{i+1}, requested three quarters of an iteration ago
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
                    if (b < GRAM_AGPR_BLOCKS)
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(G[b]) : "v"(x), "v"(y));
                    else
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(G[b]) : "v"(x), "v"(y));
                } else {
                    G[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, G[b], 0, 0, 0);
                }
            };
            if constexpr (PINNED) {  // (asm MFMAs are opaque to the scheduler and the hazard recogniser: see k_gram)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
            }
            constexpr int NM = NBLK;  // 16x16x4 instructions per group
            int b = 0, nm = 0;
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                if (NB == 1 && g >= 2) {
                    mfma4(sacc[2 * (g & 1)], opa[0], opb[0], true);
                    mfma4(sacc[2 * (g & 1) + 1], opa[1], opb[1], true);
                }
#pragma unroll
                for (int J = I; J < NB; ++J) {
                    mfma(b, p[I], p[J]);
                    if constexpr (PINNED) {
                        if (nm == 0 || nm == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (nm == 0) request_next(); else request_steps();
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
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if defined(MBAR_GRAM_STORE_NT)  // (A/B build, profiles/r6_ab_fused_record_stores_nt.txt)
                    __builtin_nontemporal_store(G[b][r], &gram_part[((gw * NBLK + b) * 4 + r) * 64 + lane]);
#else
                    gram_part[((gw * NBLK + b) * 4 + r) * 64 + lane] = G[b][r];
#endif
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------

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
hipError_t launch_fused(hipStream_t s, int nb, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                        const double* cw, const double* wsq, double* rinv_base, double* gram_part, double* psum_part,
                        const LoopCtl& lc) {
    if (!lc.ctl) return hipErrorInvalidValue;  // (the slot vectors are addressed through the control words)
    if (nb == 12 || nb == 16) return launch_fused_quad(s, nb, g, P, ld, N, cmul, cw, wsq, rinv_base, gram_part, psum_part, lc);
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_fused_nb<NB_>(s, g, P, ld, N, cmul, cw, wsq, rinv_base, gram_part, psum_part, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7) MBAR_CASE(8)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mbar
