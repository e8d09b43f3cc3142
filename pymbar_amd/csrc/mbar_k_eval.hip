// gfx950 (CDNA4 / MI355X) kernels of the MBAR solver hot path -- evaluation sweeps (log-sum-exp over states + per-state sums): k_lse, k_lse_small, k_lse_wide, k_lse_split, layout-agnostic fallbacks.
// One of the translation units of libmbar_hip.so (compiled in parallel by pymbar_amd/_build.py): the shared device helpers and
// the data-layout notes are in mbar_device.h, the host-side interface of the launchers in mbar_internal.h.
#include "mbar_device.h"

namespace mbar {

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
// (the sweep proper: shared by k_lse_small and by k_sci_small, whose prologue supplies `aden` from LDS; the look-up tables are
// in place and the workgroup has passed a barrier when this is called)
// The wave's tile request (LDS-DMA instruction j moves rows 2j, 2j+1: lane l -> row 2j + (l >> 5), bytes [16 (l & 31), +16) of
// its 512; pairs of rows whose bit in `live` is clear are requested from the first tile's columns -- an L2 hit, see below).
template <int NB>
__device__ __forceinline__ void lse_small_stage(const double* __restrict__ u, int64_t ld, const double* __restrict__ cw, int64_t tile,
                                                uint32_t live, char* buf, int lane) {
    constexpr int ROWS = NB * 16;
    const uint32_t voff = (uint32_t)(((int64_t)(lane >> 5) * ld + 2 * (lane & 31)) * 8);
#pragma unroll
    for (int j = 0; j < ROWS / 2; ++j)
        stage_piece<true>(u + (int64_t)(2 * j) * ld + (((live >> j) & 1u) ? tile * TSS : 0), voff, buf + j * 1024, lane);
    if (lane < 32) stage_piece<true>(cw + tile * TSS, (uint32_t)(lane * 16), buf + ROWS * TSS * 8, lane);
}
__device__ __forceinline__ int64_t lse_small_first_tile(int balanced /*0 or 1*/, int wave, int nwv) {
    // (wave-uniform: it enters the scalar base of every DMA address)
    const int mb = nwv - balanced * (nwv - 1), mw = 1 + balanced * ((int)gridDim.x - 1);
    int first = __builtin_amdgcn_readfirstlane((int)blockIdx.x * mb + wave * mw);
    asm volatile("" : "+s"(first));
    return (int64_t)first;
}
// One partial record per WORKGROUP from the per-lane sums of its eight waves, in a FIXED order: every wave writes its registers
// into its own (now idle) tile buffer, [state][lane]; after one barrier thread (k, s) = (tid / 16, tid % 16) adds the 32 values of
// state k in lanes 32 (s & 1) .. + 31 of wave s / 2 (the start rotated by the thread index: every bank busy, the order still a
// function of the thread alone), and a 16-lane DPP reduction finishes the state.  (Until round 5 every wave reduced each of its
// 33 registers by a six-step ds_bpermute butterfly, one after the other: 9.5 us of a 57 us launch at config 2.)
template <int ROWS>
__device__ __forceinline__ void lse_small_fold(char* smem, const double (&acc)[ROWS], double objl, double* __restrict__ psum_part,
                                               double* __restrict__ obj_part, long long* st) {
    constexpr int TILE_BYTES = ROWS * TSS * 8 + TSS * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* mine = reinterpret_cast<double*>(smem + EXP_TABLE_BYTES + wave * TILE_BYTES);
#pragma unroll
    for (int k = 0; k < ROWS; ++k) mine[k * 64 + lane] = acc[k];
    if (obj_part) mine[ROWS * 64 + lane] = objl;
    __syncthreads();
    if (st) st[5] = wall_clock64();
    {
        const int k = tid >> 4, sg = tid & 15;
        if (k < ROWS) {  // (ROWS = 16: the upper half of the workgroup idles)
            const double* src = reinterpret_cast<const double*>(smem + EXP_TABLE_BYTES + (sg >> 1) * TILE_BYTES) + k * 64 + (sg & 1) * 32;
            double v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = src[(j + tid) & 31];
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                s0 += v[j];
                s1 += v[j + 1];
                s2 += v[j + 2];
                s3 += v[j + 3];
            }
            const double tot = row16_sum((s0 + s1) + (s2 + s3));
            if (sg == 0) psum_part[(int64_t)blockIdx.x * ROWS + k] = tot;
        }
    }
    if (obj_part && tid < 64) {  // the objective terms: 512 values, eight per lane of wave 0
        double o = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) o += reinterpret_cast<const double*>(smem + EXP_TABLE_BYTES + w * TILE_BYTES)[ROWS * 64 + lane];
        o = wave_sum(o);
        if (lane == 0) obj_part[blockIdx.x] = o;
    }
    if (st) st[6] = wall_clock64();
}
// PRESTAGED: the caller has already requested the wave's first tile (with the same `live_in` mask) before its own prologue.
template <int NB, bool PRESTAGED = false>
__device__ __forceinline__ void lse_small_body(char* smem, const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
                                               const double* aden, const double* __restrict__ cw, double* __restrict__ logden0,
                                               const double* __restrict__ dn, double* __restrict__ psum_part,
                                               double* __restrict__ obj_part, uint32_t live_in = 0, int balanced = 0, int rev = 0, long long* st = nullptr, long long* st_waves = nullptr) {
    constexpr int ROWS = NB * 16;
    constexpr int U_BYTES = ROWS * TSS * 8;
    constexpr int TILE_BYTES = U_BYTES + TSS * 8;  // + the 64 sample weights of the tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    char* buf = smem + EXP_TABLE_BYTES + wave * TILE_BYTES;
    // first tile of the wave's stream (stride W): wave-major, or -- `balanced` -- workgroup-major, which spreads the streams that are
    // one tile longer evenly over the compute units (config 2: 15 625 tiles over 2048 waves; wave-major gives 161 units 64 tiles
    // and 95 units 56, workgroup-major 61 or 62 each)
    const int64_t gw = lse_small_first_tile(balanced, wave, nwv);
    const int64_t W = (int64_t)gridDim.x * nwv;
    const bool has_store = logden0 != nullptr;

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
    uint32_t live = live_in;
    if constexpr (!PRESTAGED) {
#pragma unroll
        for (int j = 0; j < ROWS / 2; ++j) live |= (__ballot(a[2 * j] != -INFINITY || a[2 * j + 1] != -INFINITY) ? 1u : 0u) << j;
    }
    // rev (0 / 1): the stream positions map to the tiles in DESCENDING order -- a loop that sweeps the same matrix again and again
    // (the self-consistent iteration) alternates the direction, so that a sweep begins with what the previous one left in the
    // caches (L2, Infinity Cache) instead of evicting it just before it is needed; plain scalar arithmetic, see lse_small_first_tile
    const int64_t rsgn = 1 - 2 * rev, roff = rev * (ntiles - 1);
    auto stage = [&](int64_t pos) { lse_small_stage<NB>(u, ld, cw, roff + rsgn * pos, live, buf, lane); };

    int64_t t = gw;
    if constexpr (!PRESTAGED) {
        if (t < ntiles) stage(t);
    }
    for (; t < ntiles; t += W) {
        if (has_store && t != gw)
            wait_vm<1>();  // [this tile][logden store of the previous one]: vmcnt counts stores too
        else
            wait_vm<0>();
        if (st && t == gw) st[3] = wall_clock64();
#ifdef MBAR_SMALL_SETPRIO
        // the two waves of a SIMD take turns at the higher issue priority (otherwise the older one always wins a tie: waves 0-3
        // of a workgroup ran 16 % ahead of waves 4-7)
        if (((t - gw) / W) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
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
        const int64_t n = (roff + rsgn * t) * TSS + lane;
        if (n < N) {
            if (logden0) logden0[n] = ldv;
            objl = fma(w, dn ? (ldv - dn[n]) : ldv, objl);
        }
    }
    if (st) st[4] = wall_clock64();
    if (st_waves && lane == 0) st_waves[wave] = wall_clock64();
    lse_small_fold<ROWS>(smem, acc, objl, psum_part, obj_part, st);
}
template <int NB, int BAL>
__global__ void __launch_bounds__(512, 2)
k_lse_small(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles,
            const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden0,
            const double* __restrict__ dn, double* __restrict__ psum_part, double* __restrict__ obj_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int TILE_BYTES = ROWS * TSS * 8 + TSS * 8;
    // which pairs of rows matter (see lse_small_body), then the wave's first tile is requested before the tables are copied
    uint32_t live = 0;
#pragma unroll
    for (int j = 0; j < ROWS / 2; ++j) live |= ((aden[2 * j] != -INFINITY || aden[2 * j + 1] != -INFINITY) ? 1u : 0u) << j;
    live = __builtin_amdgcn_readfirstlane(live);
    {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int64_t gw = lse_small_first_tile(BAL, wave, blockDim.x >> 6);
        exp_table_init_dma(smem);  // (ahead of the tile in the memory queue)
        if (gw < ntiles) lse_small_stage<NB>(u, ld, cw, gw, live, smem + EXP_TABLE_BYTES + wave * TILE_BYTES, lane);
    }
    wait_vm<0>();
    __syncthreads();
    lse_small_body<NB, true>(smem, u, ld, N, ntiles, aden, cw, logden0, dn, psum_part, obj_part, live, BAL);
}

// ---------------------------------------------------------------------------------------------
// One self-consistent iteration (mbar_solvers.py:231-242 in a loop) of a few-state problem in ONE launch: the update that
// turns the previous sweep's per-state sums into the next f rides in the PROLOGUE of the sweep that evaluates it.  Every
// workgroup sums the previous launch's partial records (one per workgroup, a few hundred) in the same fixed order -- identical
// inputs, identical bits, no broadcast -- forms f' = f - log(psum / N) with the gauge f'[first] = 0 (:588) and a' = f' + ln N in
// LDS, and then runs k_lse_small's sweep at a'.  Workgroup 0 also publishes f', the relative change (:627-631) and the history
// row.  Records and the state vector are double-buffered by the parity of the iteration: a workgroup that is still in its
// prologue reads what the previous launch wrote while an early one already writes this launch's record.
// At config 2 (K=32, N=1e6: a 52 us sweep) the separate single-workgroup update kernel + its launch gap were ~9 us per iteration.
// ---------------------------------------------------------------------------------------------
template <int NB>
__global__ void __launch_bounds__(512, 2)
k_sci_small(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, const double* __restrict__ cw, SciLoopArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = NB * 16;
    constexpr int G = 512 / ROWS;  // groups of threads that share the record sum of one state
    constexpr int TILE_BYTES = ROWS * TSS * 8 + TSS * 8;
    // (debug, MBAR_DEBUG_STAMPS: 100 MHz stamps of thread 0 of workgroups 0 and gridDim.x / 2, eight per workgroup)
    long long* st = (q.stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) ? q.stamps + (blockIdx.x == 0 ? 0 : 8) : nullptr;
    if (st) st[0] = wall_clock64();
    // Order of the requests = order of arrival: the previous sweep's records first (a few KB: they gate the update), then the
    // look-up tables (LDS-DMA), then the wave's first tile -- which lands while the update is computed.  (Until round 5 the tile
    // went first: the tables queued behind it for 5.7 us, the record loads behind the tables for another 4.)
    const int tid = threadIdx.x;
    const int par = q.parity;
    const double* rprev = q.rec + (int64_t)(par ^ 1) * q.nrec * ROWS;
    const double* fprev = q.state + (int64_t)(par ^ 1) * ROWS;
    constexpr int RMAX = 16;
    const bool direct = q.nrec <= RMAX * G;  // (the usual case: at most 256 records)
    double rv[RMAX];
    {
        const int k = tid % ROWS, g = tid / ROWS;
#pragma unroll
        for (int j = 0; j < RMAX; ++j) {
            const int64_t p = g + (int64_t)j * G;  // (unconditional loads of a clamped index: a predicated load becomes a branch)
            rv[j] = load_untracked(rprev + (p < q.nrec ? p : q.nrec - 1) * ROWS + k);
        }
    }
    // (everything else the update reads from memory, also ahead of the requests it must not wait for)
    const int kq = tid < ROWS ? tid : 0;
    double fo_r = load_untracked(fprev + kq), nk_r = load_untracked(q.Nk + kq), ln_r = load_untracked(q.lnNk + kq);
    double f1_r = load_untracked(fprev + q.first), nk1_r = load_untracked(q.Nk + q.first);
    exp_table_init_dma(smem);
    {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int64_t gw = lse_small_first_tile(q.balanced != 0 ? 1 : 0, wave, blockDim.x >> 6);
        const int rev = q.pingpong ? (q.parity & 1) : 0;
        // (the loads above are waited for by hand: not for the table and tile requests issued behind them)
        if (gw < ntiles) {
            lse_small_stage<NB>(u, ld, cw, rev * (ntiles - 1) + (1 - 2 * rev) * gw, q.live, smem + EXP_TABLE_BYTES + wave * TILE_BYTES, lane);
            wait_vm_for<EXP_TABLE_DMA_PER_WAVE + ROWS / 2 + 1>(fo_r, nk_r, ln_r, f1_r, nk1_r, rv);
        } else {
            wait_vm_for<EXP_TABLE_DMA_PER_WAVE>(fo_r, nk_r, ln_r, f1_r, nk1_r, rv);
        }
    }
    if (st) st[1] = wall_clock64();
    double* scr = reinterpret_cast<double*>(smem + EXP_TABLE_BYTES + 8 * TILE_BYTES);  // (behind the eight waves' tile buffers)
    double* ps = scr + 512;        // [ROWS] reduced per-state sums
    double* a_s = ps + ROWS;       // [ROWS] a' = f' + ln N (-inf: no samples / padding)
    double* fn_s = a_s + ROWS;     // [ROWS] f'
    double* red = fn_s + ROWS;     // [2]: gauge shift, max relative change
    {
        const int k = tid % ROWS, g = tid / ROWS;
        double sm = 0.0;
        if (direct) {
#pragma unroll
            for (int j = 0; j < RMAX; ++j) sm += (g + (int64_t)j * G < q.nrec) ? rv[j] : 0.0;  // (the order of the loop below; zeros past the end change nothing)
        } else {
#pragma unroll 8  // (same order in every workgroup; eight record loads in flight together)
            for (int64_t p = g; p < q.nrec; p += G) sm += rprev[p * ROWS + k];
        }
        scr[g * ROWS + k] = sm;
    }
    lds_barrier();
    if (tid < ROWS) {
        double tot = 0.0;
        for (int g = 0; g < G; ++g) tot += scr[g * ROWS + tid];
        ps[tid] = tot;
    }
    lds_barrier();
    if (tid == 0) red[0] = f1_r - log(ps[q.first] / nk1_r);
    lds_barrier();
    if (tid < 64) {  // one wave: states tid (and tid + 64 would not exist: ROWS <= 32)
        const double small = q.tol < 1e-8 ? q.tol : 1e-8;
        double d = 0.0;
        if (tid < ROWS) {
            const int k = tid;
            const bool sampled = k < q.K && nk_r > 0.0;
            const double fo = fo_r;
            double fnew = fo, an = -INFINITY;
            if (sampled) {
                fnew = fo - log(ps[k] / nk_r) - red[0];
                an = fnew + ln_r;
                if (k != q.first) {
                    const double div = fabs(fnew) < small ? 1.0 : fabs(fnew);
                    d = fabs(fnew - fo) / div;
                }
            }
            fn_s[k] = fnew;
            a_s[k] = an;
        }
        // NaN-propagating maximum over the wave
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const double o = __shfl_xor(d, sft);
            d = (o > d || o != o) ? o : d;
        }
        if (tid == 0) red[1] = d;
    }
    lds_barrier();
    if (blockIdx.x == 0 && tid < ROWS) {
        q.state[(int64_t)par * ROWS + tid] = fn_s[tid];
        q.f_hist[tid] = tid < q.K ? fn_s[tid] : 0.0;
        if (tid == 0) *q.delta_out = red[1];
    }
    double* rec = q.rec + (int64_t)par * q.nrec * ROWS;
    // (the body copies a' into registers before it touches the tile buffers; its first DMA lands behind the barrier below)
    double a_loc[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) a_loc[k] = a_s[k];
    wait_vm<0>();  // this wave's share of the tables (and its first tile) has landed ...
    __syncthreads();  // ... and so has everybody else's
    if (st) st[2] = wall_clock64();
    lse_small_body<NB, true>(smem, u, ld, N, ntiles, a_loc, cw, nullptr, nullptr, rec, nullptr, q.live, q.balanced != 0 ? 1 : 0, q.pingpong ? (q.parity & 1) : 0, st,
                             (q.stamps && blockIdx.x == 0) ? q.stamps + 16 : nullptr);
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
// NBW > 4 (513 .. 1024 states, round 4): a wave's slice of a tile is 12-16 KB, so each wave keeps ONE buffer -- all of its
// operands are in registers right after the loop top, and the buffer is refilled there (the exponentials of the tile hide the
// request) -- instead of the layout-agnostic kernels' two reads of the matrix per candidate.
// ---------------------------------------------------------------------------------------------
// PMODE (round 5, the host-driven loop above 256 states): `u` is the resident probability matrix P = exp(a0 - u - logden(a0)),
// aden[k] the first candidate's MULTIPLIER exp(a_k - a0_k) (0: state without samples / padding), ld_anchor = logden(a0): an
// element is P_kn * multiplier -- no exponential, no shift, the first meeting of the waves is skipped -- and
// logden_n = ld_anchor_n + log(sum).
template <int NBW, int NF, bool PMODE>
__global__ void __launch_bounds__(512, 1)
k_lse_split(const double* __restrict__ u, int64_t ld, int64_t N, int64_t ntiles, int64_t rows,
            const double* __restrict__ aden, const double* __restrict__ cw, double* __restrict__ logden,
            double* __restrict__ logden1, const double* __restrict__ dn, double* __restrict__ psum_part,
            double* __restrict__ obj_part, const double* __restrict__ ld_anchor) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RW = NBW * 16;                  // rows per wave
    constexpr int NP = RW / 8;                    // LDS-DMA pieces per wave and tile
    constexpr int U_BYTES = RW * TS * 8;
    constexpr int TILE_BYTES = U_BYTES + (PMODE ? 2 : 1) * TS * 8;  // + the tile's 16 sample weights (P mode: + its 16 anchor log-denominators)
    constexpr int NW = 8;
    constexpr int NBUF = NBW <= 4 ? 2 : 1;        // tile buffers per wave
    constexpr int NREQ = NP + (PMODE ? 2 : 1);    // LDS-DMA requests of a wave that has all of its rows, per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = lane & 15, ns = lane >> 4;
    exp_table_init(smem);
    double* xmax = reinterpret_cast<double*>(smem + EXP_TABLE_BYTES);  // [NW][TS]
    double* xsum = PMODE ? xmax : xmax + NW * TS;                      // [NF][NW][TS]  (P mode: two of them, by tile parity, no maxima)
    constexpr int MEET_VECS = PMODE ? 2 * NF : 1 + NF;
    char* buf = smem + EXP_TABLE_BYTES + MEET_VECS * NW * TS * 8 + wave * (NBUF * TILE_BYTES);
    const int64_t r0 = (int64_t)wave * RW;
    const StageOffsets so = make_stage_offsets(ld, lane);
    // rows this wave never requests: zero once, in both buffers
    for (int j = 0; j < NP; ++j)
        if (r0 + 8 * j >= rows) {
            for (int bsel = 0; bsel < NBUF; ++bsel) *reinterpret_cast<double2*>(buf + bsel * TILE_BYTES + j * 1024 + lane * 16) = double2{0.0, 0.0};
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
        a[I] = r < rows ? aden[r] : (PMODE ? 0.0 : -INFINITY);
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
        if constexpr (PMODE) stage_vec16<true>(ld_anchor, tile * TS, dst + U_BYTES + TS * 8, lane);
    };
    const int64_t G = gridDim.x;
    int64_t t = blockIdx.x;
    int cur = 0, cur2 = 0;
    // A wave keeps NBUF tiles in flight: all of a tile's operands are in registers right after the loop top, and its buffer is
    // refilled THERE with the tile NBUF periods ahead (round 5; before, the two-buffer form requested tile i + 1 only once tile i
    // had landed -- one 64 KB tile per CU in flight, 3.4 TB/s at 512 states).  Waves whose rows are all real count their requests
    // (the queue retires in order: at most one tile's NREQ requests may still be out when this tile is read); the others --
    // fewer or no requests -- wait for everything.
    const bool counted = NBUF == 2 && r0 + RW <= rows;
    if (t < ntiles) stage(t, buf);
    if (NBUF == 2 && t + G < ntiles) stage(t + G, buf + TILE_BYTES);
    for (; t < ntiles; t += G) {
        char* cbuf = buf + cur * TILE_BYTES;
        if (counted && t + G < ntiles) wait_vm<NREQ>(); else wait_vm<0>();  // this tile has landed
        double x[GROUPS][NBW], w[GROUPS], mloc[GROUPS], lda[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            w[g] = *reinterpret_cast<const double*>(cbuf + U_BYTES + (4 * g + ns) * 8);
            lda[g] = PMODE ? *reinterpret_cast<const double*>(cbuf + U_BYTES + TS * 8 + (4 * g + ns) * 8) : 0.0;
#pragma unroll
            for (int I = 0; I < NBW; ++I) x[g][I] = *reinterpret_cast<const double*>(cbuf + I * (16 * TS * 8) + pos[g]);
        }
        {  // the wave's slice is in registers: its buffer takes the tile NBUF periods ahead now
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + NBUF * G < ntiles) stage(t + NBUF * G, cbuf);
            __builtin_amdgcn_sched_barrier(0);
        }
        double m2[GROUPS];
        if constexpr (!PMODE) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
                for (int I = 0; I < NBW; ++I) x[g][I] = a[I] - x[g][I];
                mloc[g] = row16_max(tree_max<NBW>(x[g]));
                if (ks == 0) xmax[wave * TS + 4 * g + ns] = mloc[g];
            }
            lds_barrier();  // (LDS traffic only: __syncthreads() would also wait for the tiles in flight)
        }
        // (P mode meets ONCE per tile: the sums alternate between two vectors, so a fast wave writing tile i+1's sums cannot
        // touch what a slow wave still reads of tile i, and tile i+2's writers are past tile i+1's barrier)
        double* xs = PMODE ? xsum + (cur2 ? NF * NW * TS : 0) : xsum;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if constexpr (PMODE) {
                m2[g] = 0.0;
#pragma unroll
                for (int I = 0; I < NBW; ++I) x[g][I] *= a[I];
            } else {
                double m = xmax[4 * g + ns];
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) m = fmax(m, xmax[wv * TS + 4 * g + ns]);
                m2[g] = m * LOG2E_S;
#pragma unroll
                for (int I = 0; I < NBW; ++I) x[g][I] = fma(x[g][I], LOG2E_S, -m2[g]);
                exp2s_batch<NBW>(x[g]);
            }
            double s0 = tree_sum<NBW>(x[g]), s1 = NF == 2 ? dot_sum<NBW>(x[g], c) : 0.0;
            if constexpr (NF == 2) row16_sum2(s0, s1); else s0 = row16_sum(s0);
            if (ks == 0) {
                xs[wave * TS + 4 * g + ns] = s0;
                if constexpr (NF == 2) xs[NW * TS + wave * TS + 4 * g + ns] = s1;
            }
        }
        lds_barrier();  // (LDS traffic only: __syncthreads() would also wait for the tiles in flight)
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            double ssum[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                ssum[f] = xs[f * NW * TS + 4 * g + ns];
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) ssum[f] += xs[f * NW * TS + wv * TS + 4 * g + ns];  // (fixed order: every wave gets the same bits)
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // (P mode: a padded sample has an all-zero column: keep its reciprocal finite, its multiplicity is 0)
                const double r = w[g] * recip_fast(PMODE ? fmax(ssum[f], 1e-300) : ssum[f]);
#pragma unroll
                for (int I = 0; I < NBW; ++I) acc[f][I] = fma(x[g][I], r, acc[f][I]);
            }
            if (wave == 0 && ks < NF) {  // logden_n = shift + log(sum), one lane per sample and candidate
                const int64_t n = t * TS + 4 * g + ns;
                if (n < N) {
                    const double sv = ks == 0 ? ssum[0] : ssum[NF - 1];
                    const double ldv = PMODE ? lda[g] + log_pos(fmax(sv, 1e-300)) : fma(m2[g], LN2_OVER_S, log_pos(sv));
                    double* out = ks == 0 ? logden : logden1;
                    if (out) out[n] = ldv;
                    const double term = w[g] * (dn ? (ldv - dn[n]) : ldv);
                    if (ks == 0) objl[0] += term; else objl[NF - 1] += term;
                }
            }
        }
        cur ^= NBUF - 1;
        cur2 ^= 1;
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
// host-side launchers
// ---------------------------------------------------------------------------------------------

LaunchGeom lse_geometry(int nb, int nf, int num_cu, int64_t ntiles, int64_t grid_override, int variant) {
    LaunchGeom g;
    const size_t tile = (size_t)nb * 16 * TS * 8 + TS * 8;  // u tile + its 16 sample weights
    const bool small_ok = (variant & 0x10) != 0;  // set by the caller when the context qualifies (pitch, staging)
    const bool wide_ok = (variant & 0x20) != 0;   // likewise for the single-buffer wide-panel kernel
    g.variant = 1;  // one tile stream per wave (k_lse); 4 = few-state kernel, 5 = single-buffer wide-panel kernel (below)
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
    g.waves = lse_waves(nb);
    g.lds_bytes = (size_t)g.waves * 2 * tile + EXP_TABLE_BYTES;
    int64_t want = (ntiles + g.waves - 1) / g.waves;
    int64_t cap = (int64_t)num_cu * blocks_per_cu_for(g.lds_bytes);
    if (grid_override > 0) cap = grid_override;
    if (want < 1) want = 1;
    g.blocks = (int)(want < cap ? want : cap);
    g.nwaves = g.blocks * g.waves;
    g.psum_records = g.nwaves;
    return g;
}

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

template <int NB>
static hipError_t launch_lse_nb(hipStream_t s, int nf, const LaunchGeom& g, const double* u,
                                int64_t ld, int64_t N, const double* aden, const double* cw, double* l0, double* l1,
                                const double* dn, double* pp, double* op, const LoopCtl& lc) {
    // (LDS-DMA staging only: the register-staged instantiations of rounds 1-3 were a tuning knob, never the faster path)
    if (g.variant != 1) return hipErrorInvalidValue;
    if (nf == 1) return launch_lse_t<NB, 1, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
    return launch_lse_t<NB, 2, true>(s, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
}

template <int NB>
static hipError_t launch_lse_small_t(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                                     const double* aden, const double* cw, double* l0, const double* dn,
                                     double* psum_part, double* obj_part) {
    auto kern = g.balanced ? k_lse_small<NB, 1> : k_lse_small<NB, 0>;
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

hipError_t launch_sci_small(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* cw,
                            const SciLoopArgs& q) {
    if (g.variant != 4 || (ld % TSS) != 0 || q.nrec != g.blocks || g.waves != 8) return hipErrorInvalidValue;
    const size_t lds = g.lds_bytes + (size_t)(512 + 3 * 16 * nb + 2) * sizeof(double);  // + the prologue's scratch behind the tile buffers
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const int64_t ntiles = (N + TSS - 1) / TSS;
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.waves * 64), lds, s, u, ld, N, ntiles, cw, q);
        return hipGetLastError();
    };
    if (nb == 1) return go(k_sci_small<1>);
    if (nb == 2) return go(k_sci_small<2>);
    return hipErrorInvalidValue;
}

hipError_t launch_lse(hipStream_t s, int nb, int nf, const LaunchGeom& g, const double* u,
                      int64_t ld, int64_t N, const double* aden, const double* cw, double* l0, double* l1,
                      const double* dn, double* pp, double* op, const LoopCtl& lc) {
    if (lc.ctl && g.variant == 4) return hipErrorInvalidValue;
    if (g.variant == 5) {  // (geometry chose the single-buffer wide-panel kernel: nb = 12 or 16, LDS-DMA staging)
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
        if (nf != 1 || (ld % TSS) != 0) return hipErrorInvalidValue;
        switch (nb) {
            case 1: return launch_lse_small_t<1>(s, g, u, ld, N, aden, cw, l0, dn, pp, op);
            case 2: return launch_lse_small_t<2>(s, g, u, ld, N, aden, cw, l0, dn, pp, op);
            default: return hipErrorInvalidValue;
        }
    }
    switch (nb) {
#define MBAR_CASE(NB_) \
    case NB_: return launch_lse_nb<NB_>(s, nf, g, u, ld, N, aden, cw, l0, l1, dn, pp, op, lc);
        MBAR_CASE(1) MBAR_CASE(2) MBAR_CASE(3) MBAR_CASE(4) MBAR_CASE(5) MBAR_CASE(6) MBAR_CASE(7)
        MBAR_CASE(8) MBAR_CASE(12) MBAR_CASE(16)
#undef MBAR_CASE
        default: return hipErrorInvalidValue;
    }
}

static int stream_blocks(int num_cu, int64_t N) {
    int64_t want = (N + 255) / 256;
    int64_t cap = (int64_t)num_cu * 8;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// 257 .. 1024 states in one read: rows = allocated row count (a multiple of 64); returns the number of partial records.
hipError_t launch_lse_split(hipStream_t s, int num_cu, int nf, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                            const double* cw, double* logden, double* logden1, const double* dn, double* psum_part, double* obj_part,
                            int* blocks_out, const double* ld_anchor) {
    int nbw = (int)((rows + 127) / 128);
    if (nbw < 1 || nbw > 8 || nf < 1 || nf > 2) return hipErrorInvalidValue;
    if (nbw > 4) nbw = nbw <= 6 ? 6 : 8;  // (513 .. 768 / 769 .. 1024 rows: rows a wave does not have are never requested)
    const int64_t ntiles = (N + TS - 1) / TS;
    const size_t tile = (size_t)nbw * 16 * TS * 8 + (ld_anchor ? 2 : 1) * TS * 8;
    const size_t lds = EXP_TABLE_BYTES + (size_t)(ld_anchor ? 2 * nf : 1 + nf) * 8 * TS * 8 + (size_t)8 * (nbw <= 4 ? 2 : 1) * tile;
    const int blocks = (int)(ntiles < num_cu ? (ntiles < 1 ? 1 : ntiles) : num_cu);
    *blocks_out = blocks;
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, s, u, ld, N, ntiles, rows, aden, cw, logden, logden1, dn, psum_part,
                           obj_part, ld_anchor);
        return hipGetLastError();
    };
    if (ld_anchor) {  // P mode (the host-driven loop above 256 states: 17 .. 64 blocks of 16 rows)
        if (nf == 1) {
            switch (nbw) {
                case 3: return go(k_lse_split<3, 1, true>);
                case 4: return go(k_lse_split<4, 1, true>);
                case 6: return go(k_lse_split<6, 1, true>);
                case 8: return go(k_lse_split<8, 1, true>);
                default: return hipErrorInvalidValue;
            }
        }
        switch (nbw) {
            case 3: return go(k_lse_split<3, 2, true>);
            case 4: return go(k_lse_split<4, 2, true>);
            case 6: return go(k_lse_split<6, 2, true>);
            case 8: return go(k_lse_split<8, 2, true>);
            default: return hipErrorInvalidValue;
        }
    }
    if (nf == 1) {
        switch (nbw) {
            case 1: return go(k_lse_split<1, 1, false>);
            case 2: return go(k_lse_split<2, 1, false>);
            case 3: return go(k_lse_split<3, 1, false>);
            case 4: return go(k_lse_split<4, 1, false>);
            case 6: return go(k_lse_split<6, 1, false>);
            default: return go(k_lse_split<8, 1, false>);
        }
    }
    switch (nbw) {
        case 1: return go(k_lse_split<1, 2, false>);
        case 2: return go(k_lse_split<2, 2, false>);
        case 3: return go(k_lse_split<3, 2, false>);
        case 4: return go(k_lse_split<4, 2, false>);
        case 6: return go(k_lse_split<6, 2, false>);
        default: return go(k_lse_split<8, 2, false>);
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

}  // namespace mbar
