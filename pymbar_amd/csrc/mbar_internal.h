// Internal interface between the host side (mbar_capi.cpp, mbar_loops.cpp, mbar_comm.cpp, mbar_host.cpp; their shared header is
// mbar_ctx.h) and the gfx950 kernels (mbar_k_*.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbar {

constexpr int TS = 16;       // samples per wave tile (one 128-byte line per state row)
constexpr int GROUPS = 4;    // 4-sample MFMA groups per tile
constexpr int MAX_FAST_K = 256;   // fast (LDS-staged, MFMA-layout) kernels handle K <= 256
constexpr int PANEL = 64;    // Gram panel width (states) when K > 128

// Rounded-up state count of the device matrix.
inline int64_t padded_K(int64_t K) {
    if (K <= 128) return (K + 15) / 16 * 16;
    return (K + PANEL - 1) / PANEL * PANEL;
}
// Supported block counts (16 states each) of the fused evaluation kernel.
inline int lse_nb_for(int64_t Kp) {
    static const int nbs[] = {1, 2, 3, 4, 5, 6, 7, 8, 12, 16};
    int need = (int)(Kp / 16);
    for (int nb : nbs) if (nb >= need) return nb;
    return 0;
}

// Bootstrap draws as a counter-based stream (mbar_ctx_draw_bootstrap_weights on the device, mbar_bootstrap_draws on the host: the
// same function of (seed, replicate, slot)): slot j of a state with n samples draws position bootstrap_draw(...) in [0, n).
// Two rounds of the splitmix64 finaliser over key and counter; the range reduction is a 64 x 64 -> high-64 multiply (bias <= n / 2^64).
#if defined(__HIPCC__)
#define MBAR_HD __host__ __device__
#else
#define MBAR_HD
#endif
MBAR_HD inline uint64_t bootstrap_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
MBAR_HD inline int64_t bootstrap_draw(uint64_t seed, uint64_t replicate, uint64_t slot, uint64_t n) {
    const uint64_t key = bootstrap_mix64(seed + 0x9E3779B97F4A7C15ull * (replicate + 1));
    const uint64_t z = bootstrap_mix64(bootstrap_mix64(key ^ (slot * 0xD1342543DE82EF95ull)) + slot);
    return (int64_t)(((unsigned __int128)z * (unsigned __int128)n) >> 64);
}

// Control words of the device-resident solver loop (ints in device memory).  Kernels that are handed a pointer to them
// exit at once when CTL_DONE is set (iterations enqueued past convergence are no-ops) and pick the logden vector of the
// current f from three rotating slots (base + slot * slot_stride), so that a whole iteration can be enqueued -- or
// replayed from a hipGraph -- without the host knowing which candidate the previous one accepted.
enum : int {
    CTL_SLOT = 0,    // logden slot of the current f
    CTL_DONE = 1,    // 0 = running, 1 = converged, 2 = handed back to the host (CTL_REASON says why), 3 = fused loop paused:
                     // the accepted candidate is not the one whose Gram matrix the sweep speculated on, the host enqueues
                     // the separate Gram sweep and resumes
    CTL_ITER = 2,    // iterations executed
    CTL_SCI = 3,     // ... of which self-consistent steps were accepted
    CTL_NR = 4,      // ... of which Newton-Raphson steps were accepted
    CTL_REASON = 5,  // 1 = Newton system not positive definite, 2 = candidates too far apart for the fused sweep,
                     // 3 = non-finite candidate
    CTL_NEEDGRAM = 6,  // fused sweep: 1 = the accepted candidate's Gram matrix is NOT the speculated one: run the Gram sweep
    CTL_GRAMSWEEPS = 7,  // separate Gram sweeps requested so far
    CTL_SPEC = 8,    // fused sweep: candidate whose Gram matrix the sweep accumulates (always the SECOND multiplier row it is
                     // handed): 1 = Newton-Raphson (default), 0 = self-consistent (while self-consistent steps are forced,
                     // mbar_solvers.py:607 `sci_iter < min_sc_iter`: k_newton then hands the two rows over in swapped order)
    CTL_LIGHT = 9,   // fused loop: 1 = BOTH candidates of the coming sweep already satisfy the stop test (mbar_solvers.py:636) against
                     // the current f, so this iteration is the last whichever of them wins and nobody will need the Gram
                     // matrix the fused sweep would accumulate: the fused sweep returns at once and the plain two-candidate
                     // sweep on P (k_psweep, launched right behind it and idle otherwise) evaluates the candidates instead; the
                     // one-read fused sweep of 129 .. 256 states switches to an evaluation-only body of its own
    CTL_LIGHTS = 10,  // iterations evaluated that way
    CTL_WORDS = 12
};
struct LoopCtl {
    const int* ctl = nullptr;
    int64_t slot_stride = 0;
    // optional: events bound to the kernel dispatch itself (hipExtLaunchKernelGGL): start / stop time stamps of the
    // kernel with no marker packets in the stream (an event record between two kernels costs ~6 us of idle queue)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // the matrix holds no +inf entry: the full 128-state Gram panel may run its exponentials without the clamp
    bool unclamped = false;
    // the matrix handed to the Gram launcher is the resident probability matrix (P mode)
    bool pmode = false;
    // the Gram launch is conditional: the kernel exits at once unless CTL_NEEDGRAM is set (fused-sweep loop)
    bool cond_needgram = false;
    // the evaluation sweep on P is conditional: it exits at once unless CTL_LIGHT is set (last iteration of the fused loop)
    bool light_only = false;
};

struct LaunchGeom {
    int blocks;        // grid size
    int waves;         // waves per block
    int nwaves;        // number of partial records of the main output (per wave, or per tile stream)
    int psum_records;  // number of partial records of the per-state sums (Gram kernels)
    int variant;       // kernel variant chosen (see lse_geometry / gram_geometry)
    size_t lds_bytes;
    int live_blocks = 0;  // k_gram_quad / k_fused_quad: blocks of 16 states that hold real states (0: all of the panel's)
    int balanced = 0;     // few-state kernels: the waves' tile streams start workgroup-major (see lse_small_first_tile)
};

// ---- evaluation pass -------------------------------------------------------------------------
// psum_part: [nwaves][nf][16*nb], obj_part: [nwaves][nf]; logden0/1 may be null (not stored).
// cw: per-sample multiplicities (ld doubles: 1 for plain data, bootstrap counts otherwise, 0 on the padding).
// variant: flags of the specialised kernels the context qualifies for (0x10: few-state kernel, one sample per lane, K <= 32,
// one candidate; 0x20: single-buffer kernel for 129..256 states); the geometry records its choice in LaunchGeom::variant
// (1 = k_lse, one tile stream per wave; 4 = k_lse_small; 5 = k_lse_wide)
LaunchGeom lse_geometry(int nb, int nf, int num_cu, int64_t ntiles, int64_t grid_override, int variant);
hipError_t launch_lse(hipStream_t s, int nb, int nf, const LaunchGeom& g,
                      const double* u, int64_t ld, int64_t N, const double* aden /*[nf][16nb]*/,
                      const double* cw, double* logden0, double* logden1, const double* dn,
                      double* psum_part, double* obj_part, const LoopCtl& lc = LoopCtl());

// ---- Gram pass (known logden) ------------------------------------------------------------------
// Diagonal panel: states [row0, row0+16nb) against themselves, nblk = nb(nb+1)/2 blocks, block b
// enumerates (I,J) with I<=J in row-major order.  Off-diagonal panel pair: nbi x nbj blocks.
// gram_part: [nwaves][nblk][256], psum_part: [nwaves][16*nb] (diag only; may be null for off-diag).
LaunchGeom gram_geometry(int tile_rows, bool diag, int num_cu, int64_t ntiles, int64_t grid_override);
hipError_t launch_row_sub(hipStream_t s, double* row, const double* v, int64_t n);  // row[i] -= v[i]
hipError_t launch_rows_sub(hipStream_t s, double* dst, const double* src, int64_t ld, int64_t nrows, const double* v, int64_t n);
hipError_t launch_rows_obs(hipStream_t s, double* dst, const double* obs, const double* state, int64_t ld, int64_t nrows, int64_t n,
                           double* part, double* shift_out, bool have_min = false);
hipError_t launch_rows_rsub(hipStream_t s, double* dst, const double* src, int64_t ld, int64_t nrows, int64_t n);
// rows r < nrows of base (pitch ld, n valid entries): row <- log(row - shift_r), shift_r = min_r - |4 eps min_r| -> shift_out[r]
// (device); part: scratch of 256 * nrows doubles
hipError_t launch_rows_logshift(hipStream_t s, double* base, int64_t ld, int64_t nrows, int64_t n, double* part, double* shift_out);
// rows[i][k] = label[k] == i ? v[k] : +inf,  i < nrows (row pitch ld)
hipError_t launch_fill_masked_rows(hipStream_t s, double* rows, int64_t ld, int64_t n, int64_t nrows, const double* v,
                                   const int* label);
hipError_t launch_gram_diag(hipStream_t s, int nb, const LaunchGeom& g, const double* u,
                            int64_t ld, int64_t N, const double* anum /*indexed from row0*/,
                            const double* logden, int64_t row0, double* gram_part, double* psum_part,
                            const LoopCtl& lc = LoopCtl());
// (pmode: `u` is the resident probability matrix, `logden` the reciprocals 1 / s_n, the anum vectors are not read)
hipError_t launch_gram_thin(hipStream_t s, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* ai,
                            const double* aj, const double* logden, int64_t ri, int64_t rj, double* gp);
hipError_t launch_gram_off(hipStream_t s, int nbj /*4 or 8*/, const LaunchGeom& g, const double* u, int64_t ld,
                           int64_t N, const double* anum_i, const double* anum_j, const double* logden,
                           int64_t row_i0, int64_t row_j0, double* gram_part, bool pmode = false);

// 129 .. 256 states (nbt = 12 or 16 blocks of 16) in ONE read: the four waves of a workgroup share a tile stream and split
// the nbt (nbt + 1) / 2 upper-triangular blocks; gram_part: [blocks][nblk][256], block b = (I, J), I <= J, row-major.
// LDS-DMA staging only.  lc.pmode: `u` is the resident probability matrix, `logden` the reciprocals 1 / s_n.
LaunchGeom gram_quad_geometry(int nbt, int num_cu, int64_t ntiles, int64_t grid_override);
// Pout (classic operands only): the operand tiles exp(anum - u - logden) are also written out there (the probability matrix)
// P mode only: nbi (4 / 8) x 16 blocks between the panels at rows ri and rj of the resident probability matrix, see k_gram_rect
hipError_t launch_gram_rect(hipStream_t s, int nbi, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, int64_t ri, int64_t rj,
                            const double* rinv, double* gram_part);
hipError_t launch_gram_quad(hipStream_t s, int nbt, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                            const double* anum, const double* logden, double* gram_part, const LoopCtl& lc = LoopCtl(),
                            double* Pout = nullptr);
hipError_t launch_gram_quad_split(hipStream_t s, int nbt, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* anum,
                                  const double* logden, double* gram_part, int64_t split_rows, int64_t row_j0);

// ---- layout-agnostic fallbacks (any K) ---------------------------------------------------------
// 257 .. 512 states in one read: nf = 2 evaluates a second candidate through the ratio row aden[rows + k] = exp(a'_k - a_k)
// (per-state sums without that factor); psum_part [blocks][nf][rows], obj_part [blocks][nf]
hipError_t launch_lse_split(hipStream_t s, int num_cu, int nf, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                            const double* cw, double* logden, double* logden1, const double* dn, double* psum_part, double* obj_part,
                            int* blocks_out, const double* ld_anchor = nullptr /* P mode: u = P, aden = multipliers, see k_lse_split */);
hipError_t launch_lse_generic(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t K,
                              const double* aden, const double* cw, double* logden, const double* dn,
                              double* obj_part /*[blocks]*/, int* blocks_out);
hipError_t launch_colsum_generic(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t K,
                                 const double* anum, const double* cw, const double* logden,
                                 double* psum_part /*[blocks_x][K]*/, int* blocks_out);
// out[n] = logden[n] - alpha ln cw[n] (+inf where cw = 0)
hipError_t launch_shift_logden(hipStream_t s, const double* logden, const double* cw, double alpha, int64_t N,
                               double* out, const LoopCtl& lc = LoopCtl());

// ---- reductions / small kernels ----------------------------------------------------------------
// out[i] = sum_p part[p*count + i]; scratch must hold ceil(nparts/32)*count doubles.
hipError_t launch_reduce(hipStream_t s, const double* part, int64_t nparts, int64_t count,
                         double* scratch, double* out);
// robust per-state log-sum-exp over n of (anum_k - u_kn - logden_n): partial (max,sum) per chunk
hipError_t launch_lognum(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K,
                         const double* anum, const double* logden,
                         double* pmax /*[K][nchunks]*/, double* psum /*[K][nchunks]*/, int64_t nchunks);
int64_t lognum_chunks(int64_t N, int64_t K);
hipError_t launch_lognum_merge(hipStream_t s, const double* pmax, const double* psum, int64_t K,
                               int64_t nchunks, double* out_max /*[K]*/, double* out_sum /*[K]*/);
hipError_t launch_logw(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K,
                       const double* f, const double* logden, double* out, int64_t ld_out, bool exponentiate = false);
// flags |= 1 if any u[k][n] is NaN, |= 2 if any is -inf (k < K, n < N)
hipError_t launch_check_u(hipStream_t s, const double* u, int64_t ld, int64_t N, int64_t K, int* flags);
hipError_t launch_generate_harmonic(hipStream_t s, double* u, int64_t ld, int64_t N, int64_t K,
                                    uint64_t seed, const double* O_k, const double* K_k,
                                    const int64_t* cumN /*[K+1]*/, int64_t n_global0);
// device-resident SCI step: sums `nparts` partial psum records (row pitch `rows`), then
// f' = f - log(psum/N_k) on sampled states, gauge, aden' = f' + ln N_k; f' also goes to f_hist
hipError_t launch_sci_update(hipStream_t s, const double* part, int64_t nparts, int64_t rows, const double* Nk,
                             const double* lnNk, int64_t K, int64_t Kp, int first_state, double tol, double* f,
                             double* aden, double* f_hist, double* delta_out);
// Few states (K <= 32, single rank): update + sweep of ONE self-consistent iteration in one launch (k_sci_small).  Double
// buffers by parity p of the iteration: the launch reads rec[p ^ 1] (nrec records of 16 nb doubles: the per-state sums of the
// previous sweep, one per workgroup of the SAME geometry) and state[p ^ 1] (the previous f), writes rec[p], state[p], the
// history row and the relative change.
struct SciLoopArgs {
    const double* Nk;
    const double* lnNk;
    int K;
    int first;        // gauge state
    double tol;
    double* state;    // [2][16 nb]
    double* rec;      // [2][nrec][16 nb]
    int64_t nrec;
    double* f_hist;   // [16 nb] row of the batch history this iteration fills
    double* delta_out;
    int parity;
    uint32_t live;    // bit j: rows 2j, 2j+1 hold a state with samples (the others are not streamed from HBM)
    int balanced;     // tile streams start workgroup-major (LaunchGeom::balanced; the previous sweep's records do not care)
    int pingpong;     // odd iterations sweep the tiles in descending order (cache re-use between consecutive sweeps)
    long long* stamps;  // MBAR_DEBUG_STAMPS: [2][8] phase stamps (100 MHz) of thread 0 of workgroups 0 and gridDim.x / 2, or nullptr
};
hipError_t launch_sci_small(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N, const double* cw,
                            const SciLoopArgs& q);
// (*overflow, zeroed by the caller, is set when a weight is not finite)
// cw[sample] += 1 for every draw of the replicate whose sample lies in this shard [n0, n0 + N) (cw zeroed by the caller; the adds are
// of exact small integers: order-independent).  cum: [K + 1] positions of the states' runs; order: sample index of a position, or NULL
hipError_t launch_bootstrap_counts(hipStream_t s, uint64_t seed, int64_t replicate, const int64_t* cum, int64_t K, int64_t total,
                                   const int64_t* order, int64_t n0, int64_t N, double* cw);
hipError_t launch_weights_from_log(hipStream_t s, const double* v, double p, int64_t n, double* cw, double* cwsq, int* overflow);
hipError_t launch_reduce_level1(hipStream_t s, const double* part, int64_t nparts, int64_t count, double* out,
                                int64_t* nchunks);
hipError_t launch_mfma_peak(hipStream_t s, int blocks, int iters, double* sink);
// Two partial-record arrays with the same number of records reduced by one pair of launches (same summation order as
// launch_reduce on each): outA[i] = sum_p partA[p*countA + i], outB likewise.  scratch: ceil(nparts/32)*(countA+countB).
hipError_t launch_reduce2(hipStream_t s, const double* partA, int64_t countA, const double* partB, int64_t countB,
                          int64_t nparts, double* scratch, double* outA, double* outB);

// In-process all-reduce (several contexts of one process on one device): out[i] = op over the ranks' buffers, in rank order
struct LoopSrc {
    const double* p[8];
    int n;
};
hipError_t launch_loop_reduce(hipStream_t s, const LoopSrc& src, int64_t count, int op /*0 sum, 1 max*/, double* out);

// ---- device-resident adaptive iteration (mbar_solvers.py:575-640 without the host in the loop) --------------------
// State of one solve, all in device memory.  Up to 128 padded states (one diagonal Gram panel).
struct AdaptArgs {
    const double* gram_red;  // reduced Gram blocks of the panel, block b = (I, J), I <= J, row-major 16 x 16 each
    const double* lse_red;   // reduced outputs of the two-candidate sweep: psum[2][Kp], then 2 objective sums
    double* f;               // [Kp] current free energies
    double* psum;            // [Kp] sum_n p_nk at f
    double* cand;            // [2][Kp] f_sci, f_nr
    double* ratio;           // [Kp] exp(aden_nr - aden_sci): the fused sweep returns the second candidate's sums unscaled
    double* aden;            // [2][Kp] input of the sweep: aden of f_sci, then ratio
    double* anum;            // [Kp] f + ln N_k (-inf for unsampled / padded states): operand constant of the Gram pass
    const double* Nk;        // [Kp]
    const double* lnNk;      // [Kp]
    const int* sampled;      // [m] states with N_k > 0, ascending
    int m, K, Kp;
    int* ctl;                // CTL_WORDS ints
    const double* prm;       // gamma, tol, min_sc_iter, check_convergence
    double* state;           // [0] max_delta of the last iteration
    double* hist;            // rows of 4 doubles {choice, |g_sci|, |g_nr|, max_delta}
    int64_t hist_cap;
    // P mode: the sweeps work on P = exp(a0 - u - logden(a0)); aden then holds the multipliers exp(a - a0) of BOTH
    // candidates, ccur those of the current f (the Gram kernel returns G with both factors still to be applied)
    int pmode;
    const double* a0;        // [Kp] aden at the build point (-inf for unsampled / padded states)
    double* ccur;            // [Kp]
    // fused sweep: the Gram matrix in gram_red was accumulated with the multipliers cgram (the speculated Newton-Raphson
    // candidate's, or the current f's after a separate Gram sweep)
    int fused;
    double* cgram;           // [Kp]
    // fused loop: the last iteration may run without its Gram matrix (CTL_LIGHT); 0 = never (small problems: an idle launch
    // per iteration would cost more than the one lighter sweep saves)
    int light_ok;
    // MBAR_DEBUG_STAMPS=1: shader-clock stamps of the phases of k_select_newton (thread 0; [16] per launch slot, 64 slots)
    long long* stamps;
    // K x K Newton solve up to 128 states: 1 = blocked LDL^T on the matrix cores (newton_body_ldlt), 0 = register Gauss-Jordan
    int newton_ldlt;
};
// ---- P mode: resident probability matrix P = exp(a0 - u - logden(a0)) (see k_psweep) ----------------------------------
LaunchGeom psweep_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override);
// cmul: [nf][16 nb] multipliers exp(a - a0); rinv0 = base of the three slot vectors when lc.ctl is set
hipError_t launch_psweep(hipStream_t s, int nb, int nf, const LaunchGeom& g, const double* P, int64_t ld, int64_t N,
                         const double* cmul, const double* cw, double* rinv0, double* rinv1, double* psum_part,
                         const LoopCtl& lc = LoopCtl());
// fused sweep: both candidates' normalisers / per-state sums + the Gram matrix of the second (Newton-Raphson) candidate
LaunchGeom fused_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override);
hipError_t launch_fused(hipStream_t s, int nb, const LaunchGeom& g, const double* P, int64_t ld, int64_t N, const double* cmul,
                        const double* cw, const double* wsq, double* rinv_base, double* gram_part, double* psum_part,
                        const LoopCtl& lc);
// fused build: the single-candidate sweep at the anchor point (psum partial records [nwaves][16 nb]) that also writes P and
// fills the reciprocal slot with ones
LaunchGeom build_sweep_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override);
hipError_t launch_build_sweep(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                              const double* aden, const double* cw, double* P, double* rinv_slot, double* psum_part);
// ... and the Gram matrix at the anchor point as well (gram partial records in the fused sweep's per-wave layout and count).  It
// leaves NO per-state sums: they are the row sums of that Gram matrix (host::gram_row_sums on the reduced blocks).  general: the samples
// carry multiplicities (wsq = their square roots; otherwise any vector of ld doubles) or the matrix holds +inf entries.
LaunchGeom build_gram_geometry(int nb, int num_cu, int64_t ntiles, int64_t grid_override);
hipError_t launch_build_gram(hipStream_t s, int nb, const LaunchGeom& g, const double* u, int64_t ld, int64_t N,
                             const double* aden, const double* wsq, bool general, double* P, double* rinv_slot, double* gram_part);

// 129 .. 256 states: P = exp(aden_k - u_kn - logden_n) (rows = padded state count; padding and sub-normal entries: 0)
hipError_t launch_make_p(hipStream_t s, int num_cu, const double* u, int64_t ld, int64_t N, int64_t rows, const double* aden,
                         const double* logden, double* P);
hipError_t launch_fill(hipStream_t s, double* v, double value, int64_t n);
// zero fill at HBM write speed (hipMemsetAsync below 64 KB or for unaligned ranges)
hipError_t launch_zero(hipStream_t s, void* p, size_t bytes);
hipError_t launch_sqrt_vec(hipStream_t s, double* dst, const double* src, int64_t n);  // dst[i] = sqrt(src[i])
hipError_t launch_rinv_from_logden(hipStream_t s, const double* ld0, const double* ldv, const double* cw, bool weighted, int64_t N,
                                   double* out);
hipError_t launch_rinv_weighted(hipStream_t s, const double* rinv, const double* cw, int64_t N, double* out,
                                const LoopCtl& lc = LoopCtl());
// K x K Newton system (gauge-fixed, Gauss-Jordan in registers, one workgroup) + both candidates + sweep inputs
hipError_t launch_newton(hipStream_t s, const AdaptArgs& a);
// ... for 128 .. 255 unknowns: blocked Cholesky in device memory (a pair of small kernels per block column of 32);
// work: NEWTON_CHOL_WORK doubles
constexpr size_t NEWTON_CHOL_WORK = (size_t)256 * 256 + 8;
hipError_t launch_newton_chol(hipStream_t s, const AdaptArgs& a, double* work);
// gradient norms of both candidates, choice (mbar_solvers.py:607), convergence test (:627-640), next Gram operand
hipError_t launch_select(hipStream_t s, const AdaptArgs& a);
// selection of one iteration + Newton solve of the next in one launch (fused loop, up to 127 unknowns)
hipError_t launch_select_newton(hipStream_t s, const AdaptArgs& a);
// fused loop paused by k_select (CTL_DONE = 3): clear the pause and the Gram request (in front of the Gram sweep)
hipError_t launch_ctl_resume(hipStream_t s, int* ctl);

}  // namespace mbar
