/*
 * mbar_hip.h -- C ABI of libmbar_hip.so: the MI355X (gfx950) MBAR solver hot path.
 *
 * This is the drop-in boundary for ONE path of choderalab/pymbar: the array math and solver
 * loops of pymbar/mbar_solvers.py (reference paths below are relative to the pymbar source
 * tree).  The reference is pure Python with no FFI; the binding a pymbar maintainer would add is
 * a ctypes module exporting the same names as pymbar.mbar_solvers (see INTEGRATION.md and
 * pymbar_amd/mbar_solvers.py, which is that module).
 *
 * Conventions
 *   - plain C, no torch / no C++ types; every pointer is a host pointer to caller-owned memory
 *     unless stated otherwise; all floating point is IEEE fp64; sizes are int64_t.
 *   - return value 0 = MBAR_OK, negative = error; mbar_last_error() gives the message.
 *   - a context owns the device-resident (K x N_local) reduced-potential matrix of ONE rank and
 *     is not thread-safe (one context per caller thread).  One process drives one GPU.
 *   - "p_nk" below is N_k * W_nk = N_k exp(f_k - u_kn) / sum_j N_j exp(f_j - u_jn): the
 *     probability that sample n was drawn from state k (rows sum to 1; 0 for unsampled states).
 *     With s_k = sum_n W_nk:  gradient g_k = N_k (s_k - 1) = psum_k - N_k   (mbar_solvers.py:284-292)
 *                             SCI      f'_k = f_k - log s_k                   (mbar_solvers.py:231-242)
 *                             Hessian  H = diag(psum) - gram                  (mbar_solvers.py:395-411)
 *   - when a communicator is attached (mbar_ctx_comm_init / mbar_ctx_set_host_allreduce) every
 *     reduced output (psum, sumlogden, gram, lognum) is the sum over all ranks' column shards.
 */
#ifndef MBAR_HIP_H
#define MBAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbar_ctx mbar_ctx;

#define MBAR_OK 0
#define MBAR_ERR_ARG (-1)      /* bad argument                                   */
#define MBAR_ERR_HIP (-2)      /* a HIP runtime call failed                      */
#define MBAR_ERR_NODEVICE (-3) /* no usable gfx950 device                        */
#define MBAR_ERR_STATE (-4)    /* call sequence error (e.g. N_k not set)         */
#define MBAR_ERR_COMM (-5)     /* RCCL / host all-reduce failure                 */
#define MBAR_ERR_NUMERIC (-6)  /* non-finite intermediate where one is not legal */

/* mbar_eval flags */
#define MBAR_EVAL_GRAM 1u        /* also accumulate gram_ij = sum_n p_ni p_nj at f[0]          */
#define MBAR_EVAL_USE_OFFSET 2u  /* sumlogden = sum_n (logden_n - d_n), d_n from mbar_ctx_set_objective_offset */

/* kernel classes for mbar_ctx_timing */
#define MBAR_TIMER_LSE 0    /* per-sample log-sum-exp + per-state sums (evaluation pass)   */
#define MBAR_TIMER_GRAM 1   /* fp64 MFMA Gram / Hessian pass                              */
#define MBAR_TIMER_REDUCE 2 /* partial-sum reductions                                      */
#define MBAR_TIMER_OTHER 3  /* logW writer, robust per-state LSE, generator, P-mode build  */
#define MBAR_TIMER_FUSED 4  /* fused sweep of the adaptive loop (candidates + MFMA Gram)   */
#define MBAR_TIMER_NEWTON 5 /* K x K Newton solve + candidate selection (timing level 3 only)   */
#define MBAR_TIMER_COMM 6   /* the per-iteration all-reduce on the stream (timing level 3 only) */
#define MBAR_TIMER_COUNT 7

/* ---- library / device -------------------------------------------------------------------- */
int mbar_version(void);
/* Message of the last error on this context (ctx may be NULL: last error of a failed create). */
const char* mbar_last_error(const mbar_ctx* ctx);
int mbar_device_count(int* count);
int mbar_device_info(int device, char* name, int name_len, int* compute_units, int64_t* total_mem_bytes);

/* ---- context ------------------------------------------------------------------------------- */
/* Allocate a context on `device` for K states and N_local samples (this rank's column shard).
 * Replaces the implicit "arrays live in host RAM" of the reference (mbar.py:243). */
int mbar_ctx_create(mbar_ctx** out, int device, int64_t K, int64_t N_local);
void mbar_ctx_destroy(mbar_ctx* ctx);
int mbar_ctx_synchronize(mbar_ctx* ctx);
/* Device and pinned-host blocks freed by contexts are kept for re-use (hipMalloc / hipFree cost more than a sweep at the
 * sizes pymbar is typically run at, and 0.3-0.7 s for the 6-8 GB augmented matrix of an expectation call at K=128, N=4e6);
 * bounded by MBAR_CACHE_MB (environment; default an eighth of the device's memory; 0 = off) WHILE the process has a context,
 * and cut back to MBAR_CACHE_IDLE_MB (default 1024) when its last context is destroyed; an allocation that fails empties
 * the cache and is tried again.  This returns every parked block to the driver. */
int mbar_cache_trim(void);
/* Environment variables read by the library (diagnostics; none changes a result):
 *   MBAR_CACHE_MB        bound of the block cache above (MBAR_CACHE_IDLE_MB: what it keeps once no context is left)
 *   MBAR_HOST_THREADS    team size of the host-side K x K factorisation (default: one thread per 96 unknowns, at most 16 and at
 *                        most the cores that share the L3 cache with the caller's core; the team's worker threads are created on
 *                        first use, parked between calls and kept on those cores -- the caller's own thread is never touched)
 *   MBAR_HOST_TEAM_AFFINITY  0 = leave the placement of the team's workers to the scheduler
 *   MBAR_DEBUG_TIMING    per-iteration wall-clock split of the host-driven loops and of the host factorisation on stderr
 *   MBAR_DEBUG_STAMPS    in-kernel time stamps on stderr: the phases of k_select_newton (selection / set-up / elimination /
 *                        candidates, shader clocks) and of k_sci_small (tables, update, first tile, sweep per wave, fold) */
/* 128-bit content digest of a HOST buffer, computed at memory speed on `threads` host threads (0 = all cores).  The reference's
 * module-level functions are pure functions of the u_kn they are handed (mbar_solvers.py:260-292: every call reads the current
 * array); the Python binding keeps device copies of recently seen host matrices and re-uses one only when the digest of the
 * bytes now behind the address equals the digest of what it uploaded.  A change of ONE element always changes the digest. */
int mbar_host_digest(const void* data, int64_t nbytes, int threads, uint64_t* out2);
/* The K x K step of the adaptive iteration as the HOST-driven loop runs it (more than 256 states, host transport):
 * x = H^+ g - (H^+ g)[0] (mbar_solvers.py:582-583: numpy.linalg.lstsq(H, g) then the gauge shift).  H (m x m, row-major) is the
 * Hessian on the states with samples, positive semi-definite with null vector 1: the gauge-fixed (m-1) x (m-1) block is solved by
 * Cholesky factorisation -- from 320 unknowns on in column blocks whose rows are shared out over a team of host threads, results
 * independent of the team size -- and only if that breaks down (disconnected states) by the minimum-norm pseudo-inverse.
 * threads = 0: what the solver loop does; threads > 0: the blocked factorisation with that team size (tests).  Host only. */
int mbar_host_newton_direction(const double* H, const double* g, int m, int threads, double* x);
/* hipDeviceSynchronize on `device` (every stream of every context): the bracket of a timed region. */
int mbar_device_synchronize(int device);
/* Tuning / test knobs.  Every key selects between code paths that are BOTH needed somewhere (a fallback when memory is short,
 * a transport that needs the host in the loop, a state count outside a kernel's range), so that tests and A/B measurements can
 * force the path a default configuration would not take; the defaults are the measured best.  (Rounds 1-3 carried further keys
 * for kernel variants that were measured never-best -- "staging", "lse_variant", "gram_variant", "persistent" -- removed in
 * round 4 with their kernels; profiles/README.md keeps the measurements.)
 *   "grid_blocks"    0 = auto
 *   "force_generic"  1 = layout-agnostic fallback kernels for any K (what K > 512 runs)
 *   "check_finite"   default 1: NaN / -inf scan of the matrix after every upload
 *   "small_k_kernel" 1 = one-sample-per-lane sweep for K <= 32, single candidate (default), 0 = the general sweep
 *   "wide_k_kernel"  1 = single-buffer sweep with four waves per CU for 129 <= K <= 256 (default), 0 = the general sweep
 *   "device_loop"    1 = adaptive iterations run device-resident where possible (default), 0 = host-driven loop (what the
 *                    host all-reduce transport and K > 256 run)
 *   "adapt_batch"    adaptive iterations enqueued between two looks at the control words (default 8)
 *   "pmode"          1 = the device-resident loop keeps P = exp(a0 - u - logden(a0)) resident (one more K x N array, built
 *                    once per solve) and sweeps that: no exponentials in the loop (default); 0 = sweeps recompute them from u
 *                    (what runs when P does not fit on some rank)
 *   "fused"          1 = in P mode ONE sweep per iteration: the candidate sweep also accumulates the Gram matrix of the
 *                    candidate about to be accepted, a separate Gram sweep runs only when that speculation is rejected
 *                    (default); 0 = two sweeps per iteration on P (evaluation + Gram): the A/B of the fusion
 *   "gram_quad"      1 = 129 <= K <= 256: the Gram sweep reads the matrix ONCE, the four waves of a workgroup share the tile
 *                    stream and split the panel's 78 / 136 blocks (default); 0 = 128-state panels + 64 x 128 rectangles
 *                    (2.5 reads: what K > 256 runs)
 *   "device_loop_wide" 1 = the device-resident loop also serves 129 <= K <= 256 (Newton system by a blocked Cholesky
 *                    factorisation in device memory; default); 0 = host-driven loop there
 *   "wide_pmode"     1 = 129 <= K <= 256 also keep a resident probability matrix and run ONE fused sweep per iteration
 *                    (k_fused_quad; default); 0 = two sweeps on u there (what runs when P does not fit)
 *   "quad_trim"      1 = 129 .. 160 and 193 .. 224 states: the one-read Gram / fused sweeps skip the two padding blocks of the
 *                    192- / 256-row panel (default); 0 = the whole panel
 *   "host_pmode"     above 256 states the host-driven loop also builds a resident probability matrix once per solve and runs
 *                    BOTH of its sweeps on it (one multiplication per element instead of an exponential): 2 = Gram sweep in
 *                    256-state panels, one read each, + 128 x 256 rectangles (default); 1 = in the 128-state panels and 64 x 128
 *                    rectangles of the sweep on u; 0 = both sweeps on u (what runs when P does not fit)
 *   "rect_waves"     8 = the 128 x 256 rectangles of that Gram sweep run two waves per SIMD (default); 4 = one
 *   "newton_ldlt"    up to 128 states: 1 = the K x K Newton solve of the device-resident loop (mbar_solvers.py:581-583) is a blocked
 *                    LDL^T factorisation on the fp64 matrix cores, pivots from the last state upwards, two barriers per 16 pivots
 *                    (default; k_select_newton 66 -> 37 us at 127 unknowns); 0 = the register Gauss-Jordan solve of rounds 2-5
 *                    (also: environment variable MBAR_NEWTON_LDLT)
 *   "pcache"         1 = the resident probability matrix outlives the solve that built it: a later adaptive solve on the same
 *                    matrix whose start lies within 200 kT of its anchor (bootstrap replicates, protocol stages) starts with
 *                    one fused sweep instead of the build sweep (default); 0 = every solve builds (cold-solve timings)
 *   "merge_select"   1 = the selection of iteration i and the Newton solve of iteration i + 1 share a launch (default)
 *   "light_last"     fused loop: when BOTH candidates of the coming sweep already meet the stop test of
 *                    mbar_solvers.py:627-636 against the current f, the iteration is the last whichever of them wins and the Gram
 *                    matrix the fused sweep would accumulate is never used: the plain two-candidate sweep on P evaluates them
 *                    instead (k_psweep, launched behind the fused sweep every iteration and idle otherwise; at config 3 1.9 ms
 *                    in place of 3.1; 129 .. 256 states: an evaluation-only body of k_fused_quad, no extra launch).  1 = from 65
 *                    states and 5e7 matrix entries per rank on (default: elsewhere the
 *                    idle launch per iteration costs more than the lighter sweep saves -- with 64 states and fewer both sweeps
 *                    are HBM-bound), 2 = always, 0 = never
 *   "direct_results" 1 = mbar_eval on one rank without the Gram matrix: the last reduction level writes the sums into pinned host
 *                    memory itself instead of a device buffer + a copy (default); 0 = always through the device buffer
 *   "sci_merged"     1 = pure self-consistent iteration, K <= 32, one rank: update + sweep of an iteration in ONE launch
 *                    (k_sci_small; default); 0 = sweep + single-workgroup update kernel (what several ranks and K > 32 run)
 *   "graph", "sci_batch"             hipGraph batching of the solver loops
 *   "timing"         HIP-event timers (mbar_ctx_timing): 0 = off (default: an event pair per sweep costs ~10 us, a fifth of an
 *                    iteration at the problem sizes pymbar is mostly used on), 1 = event records around a launch, 2 = events
 *                    bound to the kernel dispatch in the device-resident loop (no marker packets between kernels), 3 = level 1 plus
 *                    event pairs around the reduction, the all-reduce and the Newton / selection launches of every iteration of
 *                    the device-resident loop (the per-iteration split {sweep, reduce, all-reduce, Newton + selection} that
 *                    explains a multi-GPU run; ~6 more marker packets per iteration) */
int mbar_ctx_set_option(mbar_ctx* ctx, const char* key, int64_t value);

/* ---- data ---------------------------------------------------------------------------------- */
/* Copy columns [col0_host, col0_host+ncols) of a C-contiguous host matrix u_host[K][ld_host]
 * (the u_kn of mbar_solvers.py:174-203, row pitch ld_host) into device columns
 * [col0_dev, col0_dev+ncols) of this rank's shard. */
int mbar_ctx_upload_u(mbar_ctx* ctx, const double* u_host, int64_t ld_host, int64_t col0_host,
                      int64_t ncols, int64_t col0_dev);
int mbar_ctx_download_u(mbar_ctx* ctx, double* out, int64_t ld_out);
/* Row-level assembly of an AUGMENTED matrix on the device -- the expectation family (mbar.py:732-1001) appends one
 * row per new state (u_ln) and one per observable (u_ln - log A_n) to the resident u_kn; none of it needs the host
 * N x (K + NL + S) array the reference builds (mbar.py:886-903).
 *   upload_rows: whole rows [row0, row0 + nrows) from a C-contiguous host array rows_host[nrows][ld_host >= N_local];
 *   copy_rows:   device-to-device from another context on the same device with the same N_local;
 *   row_sub:     u[row][n] -= v_host[n]  (v = log A_n); v_host = NULL subtracts the vector of the previous call again;
 *   rows_sub:    u[dst_row0 + r][n] = u[src_row0 + r][n] - v_host[n] for r < nrows in ONE launch (one observable evaluated at
 *                a run of states: the state rows are copied and shifted in the same pass); dst_row0 == src_row0: in place,
 *                otherwise the ranges must not overlap; v_host = NULL as for row_sub;
 *   rows_rsub:   u[dst_row0 + r][n] = u[src_row0 + r][n] - u[dst_row0 + r][n]: many DIFFERENT observables -- their log A rows are
 *                uploaded in one upload_rows call and turned into observable rows in one launch (disjoint ranges). */
int mbar_ctx_upload_rows(mbar_ctx* ctx, int64_t row0, int64_t nrows, const double* rows_host, int64_t ld_host);
int mbar_ctx_copy_rows(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows);
int mbar_ctx_row_sub(mbar_ctx* ctx, int64_t row, const double* v_host);
int mbar_ctx_rows_sub(mbar_ctx* ctx, int64_t dst_row0, int64_t src_row0, int64_t nrows, const double* v_host);
int mbar_ctx_rows_rsub(mbar_ctx* ctx, int64_t dst_row0, int64_t src_row0, int64_t nrows);
/* Observables into log space ON THE DEVICE (mbar.py:858-867: every observable is shifted to be positive; :886-903: its log
 * enters the augmented log-weight matrix): rows [row0, row0 + nrows) hold raw observable values A_i[n] (uploaded with
 * mbar_ctx_upload_rows, or copied from a resident matrix with mbar_ctx_copy_rows) and leave as log(A_i[n] - shift_i) with
 * shift_i = min_n A_i[n] - |4 eps min_n A_i[n]| (shift_out[i]; the reference's A_min - logfactor, which the host adds back to
 * the expectation) -- in place of three host passes over nrows x N doubles and nrows N libm logarithms.  mbar_ctx_rows_rsub then
 * turns them into observable rows u - log A.  Single-context matrices only (the minimum is not reduced across ranks). */
int mbar_ctx_rows_logshift(mbar_ctx* ctx, int64_t row0, int64_t nrows, double* shift_out);
/* The same for ONE observable evaluated at many states: A[N_local] (host) is uploaded into the context's staging vector and
 * becomes log(A - shift) there; mbar_ctx_rows_sub / mbar_ctx_row_sub with v = NULL then subtract it. */
int mbar_ctx_vec_logshift(mbar_ctx* ctx, const double* A, double* shift_out);
/* Free-energy-surface histograms (fes.py:1383-1402: one extra column of W per populated bin, W[n, K+i] = exp(log_w_n + f_i)
 * on the bin's samples and 0 elsewhere): rows [row0, row0 + nrows) of the augmented matrix become
 *   u[row0 + i][n] = label[n] == i ? v[n] : +inf      (v = the target potential u_n; label[n] = bin of sample n, -1 = none)
 * built on the device from ONE vector and ONE label array -- the N x (K + nbins) host matrix of the reference is never
 * formed.  A +inf entry is a sample of weight zero in that state. */
int mbar_ctx_fill_masked_rows(mbar_ctx* ctx, int64_t row0, int64_t nrows, const double* v_host, const int32_t* label_host);
/* Fill the shard on the device with the synthetic harmonic ladder of SURVEY.md 8(d):
 * global sample n (n_global0 <= n < n_global0+N_local) belongs to the state given by the
 * cumulative N_k_global, x_n ~ Normal(O_s, K_s^-1/2) from a counter-based RNG keyed by
 * (seed, n), u[l][n] = K_l (x_n - O_l)^2 / 2.  Same data for any sharding of n. */
int mbar_ctx_generate_harmonic(mbar_ctx* ctx, uint64_t seed, const double* O_k, const double* K_k,
                               const int64_t* N_k_global, int64_t n_global0);
/* GLOBAL sample counts per state as doubles (the float cast of mbar_solvers.py:792); zeros allowed:
 * such states contribute nothing to denominators (mbar_solvers.py:238 with b=N_k). */
int mbar_ctx_set_Nk(mbar_ctx* ctx, const double* N_k);

/* Per-sample multiplicities c_n >= 0 for this rank's N_local samples (NULL restores c_n = 1).  Every sum over
 * samples becomes sum_n c_n (...): a bootstrap replicate (mbar.py:417-449, resampling the columns of u_kn within each
 * state) is the vector of draw counts, so replicates re-use the resident matrix instead of a gathered copy. */
int mbar_ctx_set_sample_weights(mbar_ctx* ctx, const double* c_n);

/* A bootstrap replicate DRAWN ON THE DEVICE (extension: the reference draws with numpy's generator on the host, one pass over
 * N integers + a bincount per replicate -- 33 ms of host work at N = 4e6 where the replicate's solve takes 5): the multiplicities
 * c_n of mbar_ctx_set_sample_weights become the draw counts of replicate `replicate` of the counter-based stream `seed` -- slot j
 * (position j in state order, 0 <= j < cumN[K_states]) of the state that owns positions cumN[k] .. cumN[k+1]-1 draws one of
 * those positions, a pure function of (seed, replicate, j); order[p] (or NULL: p itself) is the sample at position p, for
 * x_kindices other than the default; n_global0 = first global sample of this rank's shard.  mbar_bootstrap_draws returns the
 * SAME draws on the host as the reference's bootstrap_rints row (mbar.py:433: rints[sample of slot j] = sample drawn): host only,
 * no context.  Same-seed determinism (tests/test_mbar.py:533-545 of the reference) holds by construction. */
int mbar_ctx_draw_bootstrap_weights(mbar_ctx* ctx, uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states,
                                    const int64_t* order, int64_t n_global0);
/* The layout (cumN[K_states + 1], order[cumN[K_states]] or NULL) validated and uploaded ONCE; mbar_ctx_draw_bootstrap_weights with
 * cumN = NULL (K_states, order ignored / NULL) then draws on it -- no host pass over N integers per replicate (with arrays in the
 * call their content is digested on every call to see whether the device copy is still the right one). */
int mbar_ctx_set_bootstrap_layout(mbar_ctx* ctx, const int64_t* cumN, int64_t K_states, const int64_t* order);
int mbar_bootstrap_draws(uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states, const int64_t* order,
                         int64_t* rints_out);

/* Per-sample weights c_n = (A_n - shift)^power from the observable mbar_ctx_vec_logshift left in the staging vector (as
 * log(A_n - shift)), formed on the device: no upload, no host pass over N doubles.  What it is for: ONE observable evaluated at
 * the resident states (compute_expectations(A_n), mbar.py:1039-1312) needs no augmented matrix at all -- the weight column of
 * "A at state l" is A'_n W_nl up to a constant, so the observable rows' normalisers are mbar_lognum with weights A' and the
 * covariance input of mbar.py:886-903 is the three weighted Gram matrices sum_n A'^p W W^T, p = 0, 1, 2, of the RESIDENT matrix
 * (mbar_gram_w with these weights).  mbar_ctx_set_sample_weights(ctx, NULL) restores c_n = 1. */
int mbar_ctx_weights_from_vec(mbar_ctx* ctx, double power);

/* ---- multi-GPU (one process per GPU; N sharded; one small all-reduce per pass) ------------- */
int mbar_comm_unique_id(void* id128);                 /* rank 0: ncclGetUniqueId (128 bytes)     */
int mbar_ctx_comm_init(mbar_ctx* ctx, const void* id128, int rank, int nranks); /* RCCL over xGMI */
/* Fallback transport: fn(buf, count, op, user) must all-reduce `count` doubles in place
 * (op 0 = sum, 1 = max) across ranks on the host. */
typedef int (*mbar_allreduce_fn)(double* buf, int64_t count, int op, void* user);
/* Replaces an RCCL communicator if one is attached (the two transports never coexist on a context). */
int mbar_ctx_set_host_allreduce(mbar_ctx* ctx, mbar_allreduce_fn fn, void* user, int rank, int nranks);
/* Detach whatever transport is attached (RCCL communicator destroyed, in-process group left); the context is single-rank again.  When
 * RCCL cannot be initialised on EVERY rank, every rank must call this before falling back to the host transport. */
int mbar_ctx_comm_destroy(mbar_ctx* ctx);

/* In-process transport: several contexts of ONE process on ONE device (one caller thread each) all-reduce on their compute
 * streams -- events across the streams, a rendezvous of the caller threads per collective, no host-device synchronisation.
 * It drives the same code as an RCCL communicator (collectives on the stream: the device-resident solver loop runs across
 * the "ranks"), which RCCL itself cannot be made to do on a one-GPU box (it refuses two ranks on one device); results are
 * summed in rank order on every rank, hence bit-identical across ranks.  nranks <= 8; every rank must issue the same calls. */
typedef struct mbar_loopback mbar_loopback;
int mbar_loopback_create(mbar_loopback** out, int nranks);
void mbar_loopback_destroy(mbar_loopback* group);
int mbar_ctx_set_loopback(mbar_ctx* ctx, mbar_loopback* group, int rank);

/* ---- L1 evaluation (replaces mbar_solvers.py L1 functions) -------- */
/* One fused sweep of u_kn for nf (1 or 2) free-energy vectors f[nf][K]:
 *   psum[i][k]   = sum_n p_nk(f_i)              (K per f; 0 for states with N_k = 0)
 *   sumlogden[i] = sum_n logden_n(f_i)  [ - d_n with MBAR_EVAL_USE_OFFSET ]
 *   gram[K][K]   = sum_n p_ni p_nj at f_0       (only with MBAR_EVAL_GRAM; fp64 MFMA)
 * logden_n(f_i) stays on the device in slot i.  Any output pointer may be NULL. */
int mbar_eval(mbar_ctx* ctx, const double* f, int nf, unsigned flags, double* psum,
              double* sumlogden, double* gram);
/* d_n := logden_n(f0): the per-sample offset that replaces precondition_u_kn
 * (mbar_solvers.py:697-707) for the objective; f0 = NULL clears it. */
int mbar_ctx_set_objective_offset(mbar_ctx* ctx, const double* f0);
/* lognum_k = log sum_n exp(-logden_n(f) - u_kn) for ALL K states (unsampled ones included),
 * in log space like mbar_solvers.py:240-241; self_consistent_update = -lognum. */
int mbar_lognum(mbar_ctx* ctx, const double* f, double* lognum);
/* logden_n(f) for this rank's samples (mbar_solvers.py:238). */
int mbar_logden(mbar_ctx* ctx, const double* f, double* out_n);
/* log W_nk = f_k - u_kn - logden_n written as out[k][n] with row pitch ld_out (this rank's
 * shard).  Memory-identical to the reference's F-ordered (N, K) result (mbar_solvers.py:439-449). */
int mbar_logw(mbar_ctx* ctx, const double* f, double* out_kn, int64_t ld_out);
/* The weights themselves, W_kn = exp(log W_kn) (mbar_solvers.py:476-486 `mbar_W_nk`: exp of the log weights, taken on the host
 * there -- 1 s of single-threaded numpy for config 3's 1.28e9 entries), same layout as mbar_logw. */
int mbar_w(mbar_ctx* ctx, const double* f, double* out_kn, int64_t ld_out);
/* gramW[K][K] = sum_n W_ni W_nj and wsum[k] = sum_n W_nk for ALL states (W^T W of
 * mbar.py:1816,1849 and compute_overlap mbar.py:606); fp64 MFMA. */
int mbar_gram_w(mbar_ctx* ctx, const double* f, double* gramW, double* wsum);

/* ---- extension contexts: rows appended to a resident matrix without a copy of it ------------
 * The general path of the expectation family (pymbar/mbar.py:886-903 builds an N x (K + NL + S) host array of log weights;
 * rounds 3-5 of this library copied the resident matrix device to device into a (K + NL + S)-row one per call).  An extension
 * context holds ONLY the new rows (new states, observables as unsampled states) -- same device, samples and row pitch as `base`,
 * its storage a whole number of row pitches away from base's -- and the sweeps below read [rows of base | rows of ext]:
 *   mbar_ctx_create_ext      base of at most 128 states on one rank, 129 .. 256 rows in total (else MBAR_ERR_ARG: the caller
 *                            falls back to an augmented copy); rows are written with mbar_ctx_upload_rows / mbar_ctx_copy_rows /
 *                            mbar_ctx_rows_logshift / mbar_ctx_vec_logshift on the extension and the three calls below;
 *   mbar_ctx_rows_sub_from   dst rows = src rows - v  (src = dst or its base; v as in mbar_ctx_rows_sub);
 *   mbar_ctx_rows_rsub_from  dst rows = src rows - dst rows;
 *   mbar_ctx_rows_obs_from   dst rows = base rows[state_row0 ..) - log(base rows[obs_row0 ..) - shift_r), shift_r as in
 *                            mbar_ctx_rows_logshift, handed back (observables that ARE resident rows: entropy / enthalpy); min_out /
 *                            min_in (or NULL): the rows' minima out of one call and into the next on the same resident rows, which
 *                            then skips the pass that finds them;
 *   mbar_lognum_ext          log normalisers of the extension's rows at f_base (the base's log-denominators / multiplicities);
 *   mbar_gram_w_ext          W^T W of [base | ext] at (f_base, f_ext), (K_base + K_ext)^2 row-major, ONE one-read sweep of both
 *                            matrices (k_gram_quad_split); wsum as in mbar_gram_w (N_k = 0 for the extension's rows).  gram_base
 *                            (or NULL) = mbar_gram_w of the base at f_base, kept by the caller: with it and at most 16 appended
 *                            rows only the new entries are computed (a 16 x 128 rectangle + a 16 x 16 block);
 * Evaluations and solves on an extension context fail with MBAR_ERR_STATE; destroy it before its base. */
int mbar_ctx_create_ext(mbar_ctx** out, mbar_ctx* base, int64_t K_rows);
int mbar_ctx_rows_sub_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows, const double* v_host);
int mbar_ctx_rows_rsub_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows);
int mbar_ctx_rows_obs_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* base, int64_t state_row0, int64_t obs_row0, int64_t nrows,
                           double* shift_out, const double* min_in, double* min_out);
int mbar_lognum_ext(mbar_ctx* ext, mbar_ctx* base, const double* f_base, double* lognum_ext);
int mbar_gram_w_ext(mbar_ctx* ext, mbar_ctx* base, const double* f_base, const double* f_ext, const double* gram_base, double* gramW,
                    double* wsum);

/* ---- solver loops (replace adaptive(), mbar_solvers.py:510-667) ---------------------------- */
typedef struct mbar_solve_result {
    int64_t iterations; /* iterations executed                                   */
    int64_t nr_iter;    /* ... of which Newton-Raphson steps were accepted       */
    int64_t sci_iter;   /* ... of which self-consistent steps were accepted      */
    int32_t success;    /* convergence test of mbar_solvers.py:636 met           */
    int32_t gram_sweeps; /* separate Gram sweeps executed (fused loop: rejected speculations only)     */
    double max_delta;   /* last relative change                                  */
    double gnorm;       /* |g| at the returned f                                 */
    double wall_ms;     /* host wall time of the loop                            */
    int32_t warm_starts; /* device-resident loop entries that re-used the resident probability matrix of an earlier solve */
    int32_t builds;      /* ... that built it (build sweep)                       */
    int32_t light_sweeps; /* fused loop: iterations whose sweep ran WITHOUT the speculated Gram matrix because both candidates already
                           * met the stop test against the current f (the last iteration of a solve; option "light_last") */
    int32_t reserved_;
} mbar_solve_result;

/* Adaptive NR/SCI on the states with N_k > 0 (others are left untouched).  f_inout[K].
 * history (may be NULL): rows of 4 doubles {choice(0 sci,1 nr), |g_sci|, |g_nr|, max_delta}.
 * check_convergence = 0 runs exactly maxiter iterations (benchmarking).
 * Up to 256 states the whole iteration is device-resident: K x K Newton solve (up to 128 states in one workgroup's
 * registers, 129 .. 256 by a blocked Cholesky factorisation in device memory), candidate construction, ONE fused sweep for
 * both candidates' gradients and the next Hessian's Gram matrix, choice and convergence test; the host reads a few control
 * words per batch of iterations (replayed from a hipGraph on a single rank), with ONE ncclAllReduce on the stream per
 * iteration across ranks; when the accepted candidate is not the one the sweep speculated on, the loop pauses and the host
 * enqueues that candidate's Gram sweep.  With the host all-reduce transport, above 256 states, or when the device hands a
 * solve back (Newton system not positive definite, candidates > 250 kT from the anchor of the sweeps, non-finite candidate)
 * the same iteration runs host-driven. */
int mbar_solve_adaptive(mbar_ctx* ctx, double* f_inout, double tol, int64_t maxiter,
                        int64_t min_sc_iter, double gamma, int check_convergence,
                        double* history, int64_t history_rows, mbar_solve_result* result);
/* psum[k] = sum_n p_nk at the f the last mbar_solve_adaptive on this context returned (its gradient is psum - N_k, the
 * all-state self-consistent update of sampled states f - log(psum / N_k)): what mbar_solvers.py:939 and :1012 recompute
 * with one more sweep each after a solve.  MBAR_ERR_STATE when there is none (no solve yet, or the matrix / N_k / the sample
 * weights changed since). */
int mbar_ctx_last_solve_psum(mbar_ctx* ctx, double* psum);
/* Pure self-consistent iteration, device-resident (f update and convergence measure on the
 * device).  K <= 32 on one rank: ONE launch per iteration (the update rides in the prologue of the sweep); odd iterations
 * sweep the tiles in descending order ("sci_pingpong": a sweep starts with what the previous one left in the caches).  The
 * host looks at the relative changes after the first batch ("sci_batch", 16) and then after as many iterations as their
 * geometric decay predicts are left (at most 256); without the convergence test it does not look until the end. */
int mbar_solve_sci(mbar_ctx* ctx, double* f_inout, double tol, int64_t maxiter, int check_convergence,
                   mbar_solve_result* result);

/* ---- measurement --------------------------------------------------------------------------- */
/* Accumulated HIP-event time and launch count of one kernel class since the last reset. */
int mbar_ctx_timing(mbar_ctx* ctx, int which, double* total_ms, int64_t* launches);
int mbar_ctx_timing_reset(mbar_ctx* ctx);
/* Peak-rate micro-benchmark of v_mfma_f64_16x16x4_f64 on this device (TFLOP/s). */
int mbar_mfma_f64_peak(mbar_ctx* ctx, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* MBAR_HIP_H */
