// The solver loops of libmbar_hip.so -- what the reference writes in Python: adaptive() (pymbar/mbar_solvers.py:510-667) host-driven
// and device-resident, and the pure self-consistent iteration (mbar_solvers.py:231-242 in a loop).
#include "mbar_ctx.h"

using namespace mbar;
using namespace mbar::host;

namespace mbar {
namespace host {

// ---- adaptive loop -------------------------------------------------------------------------------
// Host-driven loop (mbar_solvers.py:575-640): the K x K solve, the candidate construction and the convergence test run
// on the host between the two sweeps.  Used with the host all-reduce transport, for more than 128 states, for the
// non-default kernel variants, and as the continuation when the device-resident loop hands a solve back.
// `res` carries the counters of the iterations already executed; `f` in/out; psum at the returned f in `psum`.
int adaptive_host_loop(mbar_ctx* c, std::vector<double>& f, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                       int check_convergence, double* history, int64_t history_rows, mbar_solve_result& res,
                       std::vector<double>& psum, double& max_delta) {
    const int64_t K = c->K;
    const int m = (int)c->sampled.size();
    const int first = c->sampled[0];
    std::vector<double> f_old(K), cand(2 * (size_t)K), psum2(2 * (size_t)K);
    std::vector<double> H((size_t)m * m), g(m), x;
    std::vector<int> pos((size_t)K, -1);  // position of a state among those with samples
    for (int i = 0; i < m; ++i) pos[(size_t)c->sampled[i]] = i;
    psum.assign(K, 0.0);
    int cur = 0;  // logden slot of the current f
    // initial gradient (mbar_solvers.py:570)
    int rc = eval_core(c, f.data(), 1, 0, c->logden[cur], nullptr, psum.data(), nullptr, nullptr);
    if (rc) return rc;
    bool done = false;
    const GramPlan plan = plan_for(c);
    const bool dbg = std::getenv("MBAR_DEBUG_TIMING") != nullptr;
    // P mode of the host-driven loop (257 .. 1024 states, where this loop is the only one): the paneled Gram sweep on u spends 12
    // row-blocks of exponentials per 32 matrix instructions (0.51-0.55 of the matrix peak).  With a resident probability matrix
    // P = exp(a0 - u - logden(a0)) built once at the start point its operands are P_kn / s_n -- ONE multiplication -- with
    // s_n = sum_k P_kn exp(a_k - a0_k) = exp(logden_n(f) - logden_n(a0)) from the log-denominators the evaluation sweep on u leaves
    // anyway; the per-state factors exp(a_k - a0_k) are applied to the K x K result on the host.  One more K x N array; if it does
    // not fit, or a state moves more than 250 kT from the anchor (then: a new anchor), the classic sweep runs.
    const int64_t Kp = c->Kp;
    const bool hp_possible = c->opt_pmode && c->opt_host_pmode && Kp > 256;  // (options: the same on every rank)
    bool hp = hp_possible && !c->P_failed;
    // ("host_pmode" 2, the default: 256-state panels and 128 x 256 rectangles; 1: the 128-state panels of the sweep on u)
    const GramPlan plan_p = hp ? (c->opt_host_pmode >= 2 ? gram_plan_pmode(Kp) : plan) : GramPlan();
    std::vector<double> a0, an_host((size_t)Kp), cm((size_t)Kp, 1.0);
    auto anchor_here = [&]() -> int {  // P at the current f (whose log-denominators are in slot `cur`)
        if (!c->P && cache_malloc((void**)&c->P, (size_t)Kp * c->ld * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            c->P = nullptr;
            c->P_failed = true;
            hp = false;
            return MBAR_OK;
        }
        if (!c->pm_ld0) HIPCHK(c, cache_malloc((void**)&c->pm_ld0, (size_t)c->ld * sizeof(double)));
        if (!c->lden_eff) {
            HIPCHK(c, cache_malloc((void**)&c->lden_eff, (size_t)c->ld * sizeof(double)));
            HIPCHK(c, hipMemsetAsync(c->lden_eff, 0, (size_t)c->ld * sizeof(double), c->stream));
        }
        a0.assign((size_t)Kp, 0.0);
        build_aden(c, f.data(), a0.data(), Kp);
        std::copy(a0.begin(), a0.end(), c->hstage);
        HIPCHK(c, hipMemcpyAsync(d_aden(c), c->hstage, (size_t)Kp * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, launch_make_p(c->stream, c->num_cu, c->u, c->ld, c->N, Kp, d_aden(c), c->logden[cur], c->P));
        HIPCHK(c, hipMemcpyAsync(c->pm_ld0, c->logden[cur], (size_t)c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (hstage is re-used below)
        c->P_valid = false;  // (not the device-resident loop's warm-start matrix)
        return MBAR_OK;
    };
    if (hp) {
        rc = anchor_here();
        if (rc) return rc;
    }
    if (hp_possible && c->nranks > 1) {  // a rank whose matrix did not fit must not sweep u while its peers sweep P: the reduced
        bool ok = hp;                    // blocks differ by the per-state factors (and, with the 256-state panels, in number)
        rc = agree_all_ok(c, ok);
        if (rc) return rc;
        hp = ok;
    }
    // Pass B on the same matrix: an element of candidate 0 is P_kn exp(a_k - a0_k) -- one multiplication, no exponential, no
    // shared shift (the eight waves of a workgroup meet once per tile instead of twice) -- and logden_n = logden_n(a0) + log(sum).
    // `used` stays false (the classic sweep on u runs) when a candidate is not finite or outside the 250 kT window of the anchor.
    auto pass_b_on_p = [&](const double* cands, double* ldA, double* ldB, double* ps2, bool& used) -> int {
        used = false;
        if (!hp || !split_sweep_ok(c, Kp) || c->u_poison) return MBAR_OK;
        std::vector<double> h((size_t)2 * Kp, 0.0), a1((size_t)Kp);
        build_aden(c, cands, an_host.data(), Kp);
        build_aden(c, cands + K, a1.data(), Kp);
        for (int64_t k = 0; k < Kp; ++k) {
            if (std::isinf(a0[k])) {  // a state without samples (or padding): no element
                if (!std::isinf(an_host[k]) || !std::isinf(a1[k])) return MBAR_OK;
                continue;
            }
            const double d0 = an_host[k] - a0[k], d1 = a1[k] - an_host[k];
            if (!(std::fabs(d0) < 250.0) || !(std::fabs(d1) < 300.0)) return MBAR_OK;
            h[(size_t)k] = std::exp(d0);
            h[(size_t)Kp + k] = std::exp(d1);
        }
        const size_t n_ps = (size_t)2 * Kp, total = n_ps + 2;
        int r2 = ensure_red(c, total);
        if (r2) return r2;
        r2 = ensure(c, &c->part, &c->part_doubles, (size_t)c->num_cu * 2 * (Kp + 1));
        if (r2) return r2;
        r2 = ensure(c, &c->scratch, &c->scratch_doubles, (size_t)(c->num_cu / 32 + 2) * 2 * (Kp + 1));
        if (r2) return r2;
        std::copy(h.begin(), h.end(), c->hstage);
        HIPCHK(c, hipMemcpyAsync(d_aden(c), c->hstage, h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (ldA == c->logden[0] || ldB == c->logden[0]) c->ld0_valid = false;
        int blocks = 0;
        double* obj_part = c->part + (size_t)c->num_cu * 2 * Kp;
        {
            ScopedTimer t(c, MBAR_TIMER_LSE);
            HIPCHK(c, launch_lse_split(c->stream, c->num_cu, 2, c->P, c->ld, c->N, Kp, d_aden(c), c->cw, ldA, ldB, nullptr, c->part, obj_part,
                                       &blocks, c->pm_ld0));
        }
        {
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce(c->stream, c->part, blocks, (int64_t)n_ps, c->scratch, c->red));
            HIPCHK(c, launch_reduce(c->stream, obj_part, blocks, 2, c->scratch, c->red + n_ps));
        }
        r2 = allreduce_dev(c, c->red, (int64_t)total, 0);
        if (r2) return r2;
        HIPCHK(c, hipMemcpyAsync(c->hred, c->red, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        r2 = sync_stream(c);
        if (r2) return r2;
        for (int64_t k = 0; k < K; ++k) {
            ps2[k] = c->hred[(size_t)k];
            ps2[(size_t)K + k] = c->hred[(size_t)Kp + k] * h[(size_t)Kp + k];  // (the second candidate's sums come without its ratio)
        }
        used = true;
        return MBAR_OK;
    };
    double tA = 0, tH = 0, tB = 0, tA_unpack = 0, tH_solve = 0;
    const int64_t it0 = res.iterations;
    const double t0 = now_ms();
    for (int64_t it = it0; it < maxiter && !done; ++it) {
        // ---- pass A: Gram at f with the known logden -> Hessian (mbar_solvers.py:581) ----
        const double t_a0 = now_ms();
        {
            const GramPlan& pl = hp ? plan_p : plan;
            const size_t n_gram = pl.total_blocks * 256, total = n_gram;
            rc = ensure_red(c, total);
            if (rc) return rc;
            std::vector<double> an((size_t)c->Kp);
            build_aden(c, f.data(), an.data(), c->Kp);
            if (hp) {  // multipliers relative to the anchor; a state that has moved too far re-anchors P at the current f
                bool far = false;
                for (int64_t k = 0; k < Kp; ++k) {
                    const bool live = !std::isinf(an[k]) && !std::isinf(a0[k]);
                    const double d = live ? an[k] - a0[k] : 0.0;
                    if (!(std::fabs(d) < 250.0) || std::isinf(an[k]) != std::isinf(a0[k])) far = true;
                    cm[k] = live ? std::exp(d) : 0.0;
                }
                if (far) {
                    rc = anchor_here();
                    if (rc) return rc;
                    for (int64_t k = 0; k < Kp; ++k) cm[k] = std::isinf(a0[k]) ? 0.0 : 1.0;
                }
            }
            if (hp) {
                HIPCHK(c, launch_rinv_from_logden(c->stream, c->pm_ld0, c->logden[cur], c->cw, c->weighted, c->N, c->lden_eff));
                rc = run_gram(c, d_anum(c), c->lden_eff, 0, pl, c->P);
            } else {
                std::copy(an.begin(), an.end(), c->hstage + 2 * c->Kp);
                HIPCHK(c, hipMemcpyAsync(d_anum(c), c->hstage + 2 * c->Kp, an.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
                rc = run_gram(c, d_anum(c), c->logden[cur], 0, pl);
            }
            if (rc) return rc;
            rc = allreduce_dev(c, c->red, (int64_t)total, 0);
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->hred, c->red, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            rc = sync_stream(c);
            if (rc) return rc;
            const double t_u0 = dbg ? now_ms() : 0.0;
            // H = -Gram (with the per-state factors the P-mode sweep leaves out), m x m over the sampled states; + diag below
            unpack_gram_to_hessian(pl, c->hred, K, hp ? cm.data() : nullptr, pos.data(), m, H.data(), host_team_size(m));
            if (dbg) tA_unpack += now_ms() - t_u0;
        }
        const double t_a1 = now_ms();
        for (int i = 0; i < m; ++i) {
            const int ki = c->sampled[i];
            g[i] = psum[ki] - c->Nk[ki];
            H[(size_t)i * m + i] += psum[ki];
        }
        const double t_s0 = dbg ? now_ms() : 0.0;
        newton_direction(H, g, m, x);  // :582-583
        if (dbg) tH_solve += now_ms() - t_s0;
        double* f_sci = cand.data();
        double* f_nr = cand.data() + K;
        std::copy(f.begin(), f.end(), f_sci);
        std::copy(f.begin(), f.end(), f_nr);
        bool underflow = false;
        for (int i = 0; i < m; ++i) {
            const int k = c->sampled[i];
            f_nr[k] = f[k] - gamma * x[i];                         // :584
            f_sci[k] = f[k] - std::log(psum[k] / c->Nk[k]);        // :587 via s_k
            if (!(psum[k] > 1e-290)) underflow = true;
        }
        if (underflow) {
            // A state whose weights at the current f are below the fp64 range (a start more than ~700 kT from the answer):
            // its sum p underflowed, the reference's log-space update (:240-241) does not.  Take that path for this
            // iteration: the all-state log-space reduction (two more sweeps; slot 0 holds logden(f) again or is about
            // to be overwritten by pass B anyway).
            std::vector<double> ln((size_t)K);
            rc = mbar_lognum(c, f.data(), ln.data());
            if (rc) return rc;
            for (int i = 0; i < m; ++i) f_sci[c->sampled[i]] = -ln[(size_t)c->sampled[i]];
        }
        const double shift = f_sci[first];
        for (int i = 0; i < m; ++i) f_sci[c->sampled[i]] -= shift;  // :588
        // ---- pass B: both candidates in one sweep (:589-594) ----
        const double t_b0 = now_ms();
        const int sA = (cur + 1) % 3, sB = (cur + 2) % 3;
        bool on_p = false;
        rc = pass_b_on_p(cand.data(), c->logden[sA], c->logden[sB], psum2.data(), on_p);
        if (rc) return rc;
        if (!on_p) {
            rc = eval_core(c, cand.data(), 2, 0, c->logden[sA], c->logden[sB], psum2.data(), nullptr, nullptr);
            if (rc) return rc;
        }
        const double t_b1 = now_ms();
        tA += t_a1 - t_a0; tH += t_b0 - t_a1; tB += t_b1 - t_b0;
        double gn_sci = 0.0, gn_nr = 0.0;
        for (int i = 0; i < m; ++i) {
            const int k = c->sampled[i];
            const double a = psum2[k] - c->Nk[k], b = psum2[K + k] - c->Nk[k];
            gn_sci += a * a;
            gn_nr += b * b;
        }
        f_old = f;
        int choice;
        // (every rank holds bit-identical reduced sums, so this choice needs no collective; only the loop exit below
        // is agreed on explicitly, because a desynchronised exit would strand the other ranks in an all-reduce)
        // (:607.  A Newton candidate whose gradient is not a number -- a start so poor that H is numerically zero and the
        // step is of order 1e24 -- loses against a finite self-consistent candidate; the reference's comparison would pick
        // it, but the reference's log-space gradient never produces that NaN in the first place)
        const bool take_sci = gn_sci < gn_nr || (std::isnan(gn_nr) && !std::isnan(gn_sci)) || res.sci_iter < min_sc_iter;
        if (take_sci) {  // :607
            std::copy(f_sci, f_sci + K, f.begin());
            std::copy(psum2.begin(), psum2.begin() + K, psum.begin());
            cur = sA;
            res.sci_iter++;
            choice = 0;
        } else {
            std::copy(f_nr, f_nr + K, f.begin());
            std::copy(psum2.begin() + K, psum2.end(), psum.begin());
            cur = sB;
            res.nr_iter++;
            choice = 1;
        }
        // convergence measures on the sampled states except the first (:627-633)
        const double small = std::min(1e-8, tol);
        max_delta = 0.0;
        double max_diff = 0.0;
        bool nan_seen = false;
        for (int i = 1; i < m; ++i) {
            const int k = c->sampled[i];
            const double div = std::fabs(f[k]) < small ? 1.0 : std::fabs(f[k]);
            const double d1 = std::fabs(f[k] - f_old[k]) / div, d2 = std::fabs(f_sci[k] - f_nr[k]) / div;
            if (std::isnan(d1)) nan_seen = true;
            max_delta = std::max(max_delta, d1);
            max_diff = std::max(max_diff, d2);
        }
        if (nan_seen) max_delta = std::numeric_limits<double>::quiet_NaN();
        res.iterations = it + 1;
        res.gram_sweeps += 1;
        if (history && it < history_rows) {
            history[4 * it + 0] = choice;
            history[4 * it + 1] = std::sqrt(gn_sci);
            history[4 * it + 2] = std::sqrt(gn_nr);
            history[4 * it + 3] = max_delta;
        }
        double stop = (check_convergence && (std::isnan(max_delta) || (max_delta < tol && max_diff < std::sqrt(tol)))) ? 1.0 : 0.0;  // :636
        if (check_convergence) {
            rc = agree_with_rank0(c, &stop, 1);
            if (rc) return rc;
        }
        if (stop > 0.5) {
            res.success = 1;
            done = true;
        }
    }
    const int64_t nit = res.iterations - it0;
    if (dbg && nit > 0)
        std::fprintf(stderr, "[mbar] adaptive (host loop): %lld it, per it: passA %.3f ms (of it unpacking %.3f), host solve %.3f ms (of it the factorisation + substitutions %.3f), passB %.3f ms, total %.3f ms\n",
                     (long long)nit, tA / nit, tA_unpack / nit, tH / nit, tH_solve / nit, tB / nit, (now_ms() - t0) / nit);
    return MBAR_OK;
}

// Device-resident loop: one iteration = {Gram sweep, reduction, [all-reduce], k_newton, two-candidate sweep, reduction,
// [all-reduce], k_select}, enqueued back to back (or replayed from a hipGraph in batches); f, the candidates, the choice
// and the convergence test never leave the device, and the host reads eight control words per batch.  Iterations
// enqueued past convergence are no-ops (every kernel looks at CTL_DONE first).
bool device_loop_eligible(const mbar_ctx* c) {
    if (!c->opt_device_loop || !use_fast(c)) return false;
    if (c->nranks > 1 && !stream_transport(c)) return false;  // the host transport needs the host in the loop
    const int64_t ntiles = (c->N + TS - 1) / TS;
    const LaunchGeom gl = lse_geometry((int)(c->Kp / 16), 2, c->num_cu, ntiles, c->opt_grid, lse_variant_for(c));
    // 129 .. 256 states: the one-read Gram kernel, the four-waves-per-CU evaluation kernel and the blocked Cholesky solve
    if (c->Kp > 128) return c->opt_device_loop_wide && use_quad(c) && gl.variant == 5;
    return gl.variant == 1;
}

inline size_t ad_off_f(const mbar_ctx*) { return 0; }
inline size_t ad_off_psum(const mbar_ctx* c) { return (size_t)c->Kp; }
inline size_t ad_off_cand(const mbar_ctx* c) { return (size_t)2 * c->Kp; }
inline size_t ad_off_ratio(const mbar_ctx* c) { return (size_t)4 * c->Kp; }
inline size_t ad_off_prm(const mbar_ctx* c) { return (size_t)5 * c->Kp; }
inline size_t ad_off_state(const mbar_ctx* c) { return (size_t)5 * c->Kp + 4; }
inline size_t ad_off_hist(const mbar_ctx* c) { return (size_t)5 * c->Kp + 8; }

int ensure_ad(mbar_ctx* c, int64_t hist_rows) {
    const int64_t cap = std::max<int64_t>(1024, std::min<int64_t>(hist_rows, 1 << 20));
    if (!c->ad || c->ad_hist_cap < cap) {
        int rc = drop_graphs(c);
        if (rc) return rc;
        if (c->ad) HIPCHK(c, cache_free(c->ad));
        c->ad = nullptr;
        HIPCHK(c, cache_malloc((void**)&c->ad, (ad_off_hist(c) + (size_t)4 * cap) * sizeof(double)));
        c->ad_hist_cap = cap;
    }
    if (!c->ad_ints) HIPCHK(c, cache_malloc((void**)&c->ad_ints, (size_t)(CTL_WORDS + c->Kp) * sizeof(int)));
    if (!c->h_ctl) HIPCHK(c, cache_host_malloc((void**)&c->h_ctl, (size_t)CTL_WORDS * sizeof(int)));
    return MBAR_OK;
}

// A decision that changes the SEQUENCE of collectives (which sweeps run, which buffers are reduced) must be the same on every
// rank, or the ranks wait for each other in different all-reduces: `ok` is MIN-reduced over the ranks (a collective itself:
// every rank calls it at the same point whatever its local outcome).
int agree_all_ok(mbar_ctx* c, bool& ok) {
    if (c->nranks <= 1) return MBAR_OK;
    double v = ok ? 0.0 : 1.0;
    int rc = allreduce_host(c, &v, 1, 1);
    if (rc) return rc;
    ok = !(v > 0.0);
    return MBAR_OK;
}

// Returns MBAR_OK with handed_back = true when the loop stopped early for the host loop to continue (f, res updated).
int adaptive_device_loop(mbar_ctx* c, std::vector<double>& f, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                         int check_convergence, double* history, int64_t history_rows, mbar_solve_result& res,
                         std::vector<double>& psum, double& max_delta, bool& handed_back) {
    const int64_t K = c->K, Kp = c->Kp;
    const int m = (int)c->sampled.size();
    const int nb = (int)(Kp / 16);
    const int64_t ntiles = (c->N + TS - 1) / TS;
    handed_back = false;
    c->ld0_valid = false;  // (the loop keeps reciprocals / rotating log-denominators in the slot vectors)
    psum.assign(K, 0.0);
    // ---- buffers.  Every allocation of the solve happens here, and the ranks agree on the outcome before the first sweep:
    // a rank that could not get its buffers (or its resident probability matrix) must not wander off into a different
    // sequence of collectives than its peers.
    // P mode: the sweeps run on the resident probability matrix (one more K x N array); if it does not fit ON ANY RANK, or with
    // every rank runs the classic sweeps on u.
    const bool wide = Kp > 128;  // 129 .. 256 states: the one-read kernels whose four waves share a tile stream
    // (129 .. 256 states: P mode exists in its fused form only)
    bool pmode = c->opt_pmode && !c->P_failed && (!wide || (c->opt_wide_pmode && c->opt_fused));
    int arc = ensure_ad(c, history ? history_rows : 0);
    if (!arc && wide && !c->chol && cache_malloc((void**)&c->chol, NEWTON_CHOL_WORK * sizeof(double)) != hipSuccess)
        arc = fail(c, MBAR_ERR_HIP, "allocation of the Newton workspace failed");
    if (!arc && pmode && !c->P) {
        arc = drop_graphs(c);
        if (!arc) {
            if (cache_malloc((void**)&c->P, (size_t)Kp * c->ld * sizeof(double)) != hipSuccess) {
                (void)hipGetLastError();
                c->P = nullptr;
                c->P_failed = true;
                pmode = false;
            } else if (launch_zero(c->stream, c->P, (size_t)Kp * c->ld * sizeof(double)) != hipSuccess) {
                arc = fail(c, MBAR_ERR_HIP, "zero fill of P failed");
            }
        }
    }
    {
        bool p_ok = pmode;
        int rc = agree_all_ok(c, p_ok);
        if (rc) return rc;
        if (pmode && !p_ok) {  // a peer has no room for its P: classic sweeps everywhere (this rank keeps its array for later)
            pmode = false;
            c->error = "resident probability matrix does not fit on every rank: classic sweeps";
        }
    }
    if (!arc && pmode && !c->pm_vec && cache_malloc((void**)&c->pm_vec, (size_t)3 * Kp * sizeof(double)) != hipSuccess)
        arc = fail(c, MBAR_ERR_HIP, "allocation of the P-mode vectors failed");
    const bool fused = pmode && c->opt_fused;
    // Last iteration without its Gram matrix (CTL_LIGHT, mbar_internal.h): an idle launch per iteration against ONE lighter sweep per
    // solve.  Worth it where the fused sweep is bound by the matrix cores and the plain one by HBM -- 65 states and more (K = 128:
    // 1.9 ms against 3.1 at config 3; at 64 states and fewer both are HBM-bound and nothing is gained) -- and from ~5e7 matrix
    // entries per rank on (a sweep of ~0.13 ms); option light_last = 2 drops both bounds.
    // (129 .. 256 states: the one-read fused sweep has an evaluation-only body of its own and needs no stand-in launch)
    bool light = fused && check_convergence && c->opt_light_last != 0 &&
                 (c->opt_light_last >= 2 || (nb >= 5 && (double)Kp * (double)c->N >= 5.0e7));
    // geometry and buffers are fixed for the whole solve (nothing may allocate inside a capture)
    LaunchGeom gg = wide ? gram_quad_geometry(nb, c->num_cu, ntiles, c->opt_grid)
                         : gram_geometry(nb * 16, true, c->num_cu, ntiles, c->opt_grid);
    LaunchGeom gl = fused ? fused_geometry(nb, c->num_cu, ntiles, c->opt_grid)
                    : pmode ? psweep_geometry(nb, c->num_cu, ntiles, c->opt_grid)
                            : lse_geometry(nb, 2, c->num_cu, ntiles, c->opt_grid, lse_variant_for(c));
    if (wide) gg.live_blocks = gl.live_blocks = quad_live_blocks(c);
    if (fused) {  // the separate Gram sweep (when it runs) leaves its partial records where the fused sweep leaves them
        gg.blocks = gl.blocks;
        gg.nwaves = gl.nwaves;
    }
    // the plain sweep that stands in for the fused one leaves ITS per-state records where the fused sweep leaves them too: as many
    // waves as the fused grid has, in workgroups of the plain sweep's size
    LaunchGeom gp = psweep_geometry(wide ? 8 : nb, c->num_cu, ntiles, 0);
    if (light && !wide && gl.nwaves % gp.waves != 0) light = false;
    if (light && !wide) {
        gp.blocks = gl.nwaves / gp.waves;
        gp.nwaves = gp.psum_records = gl.nwaves;
    }
    {
        bool l_ok = light;  // (every rank derives it from its own shard length: agree, like every decision that changes what is launched)
        int rcl = agree_all_ok(c, l_ok);
        if (rcl) return rcl;
        light = light && l_ok;
    }
    // build sweep of P mode: with the fused loop it also accumulates the Gram matrix at the anchor (grid of the fused sweep)
    const LaunchGeom gb = fused ? build_gram_geometry(nb, c->num_cu, ntiles, c->opt_grid)
                                : build_sweep_geometry(nb, c->num_cu, ntiles, c->opt_grid);
    const size_t rec_g = (size_t)nb * (nb + 1) / 2 * 256;
    const size_t rec_l = (size_t)2 * Kp;
    const size_t off_gram = rec_l + 2;
    if (!arc) arc = ensure_red(c, off_gram + rec_g);
    if (!arc)
        arc = ensure(c, &c->part, &c->part_doubles,
                     std::max(std::max((size_t)gg.nwaves * rec_g, (size_t)gl.nwaves * (rec_l + 2)), (size_t)gb.nwaves * Kp));
    // (level-1 scratch of the widest reduction: the fused loop reduces the per-state sums and the Gram records in ONE pair of launches)
    if (!arc)
        arc = ensure(c, &c->scratch, &c->scratch_doubles,
                     std::max(((size_t)std::max(gg.nwaves, gl.nwaves) / 32 + 1) * (rec_g + rec_l + 2), ((size_t)gb.nwaves / 32 + 1) * (Kp + rec_g)));
    if (!arc && c->weighted && !c->lden_eff) arc = fail(c, MBAR_ERR_STATE, "weighted context without its logden buffer");
    if (!arc && fused) arc = ensure(c, &c->part_g, &c->part_g_doubles, (size_t)gl.nwaves * rec_g);
    {
        bool ok = arc == MBAR_OK;
        const std::string local_err = c->error;
        int rc = agree_all_ok(c, ok);
        if (rc) return rc;
        if (arc) return arc;
        if (!ok) return fail(c, MBAR_ERR_STATE, "a peer rank could not allocate its solver buffers");
        c->error = local_err;
    }
    int rc = MBAR_OK;
    LoopCtl lc_slot, lc_flat;
    lc_slot.ctl = lc_flat.ctl = c->ad_ints;
    lc_slot.slot_stride = c->ld;
    lc_slot.unclamped = lc_flat.unclamped = c->u_checked && !c->u_posinf;
    lc_slot.pmode = lc_flat.pmode = pmode;
    // Warm start: the resident probability matrix of an earlier solve on this matrix is still there and the start point lies
    // inside the window of its anchor -- the per-state sums, the reciprocals and the Gram matrix at f come from ONE fused sweep
    // (both multiplier rows = exp(aden(f) - a0)) instead of the build sweep (16 K N bytes of traffic and K N exponentials).
    std::vector<double> an0((size_t)Kp), cm0((size_t)Kp, 0.0);
    build_aden(c, f.data(), an0.data(), Kp);
    bool warm = fused && c->opt_pcache && c->P_valid && (int64_t)c->P_a0.size() == Kp;
    for (int64_t k = 0; warm && k < Kp; ++k) {
        const bool live = !std::isinf(an0[k]), was = !std::isinf(c->P_a0[k]);
        if (live != was) warm = false;
        else if (live) {
            const double d = an0[k] - c->P_a0[k];
            if (!(std::fabs(d) < 200.0)) warm = false;
            cm0[k] = std::exp(d);
        }
    }
    rc = agree_all_ok(c, warm);
    if (rc) return rc;
    // initial gradient (mbar_solvers.py:570).  Classic: the evaluation sweep, logden(f) stays in slot 0.  P mode: the
    // same sweep also writes P = exp(a0 - u - logden(a0)) with a0 = aden(f) and leaves 1 / s = 1 in slot 0; in the fused
    // loop it accumulates the first Hessian's Gram matrix as well (its reduced blocks wait in `red` for k_newton).
    if (!pmode) {
        rc = eval_core(c, f.data(), 1, 0, c->logden[0], nullptr, psum.data(), nullptr, nullptr);
        if (rc) return rc;
    } else if (warm) {
        std::vector<int> z((size_t)CTL_WORDS, 0);  // slot 0, running: the sweep leaves the reciprocals of its first row in slot 1
        HIPCHK(c, hipMemcpyAsync(c->ad_ints, z.data(), z.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        std::copy(cm0.begin(), cm0.end(), c->hstage);
        std::copy(cm0.begin(), cm0.end(), c->hstage + Kp);
        HIPCHK(c, hipMemcpyAsync(d_aden(c), c->hstage, (size_t)2 * Kp * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->red, 0, off_gram * sizeof(double), c->stream));
        {
            ScopedTimer t(c, MBAR_TIMER_OTHER);
            HIPCHK(c, launch_fused(c->stream, nb, gl, c->P, c->ld, c->N, d_aden(c), c->cw, c->weighted ? c->cwsq : c->cw, c->logden[0],
                                   c->part_g, c->part, lc_slot));
        }
        HIPCHK(c, launch_reduce2(c->stream, c->part, (int64_t)rec_l, c->part_g, (int64_t)rec_g, gl.nwaves, c->scratch, c->red,
                                 c->red + off_gram));
        rc = allreduce_dev(c, c->red, (int64_t)(off_gram + rec_g), 0);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->hred, c->red, (size_t)Kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        rc = sync_stream(c);
        if (rc) return rc;
        for (int64_t k = 0; k < K; ++k) psum[k] = c->hred[k] * cm0[k];  // (the sweep returns the sums without the multipliers)
        res.warm_starts += 1;
    } else if (wide) {
        // 129 .. 256 states: the probability matrix from three plain sweeps -- evaluation at the anchor (log-denominators into slot
        // 1, per-state sums), P = exp(a0 - u - logden), and (fused loop) the Gram matrix at the anchor from P with unit reciprocals
        c->P_valid = false;
        rc = eval_core(c, f.data(), 1, 0, c->logden[1], nullptr, psum.data(), nullptr, nullptr);
        if (rc) return rc;
        if (fused && !c->weighted) {
            // unweighted: the Gram sweep at the anchor forms exactly P as its operands -- it writes them out on the way (one sweep
            // instead of make-P + Gram-from-P: 8 K N bytes read + 8 K N written once)
            ScopedTimer t(c, MBAR_TIMER_OTHER);
            HIPCHK(c, launch_gram_quad(c->stream, nb, gg, c->u, c->ld, c->N, d_aden(c), c->logden[1], c->part_g, LoopCtl(), c->P));
            HIPCHK(c, launch_reduce(c->stream, c->part_g, gg.nwaves, (int64_t)rec_g, c->scratch, c->red + off_gram));
            rc = allreduce_dev(c, c->red + off_gram, (int64_t)rec_g, 0);
            if (rc) return rc;
            HIPCHK(c, launch_fill(c->stream, c->logden[0], 1.0, c->ld));
        } else {
            ScopedTimer t(c, MBAR_TIMER_OTHER);
            HIPCHK(c, launch_make_p(c->stream, c->num_cu, c->u, c->ld, c->N, Kp, d_aden(c), c->logden[1], c->P));
            HIPCHK(c, launch_fill(c->stream, c->logden[0], 1.0, c->ld));
        }
        if (fused && c->weighted) {
            const double* lden = c->logden[0];
            if (c->weighted) {
                HIPCHK(c, launch_rinv_weighted(c->stream, c->logden[0], c->cw, c->N, c->lden_eff));
                lden = c->lden_eff;
            }
            LoopCtl lp;
            lp.pmode = true;
            {
                ScopedTimer t(c, MBAR_TIMER_GRAM);
                HIPCHK(c, launch_gram_quad(c->stream, nb, gg, c->P, c->ld, c->N, d_anum(c), lden, c->part_g, lp));
            }
            HIPCHK(c, launch_reduce(c->stream, c->part_g, gg.nwaves, (int64_t)rec_g, c->scratch, c->red + off_gram));
            rc = allreduce_dev(c, c->red + off_gram, (int64_t)rec_g, 0);
            if (rc) return rc;
        }
        rc = sync_stream(c);
        if (rc) return rc;
        c->P_a0 = an0;
        c->P_valid = true;
        res.builds += 1;
    } else {
        c->P_valid = false;
        build_aden(c, f.data(), c->hstage, Kp);
        HIPCHK(c, hipMemcpyAsync(d_aden(c), c->hstage, (size_t)Kp * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->red, 0, off_gram * sizeof(double), c->stream));
        {
            ScopedTimer t(c, MBAR_TIMER_OTHER);
            if (fused)
                HIPCHK(c, launch_build_gram(c->stream, nb, gb, c->u, c->ld, c->N, d_aden(c), c->weighted ? c->cwsq : c->cw,
                                            c->weighted || !lc_slot.unclamped, c->P, c->logden[0], c->part_g));
            else
                HIPCHK(c, launch_build_sweep(c->stream, nb, gb, c->u, c->ld, c->N, d_aden(c), c->cw, c->P, c->logden[0], c->part));
        }
        if (fused) {
            // the Gram matrix at the anchor: one reduction, ONE all-reduce; the per-state sums are its row sums (the rows of p sum to
            // one: sum_n c_n p_kn = sum_j G_kj), taken on the host from the reduced blocks -- the build sweep accumulates none
            HIPCHK(c, launch_reduce(c->stream, c->part_g, gb.nwaves, (int64_t)rec_g, c->scratch, c->red + off_gram));
            rc = allreduce_dev(c, c->red + off_gram, (int64_t)rec_g, 0);
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->hred + off_gram, c->red + off_gram, rec_g * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            rc = sync_stream(c);
            if (rc) return rc;
            gram_row_sums(c->hred + off_gram, (int)nb, K, psum.data());
        } else {
            HIPCHK(c, launch_reduce(c->stream, c->part, gb.nwaves, Kp, c->scratch, c->red));
            rc = allreduce_dev(c, c->red, Kp, 0);
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->hred, c->red, (size_t)Kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            rc = sync_stream(c);
            if (rc) return rc;
            for (int64_t k = 0; k < K; ++k) psum[k] = c->hred[k];
        }
        c->P_a0 = an0;
        c->P_valid = true;
        res.builds += 1;
    }
    double* gram_part = fused ? c->part_g : c->part;

    // ---- solver state to the device ----
    {
        std::vector<double> h(ad_off_hist(c), 0.0);
        for (int64_t k = 0; k < K; ++k) {
            h[ad_off_f(c) + k] = f[k];
            h[ad_off_psum(c) + k] = psum[k];
        }
        h[ad_off_prm(c) + 0] = gamma;
        h[ad_off_prm(c) + 1] = tol;
        h[ad_off_prm(c) + 2] = (double)std::min<int64_t>(min_sc_iter, 1 << 30);
        h[ad_off_prm(c) + 3] = check_convergence ? 1.0 : 0.0;
        h[ad_off_state(c)] = std::numeric_limits<double>::quiet_NaN();
        std::vector<int> hi((size_t)CTL_WORDS + Kp, 0);
        // two-sweep loops and the classic mode run a Gram sweep per iteration; the fused loop starts with the Gram matrix
        // its build sweep accumulated (multipliers cgram = 1 at the anchor)
        hi[CTL_NEEDGRAM] = fused ? 0 : 1;
        hi[CTL_GRAMSWEEPS] = 0;
        hi[CTL_SPEC] = 1;
        hi[CTL_SLOT] = warm ? 1 : 0;
        hi[CTL_ITER] = (int)res.iterations;
        hi[CTL_SCI] = (int)res.sci_iter;
        hi[CTL_NR] = (int)res.nr_iter;
        for (int i = 0; i < m; ++i) hi[CTL_WORDS + i] = c->sampled[i];
        std::vector<double> an((size_t)Kp);
        build_aden(c, f.data(), an.data(), Kp);
        HIPCHK(c, hipMemcpyAsync(c->ad, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ad_ints, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_anum(c), an.data(), an.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (pmode) {  // anchor point a0 (= aden(f) after a build), multipliers of the current f relative to it (1 after a build)
            std::vector<double> pv((size_t)3 * Kp, 1.0);
            std::copy(c->P_a0.begin(), c->P_a0.end(), pv.begin());
            for (int64_t k = 0; k < Kp; ++k) {
                if (warm) pv[(size_t)Kp + k] = pv[(size_t)2 * Kp + k] = cm0[k];
                if (!(k < K && c->Nk[k] > 0.0)) pv[(size_t)Kp + k] = pv[(size_t)2 * Kp + k] = 0.0;
            }
            HIPCHK(c, hipMemcpyAsync(c->pm_vec, pv.data(), pv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    AdaptArgs q;
    q.gram_red = c->red + off_gram;
    q.lse_red = c->red;
    q.f = c->ad + ad_off_f(c);
    q.psum = c->ad + ad_off_psum(c);
    q.cand = c->ad + ad_off_cand(c);
    q.ratio = c->ad + ad_off_ratio(c);
    q.aden = d_aden(c);
    q.anum = d_anum(c);
    q.Nk = d_Nk(c);
    q.lnNk = d_lnNk(c);
    q.sampled = c->ad_ints + CTL_WORDS;
    q.m = m;
    q.K = (int)K;
    q.Kp = (int)Kp;
    q.ctl = c->ad_ints;
    q.prm = c->ad + ad_off_prm(c);
    q.state = c->ad + ad_off_state(c);
    q.hist = c->ad + ad_off_hist(c);
    q.hist_cap = c->ad_hist_cap;
    q.pmode = pmode ? 1 : 0;
    q.a0 = c->pm_vec;
    q.ccur = pmode ? c->pm_vec + Kp : nullptr;
    q.fused = fused ? 1 : 0;
    q.cgram = fused ? c->pm_vec + 2 * Kp : nullptr;
    q.light_ok = light ? 1 : 0;
    q.newton_ldlt = c->opt_newton_ldlt ? 1 : 0;
    if (const char* e = std::getenv("MBAR_NEWTON_LDLT")) q.newton_ldlt = std::atoi(e) != 0 ? 1 : 0;
    q.stamps = nullptr;
    if (std::getenv("MBAR_DEBUG_STAMPS")) {
        if (!c->stamps) HIPCHK(c, hipMalloc((void**)&c->stamps, 65 * 16 * sizeof(long long)));
        HIPCHK(c, hipMemsetAsync(c->stamps, 0, 65 * 16 * sizeof(long long), c->stream));
        q.stamps = c->stamps;
    }

    // Gram sweep at the current f with the known logden (the slot of the accepted candidate; P mode: the slots hold the
    // reciprocals 1 / s_n instead), reduced and all-reduced into the blocks k_newton reads.  Two-sweep loops: once per
    // iteration.  Fused loop: only after a pause (k_select found that the accepted candidate is not the one the sweep
    // speculated on) -- the host enqueues it, un-pausing first.
    auto enqueue_gram = [&](bool timed) -> int {
        const double* lden = c->logden[0];
        LoopCtl lca = lc_slot;
        if (fused) HIPCHK(c, launch_ctl_resume(c->stream, c->ad_ints));
        if (c->weighted) {  // sum_n c_n p p^T: each operand carries sqrt(c_n), folded into the exponent / the reciprocal
            if (pmode)
                HIPCHK(c, launch_rinv_weighted(c->stream, c->logden[0], c->cw, c->N, c->lden_eff, lc_slot));
            else
                HIPCHK(c, launch_shift_logden(c->stream, c->logden[0], c->cw, 0.5, c->N, c->lden_eff, lc_slot));
            lden = c->lden_eff;
            lca = lc_flat;
        }
        {
            // opt_timing 2: the events ride on the kernel dispatch itself; 1: event records around the launch
            TimerPair tp{nullptr, nullptr, MBAR_TIMER_GRAM};
            if (timed) { tp.a = get_event(c); tp.b = get_event(c); }
            const bool ext = tp.a && tp.b && c->opt_timing == 2;
            if (ext) { lca.ev_start = tp.a; lca.ev_stop = tp.b; }
            if (tp.a && tp.b && !ext) (void)hipEventRecord(tp.a, c->stream);
            if (wide)
                HIPCHK(c, launch_gram_quad(c->stream, nb, gg, pmode ? c->P : c->u, c->ld, c->N, d_anum(c), lden, gram_part, lca));
            else
                HIPCHK(c, launch_gram_diag(c->stream, nb, gg, pmode ? c->P : c->u, c->ld, c->N, d_anum(c), lden, 0, gram_part,
                                           nullptr, lca));
            if (tp.a && tp.b && !ext) (void)hipEventRecord(tp.b, c->stream);
            if (tp.a && tp.b) c->pending.push_back(tp);
        }
        HIPCHK(c, launch_reduce(c->stream, gram_part, gg.nwaves, (int64_t)rec_g, c->scratch, c->red + off_gram));
        if (stream_transport(c)) {
            int r2 = allreduce_dev(c, c->red + off_gram, (int64_t)rec_g, 0);
            if (r2) return r2;
        }
        return MBAR_OK;
    };
    // One iteration.  Fused loop: {k_newton, fused sweep, ONE reduction of its per-state sums and Gram records, ONE all-reduce
    // of both, k_select} -- the Gram matrix the next k_newton needs comes out of the same sweep as the gradients.  Two-sweep
    // loops: the Gram sweep first.
    // (fused loop: the Newton solve of an iteration rides in the launch of the previous iteration's selection -- k_select_newton --
    // so the loop proper is four launches per iteration (+ the idle stand-in sweep of light_last); a solve of its own is needed at
    // the start and after a pause)
    const bool merged = fused && !wide && c->opt_merge_select;
    bool need_newton = true;
    // timing level 3: event pairs around the non-sweep sections too (the split that explains a multi-GPU iteration)
    struct Section {
        mbar_ctx* c;
        TimerPair tp;
        Section(mbar_ctx* c_, bool on, int which) : c(c_) {
            tp.a = tp.b = nullptr;
            tp.which = which;
            if (!on) return;
            tp.a = get_event(c);
            tp.b = get_event(c);
            if (tp.a && tp.b) (void)hipEventRecord(tp.a, c->stream);
        }
        ~Section() {
            if (tp.a && tp.b) {
                (void)hipEventRecord(tp.b, c->stream);
                c->pending.push_back(tp);
            }
        }
    };
    auto enqueue_iteration = [&](bool timed) -> int {
        const bool split = timed && c->opt_timing == 3;
        if (!fused) {
            int r2 = enqueue_gram(timed);
            if (r2) return r2;
        }
        {
            Section sec(c, split && (wide || !merged || need_newton), MBAR_TIMER_NEWTON);
            if (wide)
                HIPCHK(c, launch_newton_chol(c->stream, q, c->chol));
            else if (!merged || need_newton)
                HIPCHK(c, launch_newton(c->stream, q));
        }
        need_newton = false;
        double* psum_part = c->part;
        double* obj_part = c->part + (size_t)gl.nwaves * rec_l;
        {
            TimerPair tp{nullptr, nullptr, fused ? MBAR_TIMER_FUSED : MBAR_TIMER_LSE};
            if (timed) { tp.a = get_event(c); tp.b = get_event(c); }
            const bool ext = tp.a && tp.b && c->opt_timing == 2;
            LoopCtl lcb = lc_slot;
            if (ext) { lcb.ev_start = tp.a; lcb.ev_stop = tp.b; }
            if (tp.a && tp.b && !ext) (void)hipEventRecord(tp.a, c->stream);
            if (fused) {
                HIPCHK(c, launch_fused(c->stream, nb, gl, c->P, c->ld, c->N, d_aden(c), c->cw, c->weighted ? c->cwsq : c->cw,
                                       c->logden[0], gram_part, psum_part, lcb));
                if (light && !wide) {  // (idle unless k_newton found that this iteration is the last: then the fused sweep is the idle one)
                    LoopCtl lcl = lc_slot;
                    lcl.light_only = true;
                    HIPCHK(c, launch_psweep(c->stream, nb, 2, gp, c->P, c->ld, c->N, d_aden(c), c->cw, c->logden[0], nullptr, psum_part, lcl));
                }
            } else if (pmode)
                HIPCHK(c, launch_psweep(c->stream, nb, 2, gl, c->P, c->ld, c->N, d_aden(c), c->cw, c->logden[0], nullptr, psum_part,
                                        lcb));
            else
                HIPCHK(c, launch_lse(c->stream, nb, 2, gl, c->u, c->ld, c->N, d_aden(c), c->cw, c->logden[0], nullptr,
                                     nullptr, psum_part, obj_part, lcb));
            if (tp.a && tp.b && !ext) (void)hipEventRecord(tp.b, c->stream);
            if (tp.a && tp.b) c->pending.push_back(tp);
        }
        int64_t ar_count = (int64_t)(rec_l + 2);
        {
            Section sec(c, split, MBAR_TIMER_REDUCE);
            if (fused) {
                HIPCHK(c, launch_reduce2(c->stream, psum_part, (int64_t)rec_l, gram_part, (int64_t)rec_g, gl.nwaves, c->scratch, c->red,
                                         c->red + off_gram));
                ar_count = (int64_t)(off_gram + rec_g);
            } else if (pmode) {  // (no objective sums in P mode: the adaptive loop does not use them)
                HIPCHK(c, launch_reduce(c->stream, psum_part, gl.nwaves, (int64_t)rec_l, c->scratch, c->red));
            } else {
                HIPCHK(c, launch_reduce2(c->stream, psum_part, (int64_t)rec_l, obj_part, 2, gl.nwaves, c->scratch, c->red, c->red + rec_l));
            }
        }
        if (stream_transport(c)) {
            Section sec(c, split, MBAR_TIMER_COMM);
            int r2 = allreduce_dev(c, c->red, ar_count, 0);
            if (r2) return r2;
        }
        {
            Section sec(c, split, MBAR_TIMER_NEWTON);
            if (merged)
                HIPCHK(c, launch_select_newton(c->stream, q));
            else
                HIPCHK(c, launch_select(c->stream, q));
        }
        return MBAR_OK;
    };

    // Batches between two looks at the control words: 6, 2, 4, then `adapt_batch` (8) each.  Iterations enqueued past convergence
    // (or past a pause of the fused loop) are no-ops of ~3.5 us per kernel; real solves take 5-8 iterations, and for the small
    // problems pymbar is mostly used on two wasted iterations of a fixed batch of 8 were a tenth of the solve.  After a pause the
    // batches restart at 1, 2, 4: a phase in which the self-consistent candidate keeps winning pauses every iteration.  Only
    // full-size batches replay a captured hipGraph (eager launches are as fast at these kernel counts: the queue never runs
    // dry), so a short solve never pays for a capture.
    const int64_t batch = c->opt_adapt_batch;
    const bool use_graph = c->opt_graph && !stream_transport(c);
    auto prepare_graph = [&]() -> int {
        const int64_t sig = ((int64_t)gg.blocks << 40) ^ ((int64_t)gl.blocks << 20) ^ ((int64_t)m << 12) ^ (pmode ? 128 : 0) ^ (fused ? 256 : 0) ^
                            (c->weighted ? 64 : 0) ^ (lc_slot.unclamped ? 512 : 0) ^ (merged ? 1024 : 0) ^ (light ? 2048 : 0) ^ (int64_t)nb;
        if (!c->ad_graph || c->ad_graph_batch != batch || c->ad_graph_sig != sig) {
            // (the captured iterations are the steady-state ones: no Newton solve of their own when it rides with the selection)
            const bool need_saved = need_newton;
            need_newton = false;
            struct Restore { bool& r; bool v; ~Restore() { r = v; } } restore{need_newton, need_saved};
            if (c->ad_graph) HIPCHK(c, hipGraphExecDestroy(c->ad_graph));
            c->ad_graph = nullptr;
            // eager warm-up with the stop flag raised: every kernel is launched once outside the capture (function
            // attributes, module loading) and does nothing
            int one = 1;
            HIPCHK(c, hipMemcpyAsync(c->ad_ints + CTL_DONE, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
            rc = enqueue_iteration(false);
            if (rc) return rc;
            int zero = 0;
            HIPCHK(c, hipMemcpyAsync(c->ad_ints + CTL_DONE, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipGraph_t graph = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            int crc = MBAR_OK;
            for (int64_t b = 0; b < batch && crc == MBAR_OK; ++b) crc = enqueue_iteration(false);
            hipError_t ee = hipStreamEndCapture(c->stream, &graph);
            if (crc) return crc;
            if (ee != hipSuccess) return fail(c, MBAR_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
            ee = hipGraphInstantiate(&c->ad_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ee != hipSuccess) return fail(c, MBAR_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ee));
            c->ad_graph_batch = batch;
            c->ad_graph_sig = sig;
        }
        return MBAR_OK;
    };
    int64_t it = res.iterations;
    const int64_t it_start = it;
    bool done = false;
    int64_t nbatch = 0, ramp = batch;  // ramp: cap on the batch size while recovering from a pause
    int32_t gram_sweeps = 0;
    while (it < maxiter && !done) {
        static const int64_t first_batches[3] = {6, 2, 4};
        // (a fixed number of iterations -- no convergence test -- has nothing to look for early: full batches from the start, and up
        // to four of them between two looks at the control words; a pause or a hand-back turns the rest into no-op launches as ever)
        const bool fixed_count = !check_convergence;
        const int64_t want = std::min(ramp, (nbatch < 3 && !fixed_count) ? std::min(batch, first_batches[nbatch]) : batch);
        ++nbatch;
        ramp = std::min(batch, ramp * 2);
        int64_t nbat = 0;
        for (int rep = 0; rep < (fixed_count && ramp == batch ? 4 : 1) && it + nbat < maxiter; ++rep) {
            const int64_t nb1 = std::min(want, maxiter - it - nbat);
            if (use_graph && nb1 == batch) {
                rc = prepare_graph();
                if (rc) return rc;
                if (merged && need_newton) {  // (start of the solve / after a pause: the replayed iterations have no solve of their own)
                    HIPCHK(c, launch_newton(c->stream, q));
                    need_newton = false;
                }
                HIPCHK(c, hipGraphLaunch(c->ad_graph, c->stream));
            } else {
                for (int64_t b = 0; b < nb1; ++b) {
                    rc = enqueue_iteration(c->opt_timing != 0);
                    if (rc) return rc;
                }
            }
            nbat += nb1;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ad_ints, CTL_WORDS * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        rc = sync_stream(c);
        if (rc) return rc;
        const int64_t it_new = c->h_ctl[CTL_ITER];
        if (c->h_ctl[CTL_DONE] == 1) {
            res.success = 1;
            done = true;
        } else if (c->h_ctl[CTL_DONE] == 2) {
            handed_back = true;
            done = true;
        } else if (c->h_ctl[CTL_DONE] == 3) {
            // the fused loop paused itself after iteration it_new (the rest of the batch were no-ops): the accepted candidate's
            // Gram matrix has to be swept separately.  Every rank sees the same control words, so every rank comes by here.
            if (it_new <= it || it_new > it + nbat) return fail(c, MBAR_ERR_STATE, "device-resident adaptive loop lost count of its iterations");
            if (it_new < maxiter) {
                rc = enqueue_gram(c->opt_timing != 0);
                if (rc) return rc;
                ++gram_sweeps;
                ramp = 1;
                need_newton = true;  // (the solve that rode with the selection returned on the pause flag)
            }
        } else if (it_new != it + nbat) {
            return fail(c, MBAR_ERR_STATE, "device-resident adaptive loop lost count of its iterations");
        }
        it = it_new;
    }
    // ---- results back ----
    {
        std::vector<double> h(ad_off_hist(c));
        HIPCHK(c, hipMemcpyAsync(h.data(), c->ad, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        // (only the rows of the iterations that ran HERE: after a hand-back the host loop wrote rows of its own in between)
        const int64_t row1 = history ? std::min<int64_t>(std::min<int64_t>(it, history_rows), c->ad_hist_cap) : 0;
        if (row1 > it_start)
            HIPCHK(c, hipMemcpyAsync(history + 4 * it_start, c->ad + ad_off_hist(c) + 4 * it_start, (size_t)(row1 - it_start) * 4 * sizeof(double),
                                     hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < K; ++k) {
            f[k] = h[ad_off_f(c) + k];
            psum[k] = h[ad_off_psum(c) + k];
        }
        if (it > res.iterations) max_delta = h[ad_off_state(c)];
    }
    if (q.stamps) {
        std::vector<long long> st(65 * 16);
        HIPCHK(c, hipMemcpy(st.data(), c->stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i) {
            const long long* p = st.data() + 16 * i;
            if (!p[0] || !p[5]) continue;
            std::fprintf(stderr, "[mbar] k_select_newton launch %d (shader clocks): select %lld, set-up %lld, elimination %lld, solution %lld, candidates %lld, total %lld; of the elimination: pivot loops %lld, exchange + trailing update %lld; of the selection: inputs %lld, Gram staged %lld, matrix-vector product %lld, sums %lld, maxima %lld\n",
                         i, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[5] - p[0], p[6], p[7], p[8] - p[0], p[9] - p[8], p[10] - p[9],
                         p[11] - p[10], p[12] - p[11]);
        }
    }
    res.iterations = it;
    res.sci_iter = c->h_ctl[CTL_SCI];
    res.nr_iter = c->h_ctl[CTL_NR];
    res.gram_sweeps += fused ? gram_sweeps : (int32_t)(it - it_start);
    res.light_sweeps += c->h_ctl[CTL_LIGHTS];
    if (handed_back) {
        c->P_valid = false;  // (the continuation re-anchors: a state whose weights underflow at this anchor has a zero row in P)
        static const char* why[] = {"", "the Newton system is not positive definite", "a candidate is too far from the point the sweeps are anchored at",
                                    "a candidate is not finite"};
        const int r = c->h_ctl[CTL_REASON];
        c->error = std::string("device-resident adaptive loop handed back to the host loop: ") + why[(r >= 1 && r <= 3) ? r : 0];
    }
    return MBAR_OK;
}


}  // namespace host
}  // namespace mbar

extern "C" {

int mbar_solve_adaptive(mbar_ctx* c, double* f_inout, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                        int check_convergence, double* history, int64_t history_rows, mbar_solve_result* result) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "mbar_solve_adaptive: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c || !f_inout) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (!c->have_Nk) return fail(c, MBAR_ERR_STATE, "mbar_ctx_set_Nk has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_ms();
    const int64_t K = c->K;
    std::vector<double> f(f_inout, f_inout + K), psum;
    mbar_solve_result res;
    std::memset(&res, 0, sizeof(res));
    // (the solver loops use the slot vectors for their own purposes -- and an evaluation inside them marks slot 0 valid again
    // before the loop overwrites it: cleared on EVERY way out, error returns included)
    struct Ld0Guard {
        mbar_ctx* c;
        ~Ld0Guard() { c->ld0_valid = false; }
    } ld0_guard{c};
    c->ld0_valid = false;
    double max_delta = std::numeric_limits<double>::quiet_NaN();
    int rc = refresh_poison(c);
    if (rc) return rc;
    bool on_device = device_loop_eligible(c) && !c->u_poison && f_is_finite(c, f.data(), 1) && maxiter > 0;
    int handbacks = 0;
    while (on_device) {
        bool handed_back = false;
        rc = adaptive_device_loop(c, f, tol, maxiter, min_sc_iter, gamma, check_convergence, history, history_rows, res, psum,
                                  max_delta, handed_back);
        if (rc) return rc;
        if (!handed_back) break;
        // The device handed the solve back.  A step too large for the sweeps' anchor point is a one-off (typically the
        // first Newton step from a poor start): ONE host-driven iteration, then back to the device, which re-anchors at
        // the new f.  Anything else (Newton system not positive definite, non-finite candidate) stays on the host.
        // (a non-finite candidate is usually the same situation seen from the other side -- a state whose weights
        // underflow at the current f -- and the host iteration handles it in log space; a Newton system that is not
        // positive definite, or repeated hand-backs, stay on the host)
        const int reason = c->h_ctl[CTL_REASON];
        const bool one_off = (reason == 2 || reason == 3) && ++handbacks <= 6 && res.iterations + 1 < maxiter;
        if (!one_off) {
            on_device = false;
            break;
        }
        rc = adaptive_host_loop(c, f, tol, res.iterations + 1, min_sc_iter, gamma, check_convergence, history, history_rows, res,
                                psum, max_delta);
        if (rc) return rc;
        if (res.success || !f_is_finite(c, f.data(), 1)) break;
    }
    if (!on_device && !res.success && res.iterations < maxiter) {
        rc = adaptive_host_loop(c, f, tol, maxiter, min_sc_iter, gamma, check_convergence, history, history_rows, res, psum,
                                max_delta);
        if (rc) return rc;
    }
    const int m = (int)c->sampled.size();
    double gn = 0.0;
    for (int i = 0; i < m && (int64_t)psum.size() == K; ++i) {
        const int k = c->sampled[i];
        gn += (psum[k] - c->Nk[k]) * (psum[k] - c->Nk[k]);
    }
    res.gnorm = std::sqrt(gn);
    c->ld0_valid = false;
    c->last_psum = (int64_t)psum.size() == K ? psum : std::vector<double>();
    res.max_delta = max_delta;
    res.wall_ms = now_ms() - t0;
    if (std::getenv("MBAR_DEBUG_TIMING") && res.iterations > 0)
        std::fprintf(stderr, "[mbar] adaptive: %lld iterations, %.3f ms per iteration (%s loop)\n", (long long)res.iterations,
                     res.wall_ms / res.iterations, on_device ? "device-resident" : "host-driven");
    std::copy(f.begin(), f.end(), f_inout);
    if (result) *result = res;
    return MBAR_OK;
}

int mbar_ctx_last_solve_psum(mbar_ctx* c, double* psum_out) {
    if (!c || !psum_out) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if ((int64_t)c->last_psum.size() != c->K) return fail(c, MBAR_ERR_STATE, "no adaptive solve has left its per-state sums on this context");
    std::copy(c->last_psum.begin(), c->last_psum.end(), psum_out);
    return MBAR_OK;
}

int mbar_solve_sci(mbar_ctx* c, double* f_inout, double tol, int64_t maxiter, int check_convergence,
                   mbar_solve_result* result) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "mbar_solve_sci: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c || !f_inout) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (!c->have_Nk) return fail(c, MBAR_ERR_STATE, "mbar_ctx_set_Nk has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const double t0 = now_ms();
    const int64_t K = c->K, Kp = c->Kp;
    const int first = c->sampled[0];
    mbar_solve_result res;
    std::memset(&res, 0, sizeof(res));
    struct Ld0Guard {
        mbar_ctx* c;
        ~Ld0Guard() { c->ld0_valid = false; }
    } ld0_guard{c};
    c->ld0_valid = false;
    {
        int prc = refresh_poison(c);
        if (prc) return prc;
        if (c->u_poison || !f_is_finite(c, f_inout, 1)) {  // NaN in, NaN out; isnan(max_delta) counts as converged (:636)
            for (int64_t k = 0; k < K; ++k)
                if (c->Nk[k] > 0.0) f_inout[k] = std::numeric_limits<double>::quiet_NaN();
            res.success = 1;
            res.max_delta = std::numeric_limits<double>::quiet_NaN();
            if (result) *result = res;
            return MBAR_OK;
        }
    }
    const int64_t rows = lse_rows(c);
    const int64_t batch = c->opt_sci_batch;
    if (!c->f_hist) HIPCHK(c, cache_malloc((void**)&c->f_hist, (size_t)256 * Kp * sizeof(double)));
    int rc = ensure_red(c, (size_t)rows + 8);
    if (rc) return rc;
    // initial f and aden on the device
    std::vector<double> hf((size_t)Kp, 0.0), ha((size_t)std::max(rows, Kp));
    for (int64_t k = 0; k < K; ++k) hf[k] = f_inout[k];
    build_aden(c, f_inout, ha.data(), std::max(rows, Kp));
    HIPCHK(c, hipMemcpyAsync(d_f(c), hf.data(), Kp * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_aden(c), ha.data(), std::max(rows, Kp) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> hdelta(256);
    bool done = false;
    double last_delta = std::numeric_limits<double>::quiet_NaN();
    // geometry and buffers of the fused path are fixed for the whole solve (nothing may allocate inside a capture)
    const bool fast = use_fast(c);
    const int nbk = (int)(rows / 16);
    const int64_t ntiles = (c->N + TS - 1) / TS;
    LaunchGeom g = fast ? lse_geometry(nbk, 1, c->num_cu, ntiles, c->opt_grid, lse_variant_for(c)) : LaunchGeom();
    g.balanced = c->opt_small_balanced ? 1 : 0;
    // Few states on one rank ("sci_merged", default): update and sweep of an iteration in ONE launch (k_sci_small) -- the update of
    // iteration i rides in the prologue of the sweep at f_i, so an iteration is one kernel instead of sweep + single-workgroup
    // update (config 2: ~9 us of a 62 us iteration).  Records / state double-buffered by the parity of the iteration, which the
    // captured batch bakes in: batches must be even.
    const bool merged = fast && g.variant == 4 && c->opt_sci_merged && c->nranks <= 1 && !c->comm && !stream_transport(c) &&
                        rows == Kp && batch % 2 == 0;
    if (fast) {
        rc = ensure(c, &c->part, &c->part_doubles, std::max((size_t)g.nwaves * (rows + 1), (size_t)2 * g.blocks * rows + g.blocks));
        if (rc) return rc;
        rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 16) * (rows + 1));
        if (rc) return rc;
    }
    // iteration 0 of the merged loop: the plain sweep at the start point leaves its records and f in the parity-0 buffers
    auto prime_merged = [&]() -> int {
        HIPCHK(c, hipMemcpyAsync(c->scratch, hf.data(), rows * sizeof(double), hipMemcpyHostToDevice, c->stream));
        ScopedTimer t(c, MBAR_TIMER_LSE);
        HIPCHK(c, launch_lse(c->stream, nbk, 1, g, c->u, c->ld, c->N, d_aden(c), c->cw, nullptr, nullptr, nullptr, c->part,
                             c->part + (size_t)2 * g.blocks * rows));
        return MBAR_OK;
    };
    if (merged) {
        rc = prime_merged();
        if (rc) return rc;
    }
    int64_t it = 0;  // iterations accepted so far
    long long* sci_stamps = nullptr;
    if (merged && std::getenv("MBAR_DEBUG_STAMPS")) {
        if (!c->stamps) HIPCHK(c, hipMalloc((void**)&c->stamps, 65 * 16 * sizeof(long long)));
        HIPCHK(c, hipMemsetAsync(c->stamps, 0, 64 * 8 * sizeof(long long), c->stream));
        sci_stamps = c->stamps;
    }
    auto merged_args = [&](int64_t b) {
        SciLoopArgs q;
        q.Nk = d_Nk(c);
        q.lnNk = d_lnNk(c);
        q.K = (int)K;
        q.first = first;
        q.tol = tol;
        q.state = c->scratch;
        q.rec = c->part;
        q.nrec = g.blocks;
        q.f_hist = c->f_hist + (size_t)b * Kp;
        q.delta_out = d_delta(c) + b;
        q.parity = (int)((it + b + 1) & 1);
        q.live = 0;
        for (int64_t j = 0; j < rows / 2; ++j)
            if ((2 * j < K && c->Nk[2 * j] > 0.0) || (2 * j + 1 < K && c->Nk[2 * j + 1] > 0.0)) q.live |= 1u << j;
        q.balanced = g.balanced;
        q.pingpong = c->opt_sci_pingpong ? 1 : 0;
        q.stamps = sci_stamps;
        return q;
    };
    // one SCI iteration into history slot b: sweep -> level-1 reduction -> [all-reduce] -> update (which folds the
    // last reduction level in)
    auto enqueue_iteration = [&](int64_t b, bool timed) -> int {
        double* fh = c->f_hist + (size_t)b * Kp;
        if (merged) {
            const SciLoopArgs q = merged_args(b);
            if (timed) {
                ScopedTimer t(c, MBAR_TIMER_LSE);
                HIPCHK(c, launch_sci_small(c->stream, nbk, g, c->u, c->ld, c->N, c->cw, q));
            } else {
                HIPCHK(c, launch_sci_small(c->stream, nbk, g, c->u, c->ld, c->N, c->cw, q));
            }
            return MBAR_OK;
        }
        if (fast) {
            double* psum_part = c->part;
            double* obj_part = c->part + (size_t)g.nwaves * rows;
            if (timed) {
                ScopedTimer t(c, MBAR_TIMER_LSE);
                HIPCHK(c, launch_lse(c->stream, nbk, 1, g, c->u, c->ld, c->N, d_aden(c), c->cw, nullptr,
                                     nullptr, nullptr, psum_part, obj_part));
            } else {
                HIPCHK(c, launch_lse(c->stream, nbk, 1, g, c->u, c->ld, c->N, d_aden(c), c->cw, nullptr,
                                     nullptr, nullptr, psum_part, obj_part));
            }
            const double* upd_src = psum_part;
            int64_t upd_n = g.nwaves;
            // the update kernel sums the partial records itself, 256 / KW of them in parallel per state (KW = states
            // rounded up to a power of two): worth it up to ~32 sequential adds per thread, a level-1 reduction otherwise
            int64_t kw2 = 1;
            while (kw2 < std::min<int64_t>(rows, 256)) kw2 <<= 1;
            if ((int64_t)g.nwaves * kw2 > 8192) {
                HIPCHK(c, launch_reduce_level1(c->stream, psum_part, g.nwaves, rows, c->scratch, &upd_n));
                upd_src = c->scratch;
            }
            if (c->nranks > 1 || c->comm) {
                HIPCHK(c, launch_reduce(c->stream, upd_src, upd_n, rows, c->scratch + (size_t)upd_n * rows, c->red));
                int r2 = allreduce_dev(c, c->red, rows, 0);
                if (r2) return r2;
                upd_src = c->red;
                upd_n = 1;
            }
            HIPCHK(c, launch_sci_update(c->stream, upd_src, upd_n, rows, d_Nk(c), d_lnNk(c), K, std::max(rows, Kp), first,
                                        tol, d_f(c), d_aden(c), fh, d_delta(c) + b));
        } else {
            int r2 = run_lse(c, 1, rows, nullptr, nullptr, false);
            if (r2) return r2;
            r2 = allreduce_dev(c, c->red, rows, 0);
            if (r2) return r2;
            HIPCHK(c, launch_sci_update(c->stream, c->red, 1, rows, d_Nk(c), d_lnNk(c), K, std::max(rows, Kp), first, tol,
                                        d_f(c), d_aden(c), fh, d_delta(c) + b));
        }
        return MBAR_OK;
    };
    // Launch-bound regime (a K=32, N=1e6 sweep is ~60 us): capture a whole batch into a hipGraph and replay it.
    const bool use_graph = fast && c->opt_graph && c->nranks <= 1 && !c->comm && maxiter >= batch;  // (no per-kernel events inside a graph)
    if (use_graph) {
        const int64_t sig = ((int64_t)g.blocks << 32) ^ ((int64_t)g.variant << 24) ^ (merged ? (1 << 16) : 0) ^ (g.balanced ? (1 << 17) : 0) ^ (c->opt_sci_pingpong ? (1 << 18) : 0) ^ first;
        if (!c->sci_graph || c->sci_graph_batch != batch || c->sci_graph_sig != sig || c->sci_graph_tol != tol) {
            if (c->sci_graph) HIPCHK(c, hipGraphExecDestroy(c->sci_graph));
            c->sci_graph = nullptr;
            rc = enqueue_iteration(0, false);  // eager warm-up: sets kernel attributes outside the capture
            if (rc) return rc;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipMemcpyAsync(d_f(c), hf.data(), Kp * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_aden(c), ha.data(), std::max(rows, Kp) * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipGraph_t graph = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            int crc = MBAR_OK;
            for (int64_t b = 0; b < batch && crc == MBAR_OK; ++b) crc = enqueue_iteration(b, false);
            hipError_t ee = hipStreamEndCapture(c->stream, &graph);
            if (crc) return crc;
            if (ee != hipSuccess) return fail(c, MBAR_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
            ee = hipGraphInstantiate(&c->sci_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ee != hipSuccess) return fail(c, MBAR_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ee));
            c->sci_graph_batch = batch;
            c->sci_graph_sig = sig;
            c->sci_graph_tol = tol;
        }
    }
    // Batches between two looks at the host.  Nothing to look at without the convergence test: the batches go out back to back and
    // only the last one is read.  With it: the first batch has the standard size (a captured graph), every later one the number of
    // iterations the relative change -- it decays geometrically -- still needs to reach `tol`, from its last two values (rows
    // and changes of up to 256 iterations are kept, the first one below `tol` is the answer whatever was enqueued behind it).
    // A look costs ~70 us of idle device (config 2: 92 iterations in two looks instead of six).
    std::vector<double> hrows;
    int64_t next_nb = batch;
    while (it < maxiter && !done) {
        const int64_t nb = std::min(next_nb, maxiter - it);
        if (use_graph && nb == batch) {
            HIPCHK(c, hipGraphLaunch(c->sci_graph, c->stream));
        } else {
            for (int64_t b = 0; b < nb; ++b) {
                rc = enqueue_iteration(b, c->opt_timing != 0);
                if (rc) return rc;
            }
        }
        if (!check_convergence && it + nb < maxiter) {
            it += nb;
            continue;
        }
        hrows.resize((size_t)nb * Kp);
        HIPCHK(c, hipMemcpyAsync(hdelta.data(), d_delta(c), nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hrows.data(), c->f_hist, (size_t)nb * Kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        rc = sync_stream(c);
        if (rc) return rc;
        int64_t stop = nb;  // index within the batch of the accepted iterate
        if (check_convergence) {
            for (int64_t b = 0; b < nb; ++b)
                if (std::isnan(hdelta[b]) || hdelta[b] < tol) {
                    stop = b + 1;
                    done = true;
                    break;
                }
            int64_t want = batch;
            if (!done && nb >= 2) {
                const double d1 = hdelta[nb - 1], d0 = hdelta[nb - 2];
                if (d1 > tol && d0 > d1 && d1 > 0.0) {
                    const double left = std::log(d1 / tol) / std::log(d0 / d1);
                    if (left == left) want = (int64_t)std::min(254.0, std::ceil(left)) + 2;
                }
            }
            want = std::max<int64_t>(2, std::min<int64_t>(256, want + (want & 1)));  // (even: records and state alternate by parity)
            double ctl[3] = {(double)stop, done ? 1.0 : 0.0, (double)want};
            rc = agree_with_rank0(c, ctl, 3);
            if (rc) return rc;
            stop = (int64_t)(ctl[0] + 0.5);
            done = ctl[1] > 0.5;
            next_nb = (int64_t)(ctl[2] + 0.5);
            if (done) res.success = 1;
        }
        it += stop;
        last_delta = hdelta[stop - 1];
        std::copy(hrows.begin() + (size_t)(stop - 1) * Kp, hrows.begin() + (size_t)stop * Kp, hf.begin());
    }
    if (sci_stamps) {  // (the last launch's stamps: 10 ns units relative to the workgroup's first stamp)
        long long st[24];
        HIPCHK(c, hipMemcpy(st, c->stamps, sizeof(st), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[mbar] k_sci_small workgroup 0, end of the tile loop per wave (us):");
        for (int w = 0; w < 8; ++w) std::fprintf(stderr, " %.2f", (st[16 + w] - st[0]) * 0.01);
        std::fprintf(stderr, "\n");
        for (int w = 0; w < 2; ++w) {
            const long long* p = st + 8 * w;
            std::fprintf(stderr, "[mbar] k_sci_small workgroup %s (us since its start; start offset to workgroup 0: %.2f): tables %.2f, update done %.2f, first tile in %.2f, "
                         "sweep done %.2f, barrier %.2f, record written %.2f\n", w ? "mid" : "0", (p[0] - st[0]) * 0.01, (p[1] - p[0]) * 0.01, (p[2] - p[0]) * 0.01,
                         (p[3] - p[0]) * 0.01, (p[4] - p[0]) * 0.01, (p[5] - p[0]) * 0.01, (p[6] - p[0]) * 0.01);
        }
    }
    for (int64_t k = 0; k < K; ++k)
        if (c->Nk[k] > 0.0) f_inout[k] = hf[k];
    res.iterations = it;
    res.sci_iter = it;
    res.max_delta = last_delta;
    res.wall_ms = now_ms() - t0;
    if (result) *result = res;
    return MBAR_OK;
}

}  // extern "C"
