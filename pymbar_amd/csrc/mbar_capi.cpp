// Host side of libmbar_hip.so, first of four translation units (mbar_ctx.h lists them): contexts, uploads and row operations,
// options, and the evaluation entry points of the C ABI declared in include/mbar_hip.h (sweeps + reductions: run_lse, run_gram,
// eval_core).  Owns the device-resident shard of u_kn and drives the gfx950 kernels of mbar_k_*.hip.  No PyTorch, no BLAS/LAPACK.
#include "mbar_ctx.h"

using namespace mbar;
using namespace mbar::host;

namespace mbar {
namespace host {

int fail(mbar_ctx* c, int code, const std::string& msg) {
    if (c) c->error = msg;
    g_last_error = msg;
    return code;
}
// ---- timing --------------------------------------------------------------------------------
hipEvent_t get_event(mbar_ctx* c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void flush_timers(mbar_ctx* c) {
    for (auto& tp : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tp.a, tp.b) == hipSuccess) {
            c->t_ms[tp.which] += ms;
            c->t_n[tp.which] += 1;
        }
        c->pool.push_back(tp.a);
        c->pool.push_back(tp.b);
    }
    c->pending.clear();
}
int sync_stream(mbar_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    flush_timers(c);
    return MBAR_OK;
}

// Scan the matrix once after it changed; a NaN or -inf entry poisons every reduced output (reference
// behaviour: logsumexp over all samples propagates it into every f_k).

// ---- device buffer helpers -------------------------------------------------------------------
int drop_graphs(mbar_ctx* c) {  // captured batches hold buffer pointers, sizes and the sampled-state set
    if (c->sci_graph || c->ad_graph) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->sci_graph) {
        HIPCHK(c, hipGraphExecDestroy(c->sci_graph));
        c->sci_graph = nullptr;
    }
    if (c->ad_graph) {
        HIPCHK(c, hipGraphExecDestroy(c->ad_graph));
        c->ad_graph = nullptr;
    }
    return MBAR_OK;
}
int ensure(mbar_ctx* c, double** p, size_t* have, size_t want) {
    if (*have >= want) return MBAR_OK;
    {
        int rc = drop_graphs(c);
        if (rc) return rc;
    }
    if (*p) HIPCHK(c, cache_free(*p));
    *p = nullptr;
    *have = 0;
    HIPCHK(c, cache_malloc((void**)p, want * sizeof(double)));
    *have = want;
    return MBAR_OK;
}
int refresh_poison(mbar_ctx* c) {
    if (c->u_checked) return MBAR_OK;
    c->ld0_valid = false;  // (every change of the matrix or of the transport clears u_checked)
    int* dflags = reinterpret_cast<int*>(d_delta(c) + 255);
    HIPCHK(c, hipMemsetAsync(dflags, 0, sizeof(int), c->stream));
    HIPCHK(c, launch_check_u(c->stream, c->u, c->ld, c->N, c->K, dflags));
    int h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, dflags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->nranks > 1) {
        // a NaN in ONE shard poisons the sums of every rank: agree on the flag, so that all ranks take the same early
        // return (a clean rank would otherwise wait in the all-reduce of a sweep the poisoned rank never launches)
        double v[2] = {(h & 3) ? 1.0 : 0.0, (h & 4) ? 1.0 : 0.0};
        int rc = allreduce_host(c, v, 2, 1);
        if (rc) return rc;
        h = (v[0] > 0.0 ? 1 : 0) | (v[1] > 0.0 ? 4 : 0);
    }
    c->u_poison = (h & 3) != 0;
    c->u_posinf = (h & 4) != 0;
    c->u_checked = true;
    return MBAR_OK;
}
bool f_is_finite(const mbar_ctx* c, const double* f, int nf) {
    for (int i = 0; i < nf; ++i)
        for (int64_t k = 0; k < c->K; ++k)
            if (c->Nk[k] > 0.0 && !std::isfinite(f[(size_t)i * c->K + k])) return false;
    return true;
}

// ---- evaluation building blocks -----------------------------------------------------------------
// Which specialised evaluation kernels this context qualifies for (flags for lse_geometry).  Row pitches from 7.6e7 samples up
// need 64-bit lane offsets in the tile DMA; only the general kernels are instantiated for that.
bool wide_pitch(const mbar_ctx* c) { return (uint64_t)c->ld * 56u + 128u >= (1ull << 32); }
int lse_variant_for(const mbar_ctx* c) {
    if (wide_pitch(c)) return 0;
    int v = 0;
    // bit 4: the context qualifies for the few-state kernel (one sample per lane, 64-sample tiles); lse_geometry
    // takes it for single-candidate sweeps of up to 32 states
    if (c->opt_small && c->Kp <= 32 && c->ld % 64 == 0) v |= 0x10;
    // bit 5: wide panels (129..256 states) may use the single-buffer kernel (four waves per CU instead of two)
    if (c->opt_wide && c->Kp > 128) v |= 0x20;
    return v;
}
bool use_fast(const mbar_ctx* c) { return c->K <= MAX_FAST_K && !c->opt_force_generic; }

// Host vector a[k] = f[k] + ln N_k (-inf where N_k = 0 or k >= K), written into out[rows].
void build_aden(const mbar_ctx* c, const double* f, double* out, int64_t rows) {
    const double ninf = -std::numeric_limits<double>::infinity();
    for (int64_t k = 0; k < rows; ++k) out[k] = (k < c->K && c->Nk[k] > 0.0) ? f[k] + c->lnNk[k] : ninf;
}

// 257 .. 1024 states: the one-read evaluation kernel whose eight waves split the rows of a tile
bool split_sweep_ok(const mbar_ctx* c, int64_t rows) {
    return !use_fast(c) && !c->opt_force_generic && !wide_pitch(c) && rows <= 1024 && rows % 64 == 0 && c->opt_wide;
}

// Evaluation pass for nf vectors whose aden already sits in d_aden (device, row pitch `rows`).
// Results: red[0 .. nf*rows) = psum, red[nf*rows .. nf*rows+nf) = sum logden.  Not all-reduced.
int run_lse(mbar_ctx* c, int nf, int64_t rows, double* ld0, double* ld1, bool use_offset) {
    const double* dn = use_offset ? c->dn : nullptr;
    if (use_fast(c)) {
        const int nb = (int)(rows / 16);
        const int64_t ntiles = (c->N + TS - 1) / TS;
        LaunchGeom g = lse_geometry(nb, nf, c->num_cu, ntiles, c->opt_grid, lse_variant_for(c));
        g.balanced = c->opt_small_balanced ? 1 : 0;
        const size_t rec = (size_t)nf * rows;
        int rc = ensure(c, &c->part, &c->part_doubles, (size_t)g.nwaves * (rec + nf));
        if (rc) return rc;
        rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * (rec + nf));
        if (rc) return rc;
        double* psum_part = c->part;
        double* obj_part = c->part + (size_t)g.nwaves * rec;
        {
            ScopedTimer t(c, MBAR_TIMER_LSE);
            HIPCHK(c, launch_lse(c->stream, nb, nf, g, c->u, c->ld, c->N, d_aden(c), c->cw, ld0,
                                 ld1, dn, psum_part, obj_part));
        }
        {
            // (per-state sums and objective terms in ONE pair of launches: at small sizes an evaluation is its launches -- the
            // scipy-driven protocol stages call this thirty times per solve)
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce2(c->stream, psum_part, (int64_t)rec, obj_part, nf, g.nwaves, c->scratch, c->red, c->red + rec));
        }
        return MBAR_OK;
    }
    // 257 .. 1024 states: the rows of a tile split over the eight waves of a workgroup -- ONE read of the matrix for one or
    // two candidates (the second through its ratio row, like the narrower kernels; the layout-agnostic kernels below read the
    // matrix twice per candidate: log-sum-exp pass + column-sum pass)
    if (split_sweep_ok(c, rows)) {
        int blocks = 0;
        int rc = ensure(c, &c->part, &c->part_doubles, (size_t)c->num_cu * nf * (rows + 1));
        if (rc) return rc;
        rc = ensure(c, &c->scratch, &c->scratch_doubles, (size_t)(c->num_cu / 32 + 2) * nf * (rows + 1));
        if (rc) return rc;
        double* obj_part = c->part + (size_t)c->num_cu * nf * rows;
        {
            ScopedTimer t(c, MBAR_TIMER_LSE);
            HIPCHK(c, launch_lse_split(c->stream, c->num_cu, nf, c->u, c->ld, c->N, rows, d_aden(c), c->cw, ld0, nf == 2 ? ld1 : nullptr, dn,
                                       c->part, obj_part, &blocks));
        }
        ScopedTimer t(c, MBAR_TIMER_REDUCE);
        HIPCHK(c, launch_reduce(c->stream, c->part, blocks, (int64_t)nf * rows, c->scratch, c->red));
        HIPCHK(c, launch_reduce(c->stream, obj_part, blocks, nf, c->scratch, c->red + (size_t)nf * rows));
        return MBAR_OK;
    }
    // generic: one f at a time
    for (int i = 0; i < nf; ++i) {
        int blocks = 0, cblocks = 0;
        int rc = ensure(c, &c->part, &c->part_doubles, (size_t)c->num_cu * 8 + (size_t)512 * c->K);
        if (rc) return rc;
        rc = ensure(c, &c->scratch, &c->scratch_doubles, (size_t)64 * (c->K + 8));
        if (rc) return rc;
        double* ldst = i == 0 ? ld0 : ld1;
        if (!ldst) ldst = c->logden[i];  // the column-sum kernel needs logden even if the caller does not
        {
            ScopedTimer t(c, MBAR_TIMER_LSE);
            HIPCHK(c, launch_lse_generic(c->stream, c->num_cu, c->u, c->ld, c->N, c->K, d_aden(c) + i * rows, c->cw,
                                         ldst, dn, c->part, &blocks));
        }
        {
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce(c->stream, c->part, blocks, 1, c->scratch, c->red + nf * rows + i));
        }
        {
            ScopedTimer t(c, MBAR_TIMER_LSE);
            HIPCHK(c, launch_colsum_generic(c->stream, c->num_cu, c->u, c->ld, c->N, c->K, d_aden(c) + i * rows, c->cw,
                                            ldst, c->part, &cblocks));
        }
        {
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, hipMemsetAsync(c->red + i * rows, 0, rows * sizeof(double), c->stream));
            HIPCHK(c, launch_reduce(c->stream, c->part, cblocks, c->K, c->scratch, c->red + i * rows));
        }
    }
    return MBAR_OK;
}

// Gram-pass geometry: list of launches and where their 16 x 16 blocks land.  Up to 128 states: one diagonal panel.
// Beyond: 128-state panels (+ one trailing 64-state panel); a diagonal panel is one launch (upper-triangular blocks),
// a pair of panels is covered by 64 x 128 rectangles (nbi = 4 block rows of the I panel x nbj = 8 block columns of the
// J panel = 32 blocks, the most one wave's register file holds next to the operands).
GramPlan gram_plan(int64_t Kp, bool quad) {
    GramPlan p;
    if (Kp <= 128 || quad) {  // (quad: 129 .. 256 states as ONE panel, its blocks split over the four waves of a workgroup)
        int nb = (int)(Kp / 16);
        p.items.push_back({true, 0, 0, nb, nb, nb * (nb + 1) / 2, 0});
        p.total_blocks = (size_t)nb * (nb + 1) / 2;
        return p;
    }
    struct Panel { int64_t r0; int nb; };
    std::vector<Panel> panels;
    for (int64_t r = 0; r < Kp;) {
        const int nb = Kp - r >= 128 ? 8 : (int)((Kp - r) / 16);  // Kp is a multiple of PANEL = 64 here
        panels.push_back({r, nb});
        r += 16 * nb;
    }
    size_t off = 0;
    for (const auto& a : panels) {
        const int nblk = a.nb * (a.nb + 1) / 2;
        p.items.push_back({true, a.r0, a.r0, a.nb, a.nb, nblk, off});
        off += nblk;
    }
    for (size_t ia = 0; ia < panels.size(); ++ia)
        for (size_t ib = ia + 1; ib < panels.size(); ++ib) {
            const Panel &a = panels[ia], &b = panels[ib];
            if (b.nb == 8) {  // 128 x 128: two 64 x 128 rectangles
                for (int h = 0; h < 2; ++h) {
                    p.items.push_back({false, a.r0 + 64 * h, b.r0, 4, 8, 32, off});
                    off += 32;
                }
            } else if (b.nb == 4) {  // the trailing 64-state panel against a 128-state one: I = the short panel
                p.items.push_back({false, b.r0, a.r0, 4, 8, 32, off});
                off += 32;
            } else {  // (cannot happen for Kp a multiple of 64; kept correct anyway: 64 x 64 squares)
                for (int64_t ri = a.r0; ri < a.r0 + 16 * a.nb; ri += 64)
                    for (int64_t rj = b.r0; rj < b.r0 + 16 * b.nb; rj += 64) {
                        p.items.push_back({false, ri, rj, 4, 4, 16, off});
                        off += 16;
                    }
            }
        }
    p.total_blocks = off;
    return p;
}

// The plan of the host-driven loop's P mode (more than 256 states): 256-state panels, each ONE read of its rows (k_gram_quad on
// the probability matrix), the remainder (64 / 128 / 192 states) as one more diagonal panel, and between the panels the
// rectangles of k_gram_rect: 128 x 256 (64 x 256 for the short remainder), four waves on one shared tile stream.  At 1024 states
// 4 + 12 launches that read 5632 rows of the matrix (gram_plan above: 64 launches, 11776 rows).
GramPlan gram_plan_pmode(int64_t Kp) {
    GramPlan p;
    struct Panel { int64_t r0; int nb; };
    std::vector<Panel> panels;
    for (int64_t r = 0; r < Kp;) {
        const int nb = Kp - r >= 256 ? 16 : (int)((Kp - r) / 16);  // Kp is a multiple of 64 here
        panels.push_back({r, nb});
        r += 16 * nb;
    }
    size_t off = 0;
    for (const auto& a : panels) {
        const int nblk = a.nb * (a.nb + 1) / 2;
        p.items.push_back({true, a.r0, a.r0, a.nb, a.nb, nblk, off});
        off += nblk;
    }
    for (size_t ia = 0; ia < panels.size(); ++ia)
        for (size_t ib = ia + 1; ib < panels.size(); ++ib) {
            const Panel &a = panels[ia], &b = panels[ib];  // a is a 256-state panel; b one too, or the remainder
            const Panel &rows = b.nb == 16 ? a : b, &cols = b.nb == 16 ? b : a;
            for (int done = 0; done < rows.nb;) {
                const int nbi = rows.nb - done >= 8 ? 8 : 4;
                p.items.push_back({false, rows.r0 + 16 * done, cols.r0, nbi, 16, nbi * 16, off});
                off += (size_t)nbi * 16;
                done += nbi;
            }
        }
    p.total_blocks = off;
    return p;
}

// 129 .. 256 states: the one-read kernel (k_gram_quad) needs LDS-DMA staging
bool use_quad(const mbar_ctx* c) { return c->opt_quad && use_fast(c) && c->Kp > 128 && c->Kp <= 256; }
GramPlan plan_for(const mbar_ctx* c) { return gram_plan(c->Kp, use_quad(c)); }
// blocks of 16 states of the 192- / 256-row panel that hold real states: up to 160 / 224 states the one-read kernels leave the
// last two blocks (padding rows only) out of the staging, the operand step and the matrix instructions
int quad_live_blocks(const mbar_ctx* c) { return c->opt_quad_trim ? (int)((c->K + 15) / 16) : 0; }

// Gram pass with operand exp(anum_k - u_kn - logden_n); anum (device) has Kp entries.
// Results: gram blocks at red + red_off (plan order).  The per-state operand sums are not accumulated on the
// device: rows of p sum to one (sum_k p_nk = 1, resp. sum_k N_k W_nk = 1), so they are column sums of the
// reduced Gram matrix (gram_operand_sums below).
// pmat (or nullptr: the reduced potentials) = a resident probability matrix, `logden` then the reciprocals 1 / s_n with the
// multiplicities' roots already folded in (P mode of the host-driven loop above 256 states: panels and rectangles only)
int run_gram(mbar_ctx* c, const double* anum_dev, const double* logden, size_t red_off, const GramPlan& plan, const double* pmat) {
    const int64_t ntiles = (c->N + TS - 1) / TS;
    const double* mat = pmat ? pmat : c->u;
    if (c->weighted && !pmat) {  // sum_n c_n p p^T: each operand carries sqrt(c_n), folded into the exponent
        HIPCHK(c, launch_shift_logden(c->stream, logden, c->cw, 0.5, c->N, c->lden_eff));
        logden = c->lden_eff;
    }
    for (const auto& it : plan.items) {
        if (!it.diag && it.nbj == 16) {  // P mode: rectangle against a 256-state panel, four waves on one shared tile stream
            if (!pmat) return fail(c, MBAR_ERR_STATE, "run_gram: the 256-column rectangles exist on the probability matrix only");
            LaunchGeom g = gram_quad_geometry(it.nbi + 16, c->num_cu, ntiles, c->opt_grid);
            if (it.nbi == 8 && c->opt_rect_waves == 8) {  // two waves per SIMD, 16 blocks each (+ their copies of the tile's reciprocals)
                g.waves = 8;
                g.lds_bytes += 2 * 4 * 1024;
            }
            const size_t rec = (size_t)it.nblk * 256;
            int rc = ensure(c, &c->part, &c->part_doubles, (size_t)g.nwaves * rec);
            if (rc) return rc;
            rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
            if (rc) return rc;
            {
                ScopedTimer t(c, MBAR_TIMER_GRAM);
                HIPCHK(c, launch_gram_rect(c->stream, it.nbi, g, pmat, c->ld, c->N, it.ri, it.rj, logden, c->part));
            }
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce(c->stream, c->part, g.nwaves, (int64_t)rec, c->scratch, c->red + red_off + it.off * 256));
            continue;
        }
        if (it.diag && it.nbi > 8) {  // one read of the matrix: the four waves of a workgroup split the panel's blocks
            LaunchGeom g = gram_quad_geometry(it.nbi, c->num_cu, ntiles, c->opt_grid);
            g.live_blocks = pmat ? 0 : quad_live_blocks(c);
            const size_t rec = (size_t)it.nblk * 256;
            int rc = ensure(c, &c->part, &c->part_doubles, (size_t)g.nwaves * rec);
            if (rc) return rc;
            rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
            if (rc) return rc;
            {
                ScopedTimer t(c, MBAR_TIMER_GRAM);
                LoopCtl lo;
                lo.pmode = pmat != nullptr;  // (then: the panel's rows of the probability matrix, `logden` = reciprocals)
                HIPCHK(c, launch_gram_quad(c->stream, it.nbi, g, mat + it.ri * c->ld, c->ld, c->N, anum_dev + it.ri, logden, c->part, lo));
            }
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce(c->stream, c->part, g.nwaves, (int64_t)rec, c->scratch, c->red + red_off + it.off * 256));
            continue;
        }
        const int tile_rows = it.diag ? it.nbi * 16 : (it.nbi + it.nbj) * 16;
        LaunchGeom g = gram_geometry(tile_rows, it.diag, c->num_cu, ntiles, c->opt_grid);
        const size_t rec = (size_t)it.nblk * 256;
        int rc = ensure(c, &c->part, &c->part_doubles, (size_t)g.nwaves * rec);
        if (rc) return rc;
        rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
        if (rc) return rc;
        double* gp = c->part;
        {
            ScopedTimer t(c, MBAR_TIMER_GRAM);
            if (it.diag)
            {
                LoopCtl lo;
                lo.unclamped = c->u_checked && !c->u_posinf;
                lo.pmode = pmat != nullptr;
                HIPCHK(c, launch_gram_diag(c->stream, it.nbi, g, mat, c->ld, c->N, anum_dev + it.ri, logden,
                                           it.ri, gp, nullptr, lo));
            }
            else
                HIPCHK(c, launch_gram_off(c->stream, it.nbj, g, mat, c->ld, c->N, anum_dev + it.ri, anum_dev + it.rj,
                                          logden, it.ri, it.rj, gp, pmat != nullptr));
        }
        {
            ScopedTimer t(c, MBAR_TIMER_REDUCE);
            HIPCHK(c, launch_reduce(c->stream, gp, g.nwaves, (int64_t)rec, c->scratch, c->red + red_off + it.off * 256));
        }
    }
    return MBAR_OK;
}

// out_j = sum_k w_k G[k][j]  (w = 1: operand sums of the p-mode Gram; w = N_k: sum_n W_nj of the W-mode Gram)
void gram_operand_sums(const double* G, int64_t K, const double* w, double* out) {
    for (int64_t j = 0; j < K; ++j) out[j] = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        const double wk = w ? w[k] : 1.0;
        if (wk == 0.0) continue;
        for (int64_t j = 0; j < K; ++j) out[j] += wk * G[k * K + j];
    }
}

void gram_row_sums(const double* blocks, int nb, int64_t K, double* psum) {
    std::vector<double> acc((size_t)nb * 16, 0.0);
    int b = 0;
    for (int I = 0; I < nb; ++I)
        for (int J = I; J < nb; ++J, ++b) {
            const double* blk = blocks + (size_t)b * 256;
            for (int r = 0; r < 16; ++r)
                for (int q = (I == J ? r : 0); q < 16; ++q) {
                    const double v = blk[r * 16 + q];
                    acc[(size_t)16 * I + r] += v;
                    if (I != J || q != r) acc[(size_t)16 * J + q] += v;
                }
        }
    for (int64_t k = 0; k < K; ++k) psum[k] = acc[(size_t)k];
}

// Scatter reduced blocks (host copy) into a dense symmetric K x K matrix.
void unpack_gram(const GramPlan& plan, const double* blocks, int64_t K, double* G) {
    for (const auto& it : plan.items) {
        int b = 0;
        for (int I = 0; I < it.nbi; ++I) {
            for (int J = it.diag ? I : 0; J < it.nbj; ++J) {
                const double* blk = blocks + (it.off + b) * 256;
                for (int r = 0; r < 16; ++r)
                    for (int q = 0; q < 16; ++q) {
                        const int64_t gi = it.ri + 16 * I + r, gj = it.rj + 16 * J + q;
                        if (gi < K && gj < K) {
                            const double v = blk[r * 16 + q];
                            if (it.diag && I == J) {
                                if (q >= r) { G[gi * K + gj] = v; G[gj * K + gi] = v; }
                            } else {
                                G[gi * K + gj] = v;
                                G[gj * K + gi] = v;
                            }
                        }
                    }
                ++b;
            }
        }
    }
}

// The host-driven loop's Newton matrix straight from the reduced blocks (one pass instead of blocks -> K x K matrix -> scaling ->
// m x m matrix): H[i][j] = -G[k_i][k_j] f[k_i] f[k_j] over the states with samples (pos[k] = position of state k among them, -1:
// none; f: the per-state factors the P-mode sweep leaves out, or nullptr), both triangles; the blocks are dealt over the host team.
void unpack_gram_to_hessian(const GramPlan& plan, const double* blocks, int64_t K, const double* factor, const int* pos, int m, double* H,
                            int threads) {
    struct Blk { const double* p; int64_t gi0, gj0; bool tri; };
    std::vector<Blk> list;
    list.reserve(plan.total_blocks);
    for (const auto& it : plan.items) {
        int b = 0;
        for (int I = 0; I < it.nbi; ++I)
            for (int J = it.diag ? I : 0; J < it.nbj; ++J) list.push_back({blocks + (it.off + b++) * 256, it.ri + 16 * I, it.rj + 16 * J, it.diag && I == J});
    }
    const int T = std::max(1, std::min<int>(threads, (int)list.size() / 64));
    host_team_run(T, [&](int t) {
        for (size_t n = t; n < list.size(); n += T) {
            const Blk& bk = list[n];
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = bk.gi0 + r;
                const int i = gi < K ? pos[gi] : -1;
                if (i < 0) continue;
                const double fi = factor ? -factor[gi] : -1.0;
                for (int q = bk.tri ? r : 0; q < 16; ++q) {
                    const int64_t gj = bk.gj0 + q;
                    const int j = gj < K ? pos[gj] : -1;
                    if (j < 0) continue;
                    const double v = bk.p[r * 16 + q] * (factor ? fi * factor[gj] : fi);
                    H[(size_t)i * m + j] = v;
                    H[(size_t)j * m + i] = v;
                }
            }
        }
    });
}

int ensure_red(mbar_ctx* c, size_t want) {
    if (c->red_doubles >= want) return MBAR_OK;
    {
        int rc = drop_graphs(c);
        if (rc) return rc;
    }
    if (c->red) HIPCHK(c, cache_free(c->red));
    if (c->hred) HIPCHK(c, cache_host_free(c->hred));
    c->red = nullptr;
    c->hred = nullptr;
    c->red_doubles = 0;
    HIPCHK(c, cache_malloc((void**)&c->red, want * sizeof(double)));
    HIPCHK(c, cache_host_malloc((void**)&c->hred, want * sizeof(double)));
    c->red_doubles = want;
    return MBAR_OK;
}

// Row pitch of the aden / psum vectors: the padded state count (padded_K() only produces values
// for which the fused kernel has an instantiation: 16..128 step 16, 192, 256).
int64_t lse_rows(const mbar_ctx* c) { return c->Kp; }

// Core of mbar_eval: f points to nf*K doubles on the host.  Leaves logden in ld0/ld1.
int eval_core(mbar_ctx* c, const double* f, int nf, unsigned flags, double* ld0, double* ld1, double* psum,
              double* sumlogden, double* gram) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "evaluation: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c->have_Nk) return fail(c, MBAR_ERR_STATE, "mbar_ctx_set_Nk has not been called");
    if (nf < 1 || nf > 2) return fail(c, MBAR_ERR_ARG, "nf must be 1 or 2");
    const bool want_gram = (flags & MBAR_EVAL_GRAM) != 0;
    const bool use_off = (flags & MBAR_EVAL_USE_OFFSET) != 0;
    if (use_off && !c->dn) return fail(c, MBAR_ERR_STATE, "objective offset requested but not set");
    {
        int prc = refresh_poison(c);
        if (prc) return prc;
        if (c->u_poison || !f_is_finite(c, f, nf)) {
            const double qnan = std::numeric_limits<double>::quiet_NaN();
            if (psum) std::fill(psum, psum + (size_t)nf * c->K, qnan);
            if (sumlogden) std::fill(sumlogden, sumlogden + nf, qnan);
            if (want_gram && gram) std::fill(gram, gram + (size_t)c->K * c->K, qnan);
            c->error = c->u_poison ? "u_kn contains NaN or -inf: all sums are NaN" : "f_k is not finite: all sums are NaN";
            return MBAR_OK;
        }
    }
    // only the log-denominators are asked for, and slot 0 still holds them for this very f: nothing to do
    const bool only_logden = nf == 1 && ld0 == c->logden[0] && !ld1 && !psum && !sumlogden && !want_gram && !use_off;
    if (only_logden && c->ld0_valid && (int64_t)c->ld0_f.size() == c->K && std::equal(f, f + c->K, c->ld0_f.begin())) return MBAR_OK;
    if (ld0 == c->logden[0] || ld1 == c->logden[0]) c->ld0_valid = false;
    const int64_t rows = lse_rows(c);
    GramPlan plan;
    if (want_gram) plan = plan_for(c);
    const size_t n_ps = (size_t)nf * rows, n_obj = nf;
    const size_t off_gram = n_ps + n_obj, n_gram = plan.total_blocks * 256;
    const size_t total = off_gram + n_gram;
    int rc = ensure_red(c, total);
    if (rc) return rc;
    // aden -> device.  The fused kernels evaluate a second candidate from the first one's exponentials through
    // the per-state ratio c_k = exp(a'_k - a_k) (row 1); if the two candidates are so far apart that the ratio
    // could over/underflow against the shared shift, fall back to two single-candidate sweeps.
    std::vector<double> h((size_t)nf * rows);
    for (int i = 0; i < nf; ++i) build_aden(c, f + (size_t)i * c->K, h.data() + (size_t)i * rows, rows);
    bool split = false;
    std::vector<double> ratio;  // c_k of the fused two-candidate sweep; applied to its second psum row below
    if (nf == 2 && (use_fast(c) || split_sweep_ok(c, rows))) {
        double dmax = 0.0;
        for (int64_t k = 0; k < rows; ++k) {
            const double a0 = h[k], a1 = h[rows + k];
            const double d = (std::isinf(a0) && std::isinf(a1)) ? 0.0 : a1 - a0;
            dmax = std::max(dmax, std::fabs(d));
            h[rows + k] = std::exp(d);
        }
        split = !(dmax < 300.0);
        if (!split) ratio.assign(h.begin() + rows, h.begin() + 2 * rows);
    }
    if (split) {
        std::vector<double> ps((size_t)2 * c->K), sl(2);
        int rc2 = eval_core(c, f, 1, flags, ld0, nullptr, ps.data(), &sl[0], gram);
        if (rc2) return rc2;
        rc2 = eval_core(c, f + c->K, 1, flags & ~MBAR_EVAL_GRAM, ld1, nullptr, ps.data() + c->K, &sl[1], nullptr);
        if (rc2) return rc2;
        if (psum) std::copy(ps.begin(), ps.end(), psum);
        if (sumlogden) std::copy(sl.begin(), sl.end(), sumlogden);
        return MBAR_OK;
    }
    // (pinned staging: the copy is stream-ordered before the sweep and the host only touches the buffer again after
    // the sweep's results have been read back, so no synchronisation is needed here)
    std::copy(h.begin(), h.end(), c->hstage);
    HIPCHK(c, hipMemcpyAsync(d_aden(c), c->hstage, h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // One rank, sums only: the last level of the reduction writes into the pinned host buffer itself (it is mapped into the
    // device's address space) -- one copy kernel less per evaluation; the scipy-driven protocol stages call this thirty times
    // per solve and at the sizes pymbar is mostly used at an evaluation IS its launches.
    const bool direct = c->opt_direct_results && c->nranks <= 1 && !c->comm && !stream_transport(c) && use_fast(c) && !want_gram;
    struct RedSwap {
        mbar_ctx* c; double* saved;
        RedSwap(mbar_ctx* c_, bool on) : c(c_), saved(nullptr) { if (on) { saved = c->red; c->red = c->hred; } }
        ~RedSwap() { if (saved) c->red = saved; }
    };
    {
        RedSwap swap(c, direct);
        rc = run_lse(c, nf, rows, ld0, ld1, use_off);
    }
    if (rc) return rc;
    if (want_gram) {
        // p-mode operand: anum = aden of f[0] (Kp entries)
        std::vector<double> an((size_t)c->Kp);
        build_aden(c, f, an.data(), c->Kp);
        std::copy(an.begin(), an.end(), c->hstage + 2 * c->Kp);
        HIPCHK(c, hipMemcpyAsync(d_anum(c), c->hstage + 2 * c->Kp, an.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        rc = run_gram(c, d_anum(c), ld0, off_gram, plan);
        if (rc) return rc;
    }
    if (!direct) {
        rc = allreduce_dev(c, c->red, (int64_t)total, 0);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->hred, c->red, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    rc = sync_stream(c);
    if (rc) return rc;
    if (!ratio.empty())  // the fused sweep accumulates sum_n e_nk / s'_n for the second candidate: times c_k = its psum
        for (int64_t k = 0; k < rows; ++k) c->hred[(size_t)rows + k] *= ratio[(size_t)k];
    if (psum)
        for (int i = 0; i < nf; ++i)
            for (int64_t k = 0; k < c->K; ++k) psum[(size_t)i * c->K + k] = c->hred[(size_t)i * rows + k];
    if (sumlogden)
        for (int i = 0; i < nf; ++i) sumlogden[i] = c->hred[n_ps + i];
    if (want_gram && gram) {
        std::fill(gram, gram + (size_t)c->K * c->K, 0.0);
        unpack_gram(plan, c->hred + off_gram, c->K, gram);
    }
    if (c->opt_check_finite) {
        for (size_t i = 0; i < n_ps + n_obj; ++i)
            if (!std::isfinite(c->hred[i])) {
                // not fatal for the caller's control flow (the reference also propagates NaN), but flagged
                c->error = "non-finite partial sum in evaluation pass";
                break;
            }
    }
    if (nf == 1 && ld0 == c->logden[0]) {
        c->ld0_f.assign(f, f + c->K);
        c->ld0_valid = true;
    }
    return MBAR_OK;
}


}  // namespace host
}  // namespace mbar


// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int mbar_version(void) { return 101; }

const char* mbar_last_error(const mbar_ctx* ctx) { return ctx ? ctx->error.c_str() : g_last_error.c_str(); }

int mbar_device_count(int* count) {
    if (!count) return fail(nullptr, MBAR_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, MBAR_ERR_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return MBAR_OK;
}

int mbar_device_info(int device, char* name, int name_len, int* compute_units, int64_t* total_mem_bytes) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return fail(nullptr, MBAR_ERR_NODEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", p.name[0] ? p.name : "AMD Instinct MI355X", p.gcnArchName);
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (total_mem_bytes) *total_mem_bytes = (int64_t)p.totalGlobalMem;
    return MBAR_OK;
}

int mbar_ctx_create(mbar_ctx** out, int device, int64_t K, int64_t N_local) {
    if (!out) return fail(nullptr, MBAR_ERR_ARG, "out is NULL");
    *out = nullptr;
    // (N_local = 0 is a legal shard: with more ranks than 16-sample tiles a rank owns no column, yet it must take part in every
    // collective of the loop; it keeps one all-padding tile so that every kernel has something to launch on)
    if (K < 1 || N_local < 0) return fail(nullptr, MBAR_ERR_ARG, "K must be >= 1 and N_local >= 0");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(nullptr, MBAR_ERR_NODEVICE, "no HIP device visible (libmbar_hip needs an MI355X / gfx950 GPU)");
    if (device < 0 || device >= n) return fail(nullptr, MBAR_ERR_ARG, "device index out of range");
    mbar_ctx* c = new mbar_ctx();
    g_live_contexts.fetch_add(1);
    c->device = device;
    c->K = K;
    c->Kp = padded_K(K);
    c->N = N_local;
    // row pitch: whole 16-sample tiles; whole 64-sample tiles where the few-state evaluation kernel may run (K <= 32)
    c->ld = c->Kp <= 32 ? (N_local + 63) / 64 * 64 : (N_local + TS - 1) / TS * TS;
    if (c->ld == 0) c->ld = c->Kp <= 32 ? 64 : TS;
#define CRT(expr)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            int rc_ = fail(nullptr, MBAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
            mbar_ctx_destroy(c);                                                                    \
            return rc_;                                                                             \
        }                                                                                           \
    } while (0)
    CRT(hipSetDevice(device));
    // (hipGetDeviceProperties and stream creation cost milliseconds: properties are looked up once per device, streams of
    // destroyed contexts are kept for the next one)
    DevInfo di;
    {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        auto it = g_dev_info.find(device);
        if (it == g_dev_info.end()) {
            hipDeviceProp_t p;
            CRT(hipGetDeviceProperties(&p, device));
            di.num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
            di.arch = p.gcnArchName;
            g_dev_info[device] = di;
        } else {
            di = it->second;
        }
        auto& pool = g_stream_pool[device];
        if (!pool.empty()) {
            c->stream = pool.back();
            pool.pop_back();
        }
    }
    c->num_cu = di.num_cu;
    if (std::strncmp(di.arch.c_str(), "gfx950", 6) != 0) {
        int rc = fail(nullptr, MBAR_ERR_NODEVICE, std::string("device is ") + di.arch + ", this library is built for gfx950 only");
        mbar_ctx_destroy(c);
        return rc;
    }
    if (!c->stream) CRT(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t ubytes = (size_t)c->Kp * c->ld * sizeof(double);
    CRT(cache_malloc((void**)&c->u, ubytes));
    CRT(launch_zero(c->stream, c->u, ubytes));
    // three logden vectors in ONE allocation: the device-resident loop addresses them as base + slot * ld
    CRT(cache_malloc((void**)&c->logden[0], (size_t)3 * c->ld * sizeof(double)));
    CRT(launch_zero(c->stream, c->logden[0], (size_t)3 * c->ld * sizeof(double)));
    c->logden[1] = c->logden[0] + c->ld;
    c->logden[2] = c->logden[0] + 2 * c->ld;
    CRT(cache_malloc((void**)&c->cw, (size_t)c->ld * sizeof(double)));
    CRT(launch_zero(c->stream, c->cw, (size_t)c->ld * sizeof(double)));
    if (c->N > 0) CRT(launch_fill(c->stream, c->cw, 1.0, c->N));  // (unit multiplicities, 0 on the padding: filled on the device)
    CRT(cache_malloc((void**)&c->small, small_doubles(c->Kp) * sizeof(double)));
    CRT(cache_host_malloc((void**)&c->hstage, (size_t)4 * c->Kp * sizeof(double)));
    CRT(hipMemsetAsync(c->small, 0, small_doubles(c->Kp) * sizeof(double), c->stream));
    CRT(hipStreamSynchronize(c->stream));
#undef CRT
    c->Nk.assign(K, 0.0);
    c->lnNk.assign(K, 0.0);
    *out = c;
    return MBAR_OK;
}

void mbar_ctx_destroy(mbar_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    flush_timers(c);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->u_alloc) (void)cache_free(c->u_alloc);
    else if (c->u) (void)cache_free(c->u);
    if (c->logden[0]) (void)cache_free(c->logden[0]);
    if (c->ad) (void)cache_free(c->ad);
    if (c->P) (void)cache_free(c->P);
    if (c->pm_vec) (void)cache_free(c->pm_vec);
    if (c->pm_ld0) (void)cache_free(c->pm_ld0);
    if (c->part_g) (void)cache_free(c->part_g);
    if (c->cwsq) (void)cache_free(c->cwsq);
    if (c->chol) (void)cache_free(c->chol);
    if (c->ad_ints) (void)cache_free(c->ad_ints);
    if (c->h_ctl) (void)cache_host_free(c->h_ctl);
    if (c->ad_graph) (void)hipGraphExecDestroy(c->ad_graph);
    if (c->dn) (void)cache_free(c->dn);
    if (c->cw) (void)cache_free(c->cw);
    if (c->lden_eff) (void)cache_free(c->lden_eff);
    if (c->small) (void)cache_free(c->small);
    if (c->part) (void)cache_free(c->part);
    if (c->scratch) (void)cache_free(c->scratch);
    if (c->red) (void)cache_free(c->red);
    if (c->hred) (void)cache_host_free(c->hred);
    if (c->lognum_part) (void)cache_free(c->lognum_part);
    if (c->f_hist) (void)cache_free(c->f_hist);
    if (c->hstage) (void)cache_host_free(c->hstage);
    if (c->vec_tmp) (void)cache_free(c->vec_tmp);
    if (c->boot_idx) (void)cache_free(c->boot_idx);
    if (c->stamps) (void)hipFree(c->stamps);
    if (c->sci_graph) (void)hipGraphExecDestroy(c->sci_graph);
    if (c->stream) {  // (idle: synchronised above) kept for the next context on this device
        std::lock_guard<std::mutex> lock(g_dev_mu);
        auto& pool = g_stream_pool[c->device];
        if (pool.size() < 8)
            pool.push_back(c->stream);
        else
            (void)hipStreamDestroy(c->stream);
    }
    delete c;
    if (g_live_contexts.fetch_sub(1) == 1) g_mem.trim_to(g_mem.idle_limit());
}

int mbar_ctx_synchronize(mbar_ctx* c) {
    if (!c) return fail(nullptr, MBAR_ERR_ARG, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return sync_stream(c);
}

int mbar_device_synchronize(int device) {
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(nullptr, MBAR_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
    return MBAR_OK;
}

int mbar_ctx_set_option(mbar_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return fail(c, MBAR_ERR_ARG, "NULL argument");
    const std::string k(key);
    if (k == "grid_blocks") c->opt_grid = value;
    else if (k == "force_generic") c->opt_force_generic = value;
    else if (k == "small_k_kernel") c->opt_small = value;
    else if (k == "wide_k_kernel") c->opt_wide = value;
    else if (k == "check_finite") c->opt_check_finite = value;
    else if (k == "timing") c->opt_timing = value;
    else if (k == "graph") c->opt_graph = value;
    else if (k == "sci_batch") c->opt_sci_batch = value < 1 ? 1 : (value > 256 ? 256 : value);
    else if (k == "device_loop") c->opt_device_loop = value;
    else if (k == "pmode") c->opt_pmode = value;
    else if (k == "fused") c->opt_fused = value;
    else if (k == "gram_quad") c->opt_quad = value;
    else if (k == "device_loop_wide") c->opt_device_loop_wide = value;
    else if (k == "pcache") c->opt_pcache = value;
    else if (k == "merge_select") c->opt_merge_select = value;
    else if (k == "light_last") c->opt_light_last = value;
    else if (k == "debug_download_p") c->opt_debug_download_p = value;
    else if (k == "direct_results") c->opt_direct_results = value;
    else if (k == "sci_merged") c->opt_sci_merged = value;
    else if (k == "host_pmode") c->opt_host_pmode = value;
    else if (k == "rect_waves") c->opt_rect_waves = value == 8 ? 8 : 4;
    else if (k == "newton_ldlt") {
        c->opt_newton_ldlt = value ? 1 : 0;
        (void)drop_graphs(c);
    }
    else if (k == "sci_pingpong") {
        c->opt_sci_pingpong = value;
        (void)drop_graphs(c);
    }
    else if (k == "small_balanced") {
        c->opt_small_balanced = value;
        (void)drop_graphs(c);
    }
    else if (k == "wide_pmode") c->opt_wide_pmode = value;
    else if (k == "quad_trim") {
        c->opt_quad_trim = value;
        c->P_valid = false;  // (the trimmed build never writes the padding rows of P, the untrimmed sweeps read them)
    }
    else if (k == "adapt_batch") c->opt_adapt_batch = value < 1 ? 1 : (value > 64 ? 64 : value);
    else return fail(c, MBAR_ERR_ARG, "unknown option: " + k);
    return MBAR_OK;
}

int mbar_ctx_upload_u(mbar_ctx* c, const double* u_host, int64_t ld_host, int64_t col0_host, int64_t ncols,
                      int64_t col0_dev) {
    if (!c || !u_host) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (ncols < 0 || col0_dev < 0 || col0_dev + ncols > c->N || col0_host < 0 || col0_host + ncols > ld_host)
        return fail(c, MBAR_ERR_ARG, "column range out of bounds");
    if (ncols == 0) return MBAR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(c->u + col0_dev, (size_t)c->ld * sizeof(double), u_host + col0_host,
                               (size_t)ld_host * sizeof(double), (size_t)ncols * sizeof(double), (size_t)c->K,
                               hipMemcpyHostToDevice, c->stream));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

int mbar_ctx_upload_rows(mbar_ctx* c, int64_t row0, int64_t nrows, const double* rows_host, int64_t ld_host) {
    if (!c || !rows_host) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->K || ld_host < c->N) return fail(c, MBAR_ERR_ARG, "row range out of bounds");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(c->u + row0 * c->ld, (size_t)c->ld * sizeof(double), rows_host,
                               (size_t)ld_host * sizeof(double), (size_t)c->N * sizeof(double), (size_t)nrows,
                               hipMemcpyHostToDevice, c->stream));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

int mbar_ctx_copy_rows(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows) {
    if (!dst || !src) return fail(dst, MBAR_ERR_ARG, "NULL argument");
    if (dst->device != src->device || dst->N != src->N) return fail(dst, MBAR_ERR_ARG, "contexts must share device and N_local");
    if (nrows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + nrows > dst->K || src_row0 + nrows > src->K)
        return fail(dst, MBAR_ERR_ARG, "row range out of bounds");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(dst, hipSetDevice(dst->device));
    HIPCHK(dst, hipStreamSynchronize(src->stream));  // whatever produced the source rows has finished
    if (dst->ld == src->ld)  // same pitch (same N_local): the rows are one contiguous block, padding included (it is zero in both)
        HIPCHK(dst, hipMemcpyAsync(dst->u + dst_row0 * dst->ld, src->u + src_row0 * src->ld, (size_t)nrows * dst->ld * sizeof(double),
                                   hipMemcpyDeviceToDevice, dst->stream));
    else
        HIPCHK(dst, hipMemcpy2DAsync(dst->u + dst_row0 * dst->ld, (size_t)dst->ld * sizeof(double), src->u + src_row0 * src->ld,
                                     (size_t)src->ld * sizeof(double), (size_t)dst->N * sizeof(double), (size_t)nrows,
                                     hipMemcpyDeviceToDevice, dst->stream));
    dst->u_checked = false;
    dst->P_valid = false;
    dst->last_psum.clear();
    return sync_stream(dst);
}

int mbar_ctx_row_sub(mbar_ctx* c, int64_t row, const double* v_host) {
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (row < 0 || row >= c->K) return fail(c, MBAR_ERR_ARG, "row out of range");
    if (!v_host && !c->vec_tmp) return fail(c, MBAR_ERR_STATE, "mbar_ctx_row_sub: no vector has been uploaded yet");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->vec_tmp) HIPCHK(c, cache_malloc((void**)&c->vec_tmp, (size_t)c->ld * sizeof(double)));
    double* tmp = c->vec_tmp;
    if (v_host) c->vec_holds_logshift = false;
    if (v_host)  // NULL: subtract the vector of the previous call again (one observable at many states)
        HIPCHK(c, hipMemcpyAsync(tmp, v_host, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_row_sub(c->stream, c->u + row * c->ld, tmp, c->N));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

int mbar_ctx_rows_sub(mbar_ctx* c, int64_t dst_row0, int64_t src_row0, int64_t nrows, const double* v_host) {
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (nrows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + nrows > c->K || src_row0 + nrows > c->K)
        return fail(c, MBAR_ERR_ARG, "row range out of bounds");
    if (dst_row0 != src_row0 && dst_row0 < src_row0 + nrows && src_row0 < dst_row0 + nrows)
        return fail(c, MBAR_ERR_ARG, "mbar_ctx_rows_sub: the row ranges overlap");
    if (!v_host && !c->vec_tmp) return fail(c, MBAR_ERR_STATE, "mbar_ctx_rows_sub: no vector has been uploaded yet");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->vec_tmp) HIPCHK(c, cache_malloc((void**)&c->vec_tmp, (size_t)c->ld * sizeof(double)));
    if (v_host) c->vec_holds_logshift = false;
    if (v_host)  // NULL: the vector of the previous call again (one observable at many states)
        HIPCHK(c, hipMemcpyAsync(c->vec_tmp, v_host, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rows_sub(c->stream, c->u + dst_row0 * c->ld, c->u + src_row0 * c->ld, c->ld, nrows, c->vec_tmp, c->N));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

int mbar_ctx_rows_rsub(mbar_ctx* c, int64_t dst_row0, int64_t src_row0, int64_t nrows) {
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (nrows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + nrows > c->K || src_row0 + nrows > c->K)
        return fail(c, MBAR_ERR_ARG, "row range out of bounds");
    if (dst_row0 < src_row0 + nrows && src_row0 < dst_row0 + nrows) return fail(c, MBAR_ERR_ARG, "mbar_ctx_rows_rsub: the row ranges overlap");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_rows_rsub(c->stream, c->u + dst_row0 * c->ld, c->u + src_row0 * c->ld, c->ld, nrows, c->N));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

// shared tail of the two log-shift entry points: `base` holds nrows rows of raw observable values
static int logshift_rows(mbar_ctx* c, double* base, int64_t nrows, double* shift_host) {
    int rc = ensure(c, &c->scratch, &c->scratch_doubles, (size_t)nrows * 257);
    if (rc) return rc;
    double* shift_dev = c->scratch + (size_t)nrows * 256;
    HIPCHK(c, launch_rows_logshift(c->stream, base, c->ld, nrows, c->N, c->scratch, shift_dev));
    HIPCHK(c, hipMemcpyAsync(shift_host, shift_dev, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return sync_stream(c);
}

int mbar_ctx_rows_logshift(mbar_ctx* c, int64_t row0, int64_t nrows, double* shift_out) {
    if (!c || !shift_out) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->K) return fail(c, MBAR_ERR_ARG, "row range out of bounds");
    if (nrows == 0) return MBAR_OK;
    if (c->nranks > 1) return fail(c, MBAR_ERR_STATE, "mbar_ctx_rows_logshift: the minimum is taken over this context's samples only");
    HIPCHK(c, hipSetDevice(c->device));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return logshift_rows(c, c->u + row0 * c->ld, nrows, shift_out);
}

int mbar_ctx_vec_logshift(mbar_ctx* c, const double* A_host, double* shift_out) {
    if (!c || !A_host || !shift_out) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (c->nranks > 1) return fail(c, MBAR_ERR_STATE, "mbar_ctx_vec_logshift: the minimum is taken over this context's samples only");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->vec_tmp) HIPCHK(c, cache_malloc((void**)&c->vec_tmp, (size_t)c->ld * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->vec_tmp, A_host, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->vec_holds_logshift = false;
    const int lrc = logshift_rows(c, c->vec_tmp, 1, shift_out);
    c->vec_holds_logshift = lrc == MBAR_OK;
    return lrc;
}

int mbar_ctx_fill_masked_rows(mbar_ctx* c, int64_t row0, int64_t nrows, const double* v_host, const int32_t* label_host) {
    if (!c || !v_host || !label_host) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->K) return fail(c, MBAR_ERR_ARG, "row range out of bounds");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->vec_tmp) HIPCHK(c, cache_malloc((void**)&c->vec_tmp, (size_t)c->ld * sizeof(double)));
    int* dlabel = nullptr;
    HIPCHK(c, cache_malloc((void**)&dlabel, (size_t)c->N * sizeof(int)));
    c->vec_holds_logshift = false;
    hipError_t e = hipMemcpyAsync(c->vec_tmp, v_host, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dlabel, label_host, (size_t)c->N * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_fill_masked_rows(c->stream, c->u + row0 * c->ld, c->ld, c->N, nrows, c->vec_tmp, dlabel);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)cache_free(dlabel);
    if (e != hipSuccess) return fail(c, MBAR_ERR_HIP, std::string("mbar_ctx_fill_masked_rows: ") + hipGetErrorString(e));
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    flush_timers(c);
    return MBAR_OK;
}

int mbar_ctx_download_u(mbar_ctx* c, double* out, int64_t ld_out) {
    if (!c || !out || ld_out < c->N) return fail(c, MBAR_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const double* src = c->u;
    if (c->opt_debug_download_p) {  // (tests: the resident probability matrix of the last adaptive solve instead of u)
        if (!c->P || !c->P_valid) return fail(c, MBAR_ERR_STATE, "no resident probability matrix on this context");
        src = c->P;
    }
    HIPCHK(c, hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(double), src, (size_t)c->ld * sizeof(double),
                               (size_t)c->N * sizeof(double), (size_t)c->K, hipMemcpyDeviceToHost, c->stream));
    return sync_stream(c);
}

int mbar_ctx_generate_harmonic(mbar_ctx* c, uint64_t seed, const double* O_k, const double* K_k,
                               const int64_t* N_k_global, int64_t n_global0) {
    if (!c || !O_k || !K_k || !N_k_global) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int64_t> cum((size_t)c->K + 1, 0);
    for (int64_t k = 0; k < c->K; ++k) cum[k + 1] = cum[k] + N_k_global[k];
    if (n_global0 < 0 || n_global0 + c->N > cum[c->K]) return fail(c, MBAR_ERR_ARG, "shard exceeds sum(N_k)");
    double* dO = d_misc(c);
    double* dK = d_misc(c) + c->Kp;
    int64_t* dC = reinterpret_cast<int64_t*>(d_misc(c) + 2 * c->Kp);
    HIPCHK(c, hipMemcpyAsync(dO, O_k, c->K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dK, K_k, c->K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dC, cum.data(), (c->K + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    {
        ScopedTimer t(c, MBAR_TIMER_OTHER);
        HIPCHK(c, launch_generate_harmonic(c->stream, c->u, c->ld, c->N, c->K, seed, dO, dK, dC, n_global0));
    }
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
    return sync_stream(c);
}

int mbar_ctx_set_Nk(mbar_ctx* c, const double* N_k) {
    if (!c || !N_k) return fail(c, MBAR_ERR_ARG, "NULL argument");
    c->sampled.clear();
    for (int64_t k = 0; k < c->K; ++k) {
        if (!(N_k[k] >= 0.0) || !std::isfinite(N_k[k])) return fail(c, MBAR_ERR_ARG, "N_k must be finite and >= 0");
        c->Nk[k] = N_k[k];
        c->lnNk[k] = N_k[k] > 0.0 ? std::log(N_k[k]) : -std::numeric_limits<double>::infinity();
        if (N_k[k] > 0.0) c->sampled.push_back((int)k);
    }
    if (c->sampled.empty()) return fail(c, MBAR_ERR_ARG, "at least one state must have samples");
    HIPCHK(c, hipSetDevice(c->device));
    {
        int rc = drop_graphs(c);  // the captured adaptive batch bakes in the sampled-state count
        if (rc) return rc;
    }
    std::vector<double> h(2 * (size_t)c->Kp, 0.0);
    for (int64_t k = 0; k < c->Kp; ++k) {
        h[k] = k < c->K ? c->Nk[k] : 0.0;
        h[c->Kp + k] = k < c->K ? c->lnNk[k] : -std::numeric_limits<double>::infinity();
    }
    HIPCHK(c, hipMemcpyAsync(d_Nk(c), h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->P_valid = false;  // (the rows of P of states without samples are zero: the set may have changed)
    c->ld0_valid = false;
    c->last_psum.clear();
    c->have_Nk = true;
    return MBAR_OK;
}

int mbar_ctx_set_sample_weights(mbar_ctx* c, const double* c_n) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "mbar_ctx_set_sample_weights: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    bool weighted = false;
    if (c_n) {  // (validated in place: no host copy of the vector; a draw-count vector of a 1e8-sample matrix is 0.8 GB)
        bool bad = false;
        for (int64_t n = 0; n < c->N; ++n) {
            bad |= !(c_n[n] >= 0.0) || !(c_n[n] <= 1.7976931348623157e308);
            weighted |= c_n[n] != 1.0;
        }
        if (bad) return fail(c, MBAR_ERR_ARG, "sample weights must be finite and >= 0");
    }
    if (weighted && !c->lden_eff) {
        HIPCHK(c, cache_malloc((void**)&c->lden_eff, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->lden_eff, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    if (weighted)
        HIPCHK(c, hipMemcpyAsync(c->cw, c_n, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    else
        HIPCHK(c, launch_fill(c->stream, c->cw, 1.0, c->N));  // (the padding behind N stays 0)
    if (weighted) {  // sqrt(c_n) for the MFMA operands of the fused sweep (plain 0 / 1 weights are their own square roots)
        if (!c->cwsq) {
            HIPCHK(c, cache_malloc((void**)&c->cwsq, (size_t)c->ld * sizeof(double)));
            HIPCHK(c, hipMemsetAsync(c->cwsq, 0, (size_t)c->ld * sizeof(double), c->stream));
        }
        HIPCHK(c, launch_sqrt_vec(c->stream, c->cwsq, c->cw, c->N));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->weighted = weighted;
    c->last_psum.clear();
    return MBAR_OK;
}

// The layout of the bootstrap draws (runs of the states + optional position -> sample map) on the device: validated and uploaded
// ONCE per layout.  The per-replicate call then passes cumN = NULL and touches no host array of N integers.
static int upload_bootstrap_layout(mbar_ctx* c, const int64_t* cumN, int64_t K_states, const int64_t* order, const uint64_t dg[2]) {
    const int64_t total = cumN[K_states];
    const size_t words = (size_t)(K_states + 1) + (order ? (size_t)total : 0);
    if (order)
        for (int64_t p = 0; p < total; ++p)
            if (order[p] < 0 || order[p] >= total) return fail(c, MBAR_ERR_ARG, "bootstrap layout: order entry out of range");
    if (c->boot_idx && c->boot_idx_words < words) {
        (void)cache_free(c->boot_idx);
        c->boot_idx = nullptr;
    }
    if (!c->boot_idx) HIPCHK(c, cache_malloc((void**)&c->boot_idx, words * sizeof(int64_t)));
    HIPCHK(c, hipMemcpyAsync(c->boot_idx, cumN, (size_t)(K_states + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (order)
        HIPCHK(c, hipMemcpyAsync(c->boot_idx + K_states + 1, order, (size_t)total * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the host arrays may go away)
    c->boot_idx_words = words;
    c->boot_layout_digest[0] = dg[0];
    c->boot_layout_digest[1] = dg[1];
    c->boot_states = K_states;
    c->boot_total = total;
    c->boot_has_order = order != nullptr;
    return MBAR_OK;
}

static int check_bootstrap_layout(mbar_ctx* c, const int64_t* cumN, int64_t K_states) {
    if (K_states < 1) return fail(c, MBAR_ERR_ARG, "bootstrap layout: K_states must be >= 1");
    if (cumN[0] != 0) return fail(c, MBAR_ERR_ARG, "bootstrap layout: cumN[0] must be 0");
    for (int64_t k = 0; k < K_states; ++k)
        if (cumN[k + 1] < cumN[k]) return fail(c, MBAR_ERR_ARG, "bootstrap layout: cumN must not decrease");
    return MBAR_OK;
}

int mbar_ctx_set_bootstrap_layout(mbar_ctx* c, const int64_t* cumN, int64_t K_states, const int64_t* order) {
    if (!c || !cumN) return fail(c, MBAR_ERR_ARG, "NULL argument");
    int rc = check_bootstrap_layout(c, cumN, K_states);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t none[2] = {0, 0};  // (no content digest: a later call that passes its arrays again uploads again)
    return upload_bootstrap_layout(c, cumN, K_states, order, none);
}

int mbar_ctx_draw_bootstrap_weights(mbar_ctx* c, uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states,
                                    const int64_t* order, int64_t n_global0) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "mbar_ctx_draw_bootstrap_weights: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (replicate < 0 || n_global0 < 0) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (!cumN) {  // the layout of mbar_ctx_set_bootstrap_layout (or of an earlier call with arrays)
        if (!c->boot_idx || c->boot_states < 1)
            return fail(c, MBAR_ERR_STATE, "mbar_ctx_draw_bootstrap_weights: no layout on this context (mbar_ctx_set_bootstrap_layout first)");
        if (order) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: order without cumN");
    } else {
        int rc = check_bootstrap_layout(c, cumN, K_states);
        if (rc) return rc;
        // arrays handed over with the call: uploaded only when their content differs from what is there (a digest of the order
        // array is a host pass over N integers -- callers with many replicates use mbar_ctx_set_bootstrap_layout + cumN = NULL)
        const int64_t total_in = cumN[K_states];
        const size_t words = (size_t)(K_states + 1) + (order ? (size_t)total_in : 0);
        uint64_t dg[2], a[2], b[2] = {0, 0};
        (void)mbar_host_digest(cumN, (int64_t)((K_states + 1) * sizeof(int64_t)), 1, a);
        if (order) (void)mbar_host_digest(order, (int64_t)((size_t)total_in * sizeof(int64_t)), 0, b);
        dg[0] = (a[0] ^ (b[0] * 0x9E3779B97F4A7C15ull) ^ (uint64_t)words) | 1u;  // (never the "no digest" value of set_bootstrap_layout)
        dg[1] = a[1] ^ (b[1] * 0xC2B2AE3D27D4EB4Full) ^ (order ? 1u : 0u);
        if (!c->boot_idx || c->boot_idx_words != words || c->boot_layout_digest[0] != dg[0] || c->boot_layout_digest[1] != dg[1]) {
            rc = upload_bootstrap_layout(c, cumN, K_states, order, dg);
            if (rc) return rc;
        }
    }
    K_states = c->boot_states;
    const int64_t total = c->boot_total;
    const bool has_order = c->boot_has_order;
    if (total < n_global0 + c->N) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: the runs do not cover this shard");
    if (!c->lden_eff) {
        HIPCHK(c, cache_malloc((void**)&c->lden_eff, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->lden_eff, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    if (!c->cwsq) {
        HIPCHK(c, cache_malloc((void**)&c->cwsq, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->cwsq, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    HIPCHK(c, launch_fill(c->stream, c->cw, 0.0, c->N));
    HIPCHK(c, launch_bootstrap_counts(c->stream, seed, replicate, c->boot_idx, K_states, total, has_order ? c->boot_idx + K_states + 1 : nullptr,
                                      n_global0, c->N, c->cw));
    HIPCHK(c, launch_sqrt_vec(c->stream, c->cwsq, c->cw, c->N));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->weighted = true;
    c->last_psum.clear();
    return MBAR_OK;
}

int mbar_ctx_weights_from_vec(mbar_ctx* c, double power) {
    if (c && c->ext_base) return fail(c, MBAR_ERR_STATE, "mbar_ctx_weights_from_vec: an extension context holds rows only (mbar_lognum_ext / mbar_gram_w_ext sweep them)");
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (!c->vec_tmp || !c->vec_holds_logshift)
        return fail(c, MBAR_ERR_STATE, "mbar_ctx_weights_from_vec: no observable in the staging vector (mbar_ctx_vec_logshift first; "
                                       "mbar_ctx_row_sub / rows_sub / fill_masked_rows re-use that vector)");
    if (!(std::fabs(power) <= 8.0)) return fail(c, MBAR_ERR_ARG, "mbar_ctx_weights_from_vec: |power| must be <= 8");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->lden_eff) {
        HIPCHK(c, cache_malloc((void**)&c->lden_eff, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->lden_eff, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    if (!c->cwsq) {
        HIPCHK(c, cache_malloc((void**)&c->cwsq, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->cwsq, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    int* flag = reinterpret_cast<int*>(d_misc(c));
    int overflow = 0;
    HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    HIPCHK(c, launch_weights_from_log(c->stream, c->vec_tmp, power, c->N, c->cw, c->cwsq, flag));
    HIPCHK(c, hipMemcpyAsync(&overflow, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->weighted = true;
    c->last_psum.clear();
    if (overflow)  // (the weights are formed in LINEAR space here: an observable spanning more than ~1e154 overflows its square)
        return fail(c, MBAR_ERR_NUMERIC, "mbar_ctx_weights_from_vec: (A - shift)^power is not finite for some sample; use the log-space path");
    return MBAR_OK;
}

int mbar_eval(mbar_ctx* c, const double* f, int nf, unsigned flags, double* psum, double* sumlogden, double* gram) {
    if (!c || !f) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    return eval_core(c, f, nf, flags, c->logden[0], c->logden[1], psum, sumlogden, gram);
}

int mbar_ctx_set_objective_offset(mbar_ctx* c, const double* f0) {
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (!f0) {
        if (c->dn) HIPCHK(c, cache_free(c->dn));
        c->dn = nullptr;
        return MBAR_OK;
    }
    double* tmp = nullptr;
    if (!c->dn) {
        HIPCHK(c, cache_malloc((void**)&tmp, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(tmp, 0, (size_t)c->ld * sizeof(double), c->stream));
    } else {
        tmp = c->dn;
        c->dn = nullptr;
    }
    int rc = eval_core(c, f0, 1, 0, tmp, nullptr, nullptr, nullptr, nullptr);
    c->dn = tmp;
    return rc;
}

int mbar_logden(mbar_ctx* c, const double* f, double* out_n) {
    if (!c || !f || !out_n) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = eval_core(c, f, 1, 0, c->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (c->u_poison || !f_is_finite(c, f, 1)) {
        std::fill(out_n, out_n + c->N, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    HIPCHK(c, hipMemcpyAsync(out_n, c->logden[0], (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return sync_stream(c);
}

int mbar_lognum(mbar_ctx* c, const double* f, double* lognum) {
    if (!c || !f || !lognum) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = eval_core(c, f, 1, 0, c->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (c->u_poison || !f_is_finite(c, f, 1)) {
        std::fill(lognum, lognum + c->K, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    const int64_t nch = lognum_chunks(c->N, c->K);
    rc = ensure(c, &c->lognum_part, &c->lognum_part_doubles, (size_t)2 * c->K * nch + 2 * c->K);
    if (rc) return rc;
    double* pmax = c->lognum_part;
    double* psum = pmax + (size_t)c->K * nch;
    double* omax = psum + (size_t)c->K * nch;
    double* osum = omax + c->K;
    HIPCHK(c, hipMemsetAsync(d_anum(c), 0, (size_t)c->Kp * sizeof(double), c->stream));  // anum = 0
    const double* lden = c->logden[0];
    if (c->weighted) {  // log sum_n c_n exp(...) = log sum_n exp(... + ln c_n)
        HIPCHK(c, launch_shift_logden(c->stream, c->logden[0], c->cw, 1.0, c->N, c->lden_eff));
        lden = c->lden_eff;
    }
    {
        ScopedTimer t(c, MBAR_TIMER_OTHER);
        HIPCHK(c, launch_lognum(c->stream, c->u, c->ld, c->N, c->K, d_anum(c), lden, pmax, psum, nch));
        HIPCHK(c, launch_lognum_merge(c->stream, pmax, psum, c->K, nch, omax, osum));
    }
    std::vector<double> hm(c->K), hs(c->K);
    HIPCHK(c, hipMemcpyAsync(hm.data(), omax, c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs.data(), osum, c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    rc = sync_stream(c);
    if (rc) return rc;
    if (c->nranks > 1) {
        std::vector<double> gm = hm;
        for (int64_t k0 = 0; k0 < c->K; k0 += 4 * c->Kp) {
            const int64_t cnt = std::min<int64_t>(4 * c->Kp, c->K - k0);
            rc = allreduce_host(c, gm.data() + k0, cnt, 1);
            if (rc) return rc;
        }
        for (int64_t k = 0; k < c->K; ++k) hs[k] = (hm[k] == gm[k]) ? hs[k] : hs[k] * std::exp(hm[k] - gm[k]);
        for (int64_t k0 = 0; k0 < c->K; k0 += 4 * c->Kp) {
            const int64_t cnt = std::min<int64_t>(4 * c->Kp, c->K - k0);
            rc = allreduce_host(c, hs.data() + k0, cnt, 0);
            if (rc) return rc;
        }
        hm = gm;
    }
    for (int64_t k = 0; k < c->K; ++k) lognum[k] = hm[k] + std::log(hs[k]);
    return MBAR_OK;
}

static int logw_impl(mbar_ctx* c, const double* f, double* out_kn, int64_t ld_out, bool exponentiate);
int mbar_logw(mbar_ctx* c, const double* f, double* out_kn, int64_t ld_out) { return logw_impl(c, f, out_kn, ld_out, false); }
int mbar_w(mbar_ctx* c, const double* f, double* out_kn, int64_t ld_out) { return logw_impl(c, f, out_kn, ld_out, true); }
static int logw_impl(mbar_ctx* c, const double* f, double* out_kn, int64_t ld_out, bool exponentiate) {
    if (!c || !f || !out_kn || ld_out < c->N) return fail(c, MBAR_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = eval_core(c, f, 1, 0, c->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (c->u_poison || !f_is_finite(c, f, 1)) {
        for (int64_t k = 0; k < c->K; ++k) std::fill(out_kn + k * ld_out, out_kn + k * ld_out + c->N, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    // stream the result through a device staging buffer in row blocks to bound extra memory
    const int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>(c->K, (int64_t)((256ull << 20) / ((size_t)c->ld * 8))));
    double* stage = nullptr;
    HIPCHK(c, cache_malloc((void**)&stage, (size_t)rows_per * c->ld * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(d_f(c), f, c->K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int64_t k0 = 0; k0 < c->K; k0 += rows_per) {
        const int64_t nr = std::min(rows_per, c->K - k0);
        {
            ScopedTimer t(c, MBAR_TIMER_OTHER);
            hipError_t e = launch_logw(c->stream, c->u + k0 * c->ld, c->ld, c->N, nr, d_f(c) + k0, c->logden[0], stage, c->ld, exponentiate);
            if (e != hipSuccess) { (void)cache_free(stage); return fail(c, MBAR_ERR_HIP, hipGetErrorString(e)); }
        }
        hipError_t e = hipMemcpy2DAsync(out_kn + k0 * ld_out, (size_t)ld_out * sizeof(double), stage,
                                        (size_t)c->ld * sizeof(double), (size_t)c->N * sizeof(double), (size_t)nr,
                                        hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)cache_free(stage); return fail(c, MBAR_ERR_HIP, hipGetErrorString(e)); }
    }
    flush_timers(c);
    HIPCHK(c, cache_free(stage));
    return MBAR_OK;
}

int mbar_gram_w(mbar_ctx* c, const double* f, double* gramW, double* wsum) {
    if (!c || !f) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = eval_core(c, f, 1, 0, c->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (c->u_poison || !f_is_finite(c, f, 1)) {
        if (gramW) std::fill(gramW, gramW + (size_t)c->K * c->K, std::numeric_limits<double>::quiet_NaN());
        if (wsum) std::fill(wsum, wsum + c->K, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    GramPlan plan = plan_for(c);
    const size_t n_gram = plan.total_blocks * 256, total = n_gram;
    rc = ensure_red(c, total);
    if (rc) return rc;
    std::vector<double> an((size_t)c->Kp, -std::numeric_limits<double>::infinity());
    for (int64_t k = 0; k < c->K; ++k) an[k] = f[k];
    HIPCHK(c, hipMemcpyAsync(d_anum(c), an.data(), an.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = run_gram(c, d_anum(c), c->logden[0], 0, plan);
    if (rc) return rc;
    rc = allreduce_dev(c, c->red, (int64_t)total, 0);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->hred, c->red, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    rc = sync_stream(c);
    if (rc) return rc;
    std::vector<double> Gtmp;
    double* G = gramW;
    if (!G) {
        Gtmp.assign((size_t)c->K * c->K, 0.0);
        G = Gtmp.data();
    }
    std::fill(G, G + (size_t)c->K * c->K, 0.0);
    unpack_gram(plan, c->hred, c->K, G);
    if (wsum) gram_operand_sums(G, c->K, c->Nk.data(), wsum);  // sum_n W_nj = sum_k N_k (W^T W)_kj
    return MBAR_OK;
}

// ---- extension contexts: rows appended to a resident matrix WITHOUT a copy of it (the general path of the expectation family,
// pymbar/mbar.py:886-903 builds an N x (K + NL + S) host array there) -------------------------------------------------------------
// The extension holds only the new rows -- same device, N_local and row pitch as `base` -- and its storage starts a whole number
// of row pitches away from base's, so that a one-read sweep addresses [rows of base | rows of ext] as ONE 192- / 256-row panel
// (k_gram_quad_split).  K_rows real rows; the allocated rows make 16 ceil(base K / 16) + rows = 192 or 256.
int mbar_ctx_create_ext(mbar_ctx** out, mbar_ctx* base, int64_t K_rows) {
    if (!out) return fail(nullptr, MBAR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!base || K_rows < 1) return fail(base, MBAR_ERR_ARG, "mbar_ctx_create_ext: base context and K_rows >= 1");
    if (base->ext_base) return fail(base, MBAR_ERR_ARG, "mbar_ctx_create_ext: the base is an extension itself");
    const int64_t need = base->Kp + (K_rows + 15) / 16 * 16;
    if (base->Kp > 128 || need <= 128 || need > 256 || base->nranks > 1 || wide_pitch(base) || !use_fast(base) || !base->opt_quad)
        return fail(base, MBAR_ERR_ARG, "mbar_ctx_create_ext: needs a base of at most 128 states on one rank and 129 .. 256 rows in total");
    const int64_t Kp_ext = (need <= 192 ? 192 : 256) - base->Kp;
    mbar_ctx* c = new mbar_ctx();
    g_live_contexts.fetch_add(1);
    c->device = base->device;
    c->K = K_rows;
    c->Kp = Kp_ext;
    c->N = base->N;
    c->ld = base->ld;
    c->num_cu = base->num_cu;
    c->ext_base = base;
#define CRT(expr)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            int rc_ = fail(nullptr, MBAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
            mbar_ctx_destroy(c);                                                                    \
            return rc_;                                                                             \
        }                                                                                           \
    } while (0)
    CRT(hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        auto& pool = g_stream_pool[c->device];
        if (!pool.empty()) {
            c->stream = pool.back();
            pool.pop_back();
        }
    }
    if (!c->stream) CRT(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t pitch = (size_t)c->ld * sizeof(double);
    const size_t ubytes = (size_t)c->Kp * pitch;
    CRT(cache_malloc((void**)&c->u_alloc, ubytes + pitch));
    {   // first address of the allocation that is congruent to base->u modulo the row pitch
        const uintptr_t a = reinterpret_cast<uintptr_t>(c->u_alloc), b = reinterpret_cast<uintptr_t>(base->u);
        const uintptr_t r = a >= b ? (a - b) % pitch : (pitch - (b - a) % pitch) % pitch;
        c->u = reinterpret_cast<double*>(a + (r == 0 ? 0 : pitch - r));
    }
    // (only what no row operation writes is zeroed -- the padding rows and the padding columns behind N_local: the K_rows real rows
    // are whole-row uploads, copies or kernel outputs of the caller before the first sweep; zeroing 128 rows x 4e6 samples was
    // 0.8 ms of a 20 ms call)
    if (c->Kp > c->K) CRT(launch_zero(c->stream, c->u + (size_t)c->K * c->ld, (size_t)(c->Kp - c->K) * pitch));
    if (c->ld > c->N)
        CRT(hipMemset2DAsync(c->u + c->N, pitch, 0, (size_t)(c->ld - c->N) * sizeof(double), (size_t)c->K, c->stream));
    CRT(cache_malloc((void**)&c->logden[0], (size_t)3 * 16 * sizeof(double)));  // (never swept on its own: the log-denominators are the base's)
    c->logden[1] = c->logden[2] = c->logden[0];
    CRT(cache_malloc((void**)&c->small, small_doubles(256) * sizeof(double)));
    CRT(cache_host_malloc((void**)&c->hstage, (size_t)4 * 256 * sizeof(double)));
    CRT(hipMemsetAsync(c->small, 0, small_doubles(256) * sizeof(double), c->stream));
    CRT(hipStreamSynchronize(c->stream));
#undef CRT
    c->Nk.assign(K_rows, 0.0);
    c->lnNk.assign(K_rows, 0.0);
    *out = c;
    return MBAR_OK;
}

static int ext_pair_ok(mbar_ctx* ext, mbar_ctx* base, const char* who) {
    if (!ext || !base) return fail(ext, MBAR_ERR_ARG, std::string(who) + ": NULL context");
    if (ext->ext_base != base) return fail(ext, MBAR_ERR_ARG, std::string(who) + ": the first context is not an extension of the second");
    return MBAR_OK;
}
static void ext_touched(mbar_ctx* c) {
    c->u_checked = false;
    c->P_valid = false;
    c->last_psum.clear();
}

// dst rows = src rows - v (v_host, or NULL: the vector staged in dst by the previous call / mbar_ctx_vec_logshift); src is dst or its base
int mbar_ctx_rows_sub_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows, const double* v_host) {
    if (!dst || !src) return fail(dst, MBAR_ERR_ARG, "NULL argument");
    if (src != dst && dst->ext_base != src) return fail(dst, MBAR_ERR_ARG, "mbar_ctx_rows_sub_from: src must be dst or its base");
    if (nrows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + nrows > dst->K || src_row0 + nrows > src->K)
        return fail(dst, MBAR_ERR_ARG, "row range out of bounds");
    if (src == dst && dst_row0 != src_row0 && dst_row0 < src_row0 + nrows && src_row0 < dst_row0 + nrows)
        return fail(dst, MBAR_ERR_ARG, "mbar_ctx_rows_sub_from: the row ranges overlap");
    if (!v_host && !dst->vec_tmp) return fail(dst, MBAR_ERR_STATE, "mbar_ctx_rows_sub_from: no vector has been uploaded yet");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(dst, hipSetDevice(dst->device));
    if (src != dst) HIPCHK(dst, hipStreamSynchronize(src->stream));
    if (!dst->vec_tmp) HIPCHK(dst, cache_malloc((void**)&dst->vec_tmp, (size_t)dst->ld * sizeof(double)));
    if (v_host) {
        dst->vec_holds_logshift = false;
        HIPCHK(dst, hipMemcpyAsync(dst->vec_tmp, v_host, (size_t)dst->N * sizeof(double), hipMemcpyHostToDevice, dst->stream));
    }
    HIPCHK(dst, launch_rows_sub(dst->stream, dst->u + dst_row0 * dst->ld, src->u + src_row0 * src->ld, dst->ld, nrows, dst->vec_tmp, dst->N));
    ext_touched(dst);
    return sync_stream(dst);
}

// dst rows = src rows - dst rows
int mbar_ctx_rows_rsub_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* src, int64_t src_row0, int64_t nrows) {
    if (!dst || !src) return fail(dst, MBAR_ERR_ARG, "NULL argument");
    if (src != dst && dst->ext_base != src) return fail(dst, MBAR_ERR_ARG, "mbar_ctx_rows_rsub_from: src must be dst or its base");
    if (nrows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + nrows > dst->K || src_row0 + nrows > src->K)
        return fail(dst, MBAR_ERR_ARG, "row range out of bounds");
    if (src == dst && dst_row0 < src_row0 + nrows && src_row0 < dst_row0 + nrows)
        return fail(dst, MBAR_ERR_ARG, "mbar_ctx_rows_rsub_from: the row ranges overlap");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(dst, hipSetDevice(dst->device));
    if (src != dst) HIPCHK(dst, hipStreamSynchronize(src->stream));
    HIPCHK(dst, launch_rows_rsub(dst->stream, dst->u + dst_row0 * dst->ld, src->u + src_row0 * src->ld, dst->ld, nrows, dst->N));
    ext_touched(dst);
    return sync_stream(dst);
}

// dst rows = base rows [state_row0 ..) - log(base rows [obs_row0 ..) - shift_r), shift_r = min_r - |4 eps min_r| (mbar.py:827-832)
// handed back: observables that ARE rows of the resident matrix (entropy / enthalpy: the reduced potentials), each at its own
// state, in one read of the rows concerned -- no copy of them, no pass in place
// min_in (or NULL): the minima of the observable rows, as min_out of an earlier call on the same resident rows handed them back --
// the pass that finds them is skipped then; min_out (or NULL) receives them.
int mbar_ctx_rows_obs_from(mbar_ctx* dst, int64_t dst_row0, mbar_ctx* base, int64_t state_row0, int64_t obs_row0, int64_t nrows,
                           double* shift_out, const double* min_in, double* min_out) {
    int rc = ext_pair_ok(dst, base, "mbar_ctx_rows_obs_from");
    if (rc) return rc;
    if (!shift_out) return fail(dst, MBAR_ERR_ARG, "NULL argument");
    if (nrows < 0 || dst_row0 < 0 || state_row0 < 0 || obs_row0 < 0 || dst_row0 + nrows > dst->K || state_row0 + nrows > base->K ||
        obs_row0 + nrows > base->K)
        return fail(dst, MBAR_ERR_ARG, "row range out of bounds");
    if (nrows == 0) return MBAR_OK;
    HIPCHK(dst, hipSetDevice(dst->device));
    HIPCHK(dst, hipStreamSynchronize(base->stream));
    rc = ensure(dst, &dst->scratch, &dst->scratch_doubles, (size_t)nrows * 257);
    if (rc) return rc;
    double* shift_dev = dst->scratch + (size_t)nrows * 256;
    if (min_in) HIPCHK(dst, hipMemcpyAsync(dst->scratch, min_in, (size_t)nrows * sizeof(double), hipMemcpyHostToDevice, dst->stream));
    HIPCHK(dst, launch_rows_obs(dst->stream, dst->u + dst_row0 * dst->ld, base->u + obs_row0 * base->ld, base->u + state_row0 * base->ld,
                                dst->ld, nrows, dst->N, dst->scratch, shift_dev, min_in != nullptr));
    HIPCHK(dst, hipMemcpyAsync(shift_out, shift_dev, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, dst->stream));
    ext_touched(dst);
    rc = sync_stream(dst);
    if (rc) return rc;
    if (min_out) {  // shift = m - |4 eps m|  <=>  m = shift / (1 -+ 4 eps): the minimum itself is what the kernel reduces, so hand back ITS bits
        if (min_in) {
            std::copy(min_in, min_in + nrows, min_out);
        } else {
            // (the partial minima are still in the scratch rows: reduce them on the host, the same fmin)
            const int64_t want = (dst->N + 2047) / 2048;
            const int64_t gx = want < 256 ? (want < 1 ? 1 : want) : 256;
            std::vector<double> part((size_t)nrows * gx);
            HIPCHK(dst, hipMemcpy(part.data(), dst->scratch, part.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t r = 0; r < nrows; ++r) {
                double m = std::numeric_limits<double>::infinity();
                for (int64_t i = 0; i < gx; ++i) m = std::fmin(m, part[(size_t)r * gx + i]);
                min_out[r] = m;
            }
        }
    }
    return MBAR_OK;
}

// Log normalisers of the extension's rows as unsampled states of the base's mixture at f_base (K_base entries):
// lognum_ext[r] = log sum_n c_n exp(-row_rn - logden_n(f_base)) with the base's log-denominators and sample multiplicities.
int mbar_lognum_ext(mbar_ctx* ext, mbar_ctx* base, const double* f_base, double* lognum_ext) {
    int rc = ext_pair_ok(ext, base, "mbar_lognum_ext");
    if (rc) return rc;
    if (!f_base || !lognum_ext) return fail(ext, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(ext, hipSetDevice(ext->device));
    rc = eval_core(base, f_base, 1, 0, base->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return fail(ext, rc, std::string("mbar_lognum_ext: ") + mbar_last_error(base));
    rc = refresh_poison(ext);
    if (rc) return rc;
    if (base->u_poison || ext->u_poison || !f_is_finite(base, f_base, 1)) {
        std::fill(lognum_ext, lognum_ext + ext->K, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    const int64_t nch = lognum_chunks(ext->N, ext->K);
    rc = ensure(ext, &ext->lognum_part, &ext->lognum_part_doubles, (size_t)2 * ext->K * nch + 2 * ext->K);
    if (rc) return rc;
    double* pmax = ext->lognum_part;
    double* psum = pmax + (size_t)ext->K * nch;
    double* omax = psum + (size_t)ext->K * nch;
    double* osum = omax + ext->K;
    HIPCHK(ext, hipMemsetAsync(d_anum(ext), 0, (size_t)ext->Kp * sizeof(double), ext->stream));  // anum = 0
    const double* lden = base->logden[0];
    if (base->weighted) {  // log sum_n c_n exp(...) = log sum_n exp(... + ln c_n)
        HIPCHK(ext, launch_shift_logden(ext->stream, base->logden[0], base->cw, 1.0, base->N, base->lden_eff));
        lden = base->lden_eff;
    }
    {
        ScopedTimer t(ext, MBAR_TIMER_OTHER);
        HIPCHK(ext, launch_lognum(ext->stream, ext->u, ext->ld, ext->N, ext->K, d_anum(ext), lden, pmax, psum, nch));
        HIPCHK(ext, launch_lognum_merge(ext->stream, pmax, psum, ext->K, nch, omax, osum));
    }
    std::vector<double> hm(ext->K), hs(ext->K);
    HIPCHK(ext, hipMemcpyAsync(hm.data(), omax, ext->K * sizeof(double), hipMemcpyDeviceToHost, ext->stream));
    HIPCHK(ext, hipMemcpyAsync(hs.data(), osum, ext->K * sizeof(double), hipMemcpyDeviceToHost, ext->stream));
    rc = sync_stream(ext);
    if (rc) return rc;
    for (int64_t k = 0; k < ext->K; ++k) lognum_ext[k] = hm[k] + std::log(hs[k]);
    return MBAR_OK;
}

// W^T W of the weight columns of [base rows | extension rows] at (f_base, f_ext): ONE one-read sweep over the two matrices
// (gramW: (K_base + K_ext)^2 row-major; wsum as in mbar_gram_w, with N_k = 0 for the extension's rows).
// gram_base (or NULL): W^T W of the base's own states at f_base (mbar_gram_w of the base: the caller keeps it across calls).  With
// it, and at most 16 appended rows on a base of 128 padded states, only the new entries are computed: a 16 x 128 rectangle between
// the appended block row and the resident panel + the 16 x 16 block of the appended rows (a fifth of the joint sweep's matrix
// instructions; the sweep is then bound by HBM and the operands' exponentials).
int mbar_gram_w_ext(mbar_ctx* ext, mbar_ctx* base, const double* f_base, const double* f_ext, const double* gram_base, double* gramW,
                    double* wsum) {
    int rc = ext_pair_ok(ext, base, "mbar_gram_w_ext");
    if (rc) return rc;
    if (!f_base || !f_ext || !gramW) return fail(ext, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(ext, hipSetDevice(ext->device));
    rc = eval_core(base, f_base, 1, 0, base->logden[0], nullptr, nullptr, nullptr, nullptr);
    if (rc) return fail(ext, rc, std::string("mbar_gram_w_ext: ") + mbar_last_error(base));
    rc = refresh_poison(ext);
    if (rc) return rc;
    const int64_t Kb = base->K, Ke = ext->K, Kt = Kb + Ke, rows = base->Kp + ext->Kp;
    bool finite = f_is_finite(base, f_base, 1);
    for (int64_t k = 0; k < Ke; ++k) finite = finite && std::isfinite(f_ext[k]);
    if (base->u_poison || ext->u_poison || !finite) {
        std::fill(gramW, gramW + (size_t)Kt * Kt, std::numeric_limits<double>::quiet_NaN());
        if (wsum) std::fill(wsum, wsum + Kt, std::numeric_limits<double>::quiet_NaN());
        return MBAR_OK;
    }
    const double* logden_thin = base->logden[0];
    if (gram_base && Ke <= 16 && base->Kp == 128) {
        if (base->weighted) {
            HIPCHK(ext, launch_shift_logden(ext->stream, logden_thin, base->cw, 0.5, base->N, base->lden_eff));
            logden_thin = base->lden_eff;
        }
        const double ninf = -std::numeric_limits<double>::infinity();
        std::vector<double> an((size_t)128 + 16, ninf);  // [f_base padded to 128 | f_ext padded to 16]
        for (int64_t k = 0; k < Kb; ++k) an[k] = f_base[k];
        for (int64_t k = 0; k < Ke; ++k) an[(size_t)128 + k] = f_ext[k];
        double* an_dev = ext->small;
        HIPCHK(ext, hipMemcpyAsync(an_dev, an.data(), an.size() * sizeof(double), hipMemcpyHostToDevice, ext->stream));
        const int64_t ntiles = (ext->N + TS - 1) / TS;
        const int64_t row_e = (reinterpret_cast<intptr_t>(ext->u) - reinterpret_cast<intptr_t>(base->u)) / (intptr_t)((size_t)base->ld * sizeof(double));
        rc = ensure_red(ext, (size_t)9 * 256);
        if (rc) return rc;
        {   // rectangle: I = the appended block row, J = the resident panel
            LaunchGeom g = gram_geometry(144, false, ext->num_cu, ntiles, ext->opt_grid);
            const size_t rec = (size_t)8 * 256;
            rc = ensure(ext, &ext->part, &ext->part_doubles, (size_t)g.nwaves * rec);
            if (rc) return rc;
            rc = ensure(ext, &ext->scratch, &ext->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
            if (rc) return rc;
            {
                ScopedTimer t(ext, MBAR_TIMER_GRAM);
                HIPCHK(ext, launch_gram_thin(ext->stream, g, base->u, base->ld, base->N, an_dev + 128, an_dev, logden_thin, row_e, 0, ext->part));
            }
            ScopedTimer t(ext, MBAR_TIMER_REDUCE);
            HIPCHK(ext, launch_reduce(ext->stream, ext->part, g.nwaves, (int64_t)rec, ext->scratch, ext->red));
        }
        {   // the appended rows among themselves
            LaunchGeom g = gram_geometry(16, true, ext->num_cu, ntiles, ext->opt_grid);
            const size_t rec = 256;
            rc = ensure(ext, &ext->part, &ext->part_doubles, (size_t)g.nwaves * rec);
            if (rc) return rc;
            rc = ensure(ext, &ext->scratch, &ext->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
            if (rc) return rc;
            {
                ScopedTimer t(ext, MBAR_TIMER_GRAM);
                LoopCtl lo;
                HIPCHK(ext, launch_gram_diag(ext->stream, 1, g, ext->u, ext->ld, ext->N, an_dev + 128, logden_thin, 0, ext->part, nullptr, lo));
            }
            ScopedTimer t(ext, MBAR_TIMER_REDUCE);
            HIPCHK(ext, launch_reduce(ext->stream, ext->part, g.nwaves, (int64_t)rec, ext->scratch, ext->red + 8 * 256));
        }
        HIPCHK(ext, hipMemcpyAsync(ext->hred, ext->red, (size_t)9 * 256 * sizeof(double), hipMemcpyDeviceToHost, ext->stream));
        rc = sync_stream(ext);
        if (rc) return rc;
        for (int64_t i = 0; i < Kb; ++i)
            for (int64_t j = 0; j < Kb; ++j) gramW[(size_t)i * Kt + j] = gram_base[(size_t)i * Kb + j];
        for (int64_t r = 0; r < Ke; ++r) {
            for (int64_t j = 0; j < Kb; ++j) {
                const double v = ext->hred[(size_t)(j / 16) * 256 + r * 16 + (j % 16)];  // block (0, J): element (r, q)
                gramW[(size_t)(Kb + r) * Kt + j] = v;
                gramW[(size_t)j * Kt + Kb + r] = v;
            }
            for (int64_t q = 0; q < Ke; ++q) gramW[(size_t)(Kb + r) * Kt + Kb + q] = ext->hred[(size_t)8 * 256 + r * 16 + q];
        }
        if (wsum) {
            std::vector<double> Nk((size_t)Kt, 0.0);
            for (int64_t k = 0; k < Kb; ++k) Nk[k] = base->Nk[k];
            gram_operand_sums(gramW, Kt, Nk.data(), wsum);
        }
        return MBAR_OK;
    }
    const int nbt = (int)(rows / 16);
    GramPlan plan = gram_plan(rows, true);
    const size_t total = plan.total_blocks * 256;
    rc = ensure_red(ext, total);
    if (rc) return rc;
    // operand constants of the joint panel: f_base, padding, f_ext, padding (-inf: a zero operand)
    std::vector<double> an((size_t)rows, -std::numeric_limits<double>::infinity());
    for (int64_t k = 0; k < Kb; ++k) an[k] = f_base[k];
    for (int64_t k = 0; k < Ke; ++k) an[(size_t)base->Kp + k] = f_ext[k];
    double* an_dev = ext->small;  // (small_doubles(256) of them)
    HIPCHK(ext, hipMemcpyAsync(an_dev, an.data(), an.size() * sizeof(double), hipMemcpyHostToDevice, ext->stream));
    const double* logden = base->logden[0];
    if (base->weighted) {  // sum_n c_n p p^T: each operand carries sqrt(c_n), folded into the exponent
        HIPCHK(ext, launch_shift_logden(ext->stream, logden, base->cw, 0.5, base->N, base->lden_eff));
        logden = base->lden_eff;
    }
    const int64_t ntiles = (ext->N + TS - 1) / TS;
    LaunchGeom g = gram_quad_geometry(nbt, ext->num_cu, ntiles, ext->opt_grid);
    // (padding blocks at the END of the joint panel are left out of the sweep like quad_trim does for one matrix)
    g.live_blocks = ext->opt_quad_trim ? (int)((base->Kp + Ke + 15) / 16) : 0;
    const size_t rec = (size_t)plan.items[0].nblk * 256;
    rc = ensure(ext, &ext->part, &ext->part_doubles, (size_t)g.nwaves * rec);
    if (rc) return rc;
    rc = ensure(ext, &ext->scratch, &ext->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
    if (rc) return rc;
    const int64_t row_j0 = (reinterpret_cast<intptr_t>(ext->u) - reinterpret_cast<intptr_t>(base->u)) / (intptr_t)((size_t)base->ld * sizeof(double));
    {
        ScopedTimer t(ext, MBAR_TIMER_GRAM);
        HIPCHK(ext, launch_gram_quad_split(ext->stream, nbt, g, base->u, base->ld, base->N, an_dev, logden, ext->part, base->Kp, row_j0));
    }
    {
        ScopedTimer t(ext, MBAR_TIMER_REDUCE);
        HIPCHK(ext, launch_reduce(ext->stream, ext->part, g.nwaves, (int64_t)rec, ext->scratch, ext->red));
    }
    HIPCHK(ext, hipMemcpyAsync(ext->hred, ext->red, total * sizeof(double), hipMemcpyDeviceToHost, ext->stream));
    rc = sync_stream(ext);
    if (rc) return rc;
    std::vector<double> Gp((size_t)rows * rows, 0.0);
    unpack_gram(plan, ext->hred, rows, Gp.data());
    auto src = [&](int64_t k) { return k < Kb ? k : base->Kp + (k - Kb); };
    for (int64_t i = 0; i < Kt; ++i)
        for (int64_t j = 0; j < Kt; ++j) gramW[(size_t)i * Kt + j] = Gp[(size_t)src(i) * rows + src(j)];
    if (wsum) {
        std::vector<double> Nk((size_t)Kt, 0.0);
        for (int64_t k = 0; k < Kb; ++k) Nk[k] = base->Nk[k];
        gram_operand_sums(gramW, Kt, Nk.data(), wsum);
    }
    return MBAR_OK;
}

int mbar_ctx_timing(mbar_ctx* c, int which, double* total_ms, int64_t* launches) {
    if (!c || which < 0 || which >= MBAR_TIMER_COUNT) return fail(c, MBAR_ERR_ARG, "bad argument");
    if (total_ms) *total_ms = c->t_ms[which];
    if (launches) *launches = c->t_n[which];
    return MBAR_OK;
}

int mbar_ctx_timing_reset(mbar_ctx* c) {
    if (!c) return fail(c, MBAR_ERR_ARG, "NULL argument");
    for (int i = 0; i < MBAR_TIMER_COUNT; ++i) {
        c->t_ms[i] = 0.0;
        c->t_n[i] = 0;
    }
    return MBAR_OK;
}

int mbar_mfma_f64_peak(mbar_ctx* c, double* tflops) {
    if (!c || !tflops) return fail(c, MBAR_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int blocks = c->num_cu * 2, iters = 20000;
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    HIPCHK(c, launch_mfma_peak(c->stream, blocks, 100, d_misc(c)));  // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    HIPCHK(c, launch_mfma_peak(c->stream, blocks, iters, d_misc(c)));
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
    *tflops = flop / (ms * 1e-3) * 1e-12;
    return MBAR_OK;
}

}  // extern "C"

