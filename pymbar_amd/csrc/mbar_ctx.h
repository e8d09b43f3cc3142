// Internal header of the host side of libmbar_hip.so (not part of the public ABI): the context, the caching allocator, and the
// helpers the translation units of the host side share -- mbar_capi.cpp (contexts, uploads, options, evaluations), mbar_loops.cpp
// (the solver loops the reference writes in Python), mbar_comm.cpp (RCCL loader, in-process transport, all-reduce),
// mbar_host.cpp (allocator state, host-side K x K linear algebra, content digest).
#pragma once
#include "../../include/mbar_hip.h"
#include "mbar_internal.h"

#include <dlfcn.h>
#include <sched.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>


namespace mbar {
namespace host {

extern thread_local std::string g_last_error;

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (handle) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            err = std::string("dlopen(librccl) failed: ") + dlerror();
            return false;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) {
            err = "librccl is missing a required symbol";
            return false;
        }
        return true;
    }
};
extern RcclApi g_rccl;

// ---- caching allocator ----------------------------------------------------------------------------------------------
// hipMalloc / hipFree / hipHostMalloc cost 0.1 - 1 ms each (hipFree also synchronises the device): a context makes ~15
// allocations, and pymbar's real workloads (K ~ 40, N ~ 1e5: sweeps of ~10 us) build and drop contexts all the time -- the
// MBAR object, one augmented matrix per expectation call, one temporary per module-level function call.  Freed blocks are
// therefore kept (per device, bounded: MBAR_CACHE_MB, default an eighth of the device's memory -- 36 GB of 288: room for config 3's
// matrix + probability matrix or one augmented expectation matrix, while other users of the GPU keep 7/8 -- and 64 MB of
// pinned host memory; blocks of more than half the bound go straight back to the driver) and handed out again to requests of
// about the same size.  (The bound used to be 2 GB: the augmented matrix of an expectation call at K=128, N=4e6 is 6-8 GB, and
// its hipMalloc / hipFree pair cost 0.3-0.7 s per call against 15-45 ms of work.)  An allocation that fails empties the cache
// and is tried again, and mbar_cache_trim() hands everything back.  Every API call of this library leaves its stream idle
// before it frees anything, so a cached block has no work in flight.
struct MemCache {
    struct Pool {
        std::multimap<size_t, void*> free_blocks;
        size_t cached = 0, limit = 0;
    };
    std::mutex mu;
    std::map<int, Pool> dev;                       // device ordinal -> pool
    Pool pinned;
    std::unordered_map<void*, std::pair<size_t, int>> live;  // every block handed out: size, device (-1 = pinned host)
    bool configured = false;
    void configure() {
        if (configured) return;
        configured = true;
        if (const char* e = std::getenv("MBAR_CACHE_MB")) {
            dev_limit = (size_t)std::strtoull(e, nullptr, 10) << 20;
            limit_from_env = true;
        }
        pinned.limit = limit_from_env ? std::min<size_t>(dev_limit, (size_t)64 << 20) : (size_t)64 << 20;
    }
    size_t dev_limit = (size_t)2048 << 20;
    bool limit_from_env = false;
    size_t device_limit() const {  // (called with the device current)
        if (limit_from_env) return dev_limit;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            return dev_limit;
        }
        return std::max(dev_limit, total_b / 8);
    }
    static size_t round_up(size_t b) { return (b + 4095) / 4096 * 4096; }
    static void* take(Pool& p, size_t want) {
        auto it = p.free_blocks.lower_bound(want);
        if (it == p.free_blocks.end() || it->first > want + want / 4 + 65536) return nullptr;  // (no big block for a small request)
        void* q = it->second;
        p.cached -= it->first;
        p.free_blocks.erase(it);
        return q;
    }
    hipError_t alloc(void** out, size_t bytes, bool host) {
        std::lock_guard<std::mutex> lock(mu);
        configure();
        const size_t want = round_up(bytes ? bytes : 1);
        int d = -1;
        if (!host) {
            hipError_t e = hipGetDevice(&d);
            if (e != hipSuccess) return e;
        }
        Pool& p = host ? pinned : dev[d];
        if (!host && p.limit == 0) p.limit = device_limit();
        size_t got = want;
        void* q = take(p, want);
        if (q) {
            got = live[q].first;
        } else {
            hipError_t e = host ? hipHostMalloc(&q, want, hipHostMallocDefault) : hipMalloc(&q, want);
            if (e != hipSuccess && !p.free_blocks.empty()) {  // out of memory with blocks parked here: give them back, retry
                (void)hipGetLastError();
                for (auto& kv : p.free_blocks) {
                    live.erase(kv.second);
                    if (host) (void)hipHostFree(kv.second); else (void)hipFree(kv.second);
                }
                p.free_blocks.clear();
                p.cached = 0;
                e = host ? hipHostMalloc(&q, want, hipHostMallocDefault) : hipMalloc(&q, want);
            }
            if (e != hipSuccess) return e;
            live[q] = {want, d};
        }
        (void)got;
        *out = q;
        return hipSuccess;
    }
    hipError_t release(void* q) {
        if (!q) return hipSuccess;
        std::lock_guard<std::mutex> lock(mu);
        auto it = live.find(q);
        if (it == live.end()) return hipErrorInvalidValue;
        const size_t sz = it->second.first;
        const int d = it->second.second;
        Pool& p = d < 0 ? pinned : dev[d];
        if (p.cached + sz <= p.limit && sz <= p.limit / 2) {
            p.free_blocks.emplace(sz, q);
            p.cached += sz;
            return hipSuccess;
        }
        live.erase(it);
        return d < 0 ? hipHostFree(q) : hipFree(q);
    }
    // Parked device blocks beyond `keep_bytes` per device go back to the driver, largest first (called when the process's last
    // context is destroyed: a drop-in that is done with its matrices must not sit on an eighth of a shared GPU)
    void trim_to(size_t keep_bytes) {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& dp : dev) {
            Pool& p = dp.second;
            while (p.cached > keep_bytes && !p.free_blocks.empty()) {
                auto it = std::prev(p.free_blocks.end());
                live.erase(it->second);
                (void)hipSetDevice(dp.first);
                (void)hipFree(it->second);
                p.cached -= it->first;
                p.free_blocks.erase(it);
            }
        }
    }
    size_t idle_limit() {
        if (const char* e = std::getenv("MBAR_CACHE_IDLE_MB")) return (size_t)std::strtoull(e, nullptr, 10) << 20;
        return (size_t)1024 << 20;
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& dp : dev) {
            for (auto& kv : dp.second.free_blocks) {
                live.erase(kv.second);
                (void)hipSetDevice(dp.first);
                (void)hipFree(kv.second);
            }
            dp.second.free_blocks.clear();
            dp.second.cached = 0;
        }
        for (auto& kv : pinned.free_blocks) {
            live.erase(kv.second);
            (void)hipHostFree(kv.second);
        }
        pinned.free_blocks.clear();
        pinned.cached = 0;
    }
};
extern MemCache g_mem;
extern std::atomic<int> g_live_contexts;
struct DevInfo {
    int num_cu = 256;
    std::string arch;
};
extern std::mutex g_dev_mu;
extern std::map<int, DevInfo> g_dev_info;
extern std::map<int, std::vector<hipStream_t>> g_stream_pool;
inline hipError_t cache_malloc(void** p, size_t bytes) { return g_mem.alloc(p, bytes, false); }
inline hipError_t cache_free(void* p) { return g_mem.release(p); }
inline hipError_t cache_host_malloc(void** p, size_t bytes) { return g_mem.alloc(p, bytes, true); }
inline hipError_t cache_host_free(void* p) { return g_mem.release(p); }

struct TimerPair {
    hipEvent_t a, b;
    int which;
};


}  // namespace host
}  // namespace mbar
using mbar::host::TimerPair;

// In-process transport: the contexts of several caller threads on ONE device meet in a stream-ordered all-reduce (events
// across their streams, a rendezvous of the host threads per collective, no host-device synchronisation).  It drives exactly
// the code a RCCL communicator drives -- the collective sits on the compute stream, so the device-resident loop runs across
// "ranks" -- and exists so that this code can be tested on a one-GPU box (RCCL refuses two ranks on one device).
struct mbar_loopback {
    int nranks = 0, device = -1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<const double*> src;
    std::vector<int64_t> cnt;
    std::vector<int> op;
    std::vector<hipEvent_t> ready, done;
    std::vector<double*> tmp;
    std::vector<size_t> tmp_doubles;
    std::vector<int> attached;
};

struct mbar_ctx {
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    int64_t K = 0, Kp = 0, N = 0, ld = 0;
    bool have_Nk = false;
    bool u_checked = false, u_poison = false;  // NaN / -inf entries found in the matrix
    // logden[0] holds the per-sample log-denominators of THIS f (for the current matrix and N_k): the class methods ask for the
    // log-space numerators, W^T W, log W ... at the same f_k one after the other, and each would otherwise begin with the same
    // evaluation sweep (config 3: 1.9 ms each, five of them in one compute_expectations call)
    std::vector<double> ld0_f;
    bool ld0_valid = false;
    bool u_posinf = true;                      // +inf entries (legal) may be present: keep the exponentials clamped
    std::vector<double> Nk, lnNk;   // K
    std::vector<int> sampled;       // indices with N_k > 0
    // device
    double* u = nullptr;
    double* u_alloc = nullptr;  // extension contexts (mbar_ctx_create_ext): the allocation `u` points into
    mbar_ctx* ext_base = nullptr;  // ... and the context whose rows theirs are appended to
    double* logden[3] = {nullptr, nullptr, nullptr};
    double* dn = nullptr;           // objective offsets (or null)
    double* cw = nullptr;           // per-sample multiplicities (ld doubles; 1 on data, 0 on padding by default)
    double* lden_eff = nullptr;     // logden - alpha ln c for the kernels that consume logden (only when weighted)
    bool weighted = false;
    double* small = nullptr;        // aden[2][Kp] | anum[Kp] | f[Kp] | Nk[Kp] | lnNk[Kp] | delta[...]
    double* part = nullptr;         // per-wave partial records
    size_t part_doubles = 0;
    double* scratch = nullptr;      // level-1 reduction scratch
    size_t scratch_doubles = 0;
    double* red = nullptr;          // reduced outputs (contiguous: psum | obj | gram blocks)
    size_t red_doubles = 0;
    double* hred = nullptr;         // pinned host mirror of red
    double* hstage = nullptr;       // pinned staging for the small per-sweep uploads (2 Kp doubles), no sync needed
    double* lognum_part = nullptr;
    size_t lognum_part_doubles = 0;
    double* f_hist = nullptr;       // SCI f history [batch][Kp]
    double* vec_tmp = nullptr;      // staging for one N_local-vector (mbar_ctx_row_sub)
    int64_t* boot_idx = nullptr;    // bootstrap draws: cum[K + 1] | order[total] (mbar_ctx_draw_bootstrap_weights keeps the last layout)
    size_t boot_idx_words = 0;
    uint64_t boot_layout_digest[2] = {0, 0};
    int64_t boot_states = 0, boot_total = 0;  // the layout on the device: number of runs, positions in total
    bool boot_has_order = false;
    bool vec_holds_logshift = false;  // vec_tmp holds log(A - shift) of mbar_ctx_vec_logshift (and not some other call's vector)
    // captured SCI batch (launch-bound loop: 3 small kernels per iteration replayed from a hipGraph)
    hipGraphExec_t sci_graph = nullptr;
    int64_t sci_graph_batch = 0, sci_graph_sig = 0;
    double sci_graph_tol = 0.0;
    // device-resident adaptive loop: solver state (f, psum, candidates, ratio, parameters, history), control words and
    // the sampled-state list live on the device; a batch of whole iterations can be replayed from a hipGraph
    double* ad = nullptr;
    int64_t ad_hist_cap = 0;
    int* ad_ints = nullptr;         // ctl[CTL_WORDS] | sampled[Kp]
    int* h_ctl = nullptr;           // pinned mirror of the control words
    hipGraphExec_t ad_graph = nullptr;
    int64_t ad_graph_batch = 0, ad_graph_sig = 0;
    // P mode of that loop: resident probability matrix exp(a0 - u - logden(a0)), Kp x ld doubles, built once per solve
    double* P = nullptr;
    bool P_failed = false;          // the allocation did not fit: stay in the classic mode for the life of the context
    double* pm_vec = nullptr;       // a0[Kp] | ccur[Kp] | cgram[Kp]
    double* pm_ld0 = nullptr;       // host-driven loop on P (257 .. 1024 states): log-denominators at the anchor, ld doubles
    double* part_g = nullptr;       // Gram partial records of the fused-sweep loop (the psum records use `part`)
    size_t part_g_doubles = 0;
    double* cwsq = nullptr;         // sqrt of the per-sample multiplicities (only when weighted; else cw itself serves)
    double* chol = nullptr;         // workspace of the blocked Cholesky Newton solve (129 .. 256 states)
    long long* stamps = nullptr;    // MBAR_DEBUG_STAMPS: phase stamps of k_select_newton (64 launches x 16)
    // P outlives the solve that built it: a later solve on the same matrix whose start lies within the window of the anchor
    // (bootstrap replicates, protocol stages, continuation) starts with ONE fused sweep instead of the build sweep
    std::vector<double> last_psum;  // per-state sums at the f the last adaptive solve returned (empty: none)
    bool P_valid = false;
    std::vector<double> P_a0;       // anchor of the resident probability matrix: aden at the build point (Kp entries)
    // options
    int64_t opt_grid = 0, opt_force_generic = 0, opt_check_finite = 1, opt_sci_batch = 16, opt_timing = 0, opt_graph = 1, opt_small = 1, opt_wide = 1;
    int64_t opt_device_loop = 1, opt_adapt_batch = 8, opt_pmode = 1, opt_fused = 1, opt_quad = 1, opt_device_loop_wide = 1, opt_pcache = 1, opt_merge_select = 1, opt_sci_merged = 1, opt_wide_pmode = 1, opt_quad_trim = 1, opt_light_last = 1, opt_direct_results = 1, opt_debug_download_p = 0;
    int64_t opt_small_balanced = 1, opt_sci_pingpong = 1, opt_host_pmode = 2, opt_rect_waves = 8, opt_newton_ldlt = 1;

    // comm
    ncclComm_t comm = nullptr;
    mbar_loopback* loop = nullptr;  // in-process transport (tests): like comm, a collective on the compute stream
    mbar_allreduce_fn host_reduce = nullptr;
    void* host_reduce_user = nullptr;
    int rank = 0, nranks = 1;
    // timing
    std::vector<TimerPair> pending;
    std::vector<hipEvent_t> pool;
    double t_ms[MBAR_TIMER_COUNT] = {0, 0, 0, 0, 0};
    int64_t t_n[MBAR_TIMER_COUNT] = {0, 0, 0, 0, 0};
    std::string error;
};


namespace mbar {
namespace host {

int fail(mbar_ctx* c, int code, const std::string& msg);
#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(ctx, MBAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)


hipEvent_t get_event(mbar_ctx* c);
struct ScopedTimer {
    mbar_ctx* c;
    TimerPair tp;
    bool on;
    ScopedTimer(mbar_ctx* c_, int which) : c(c_), on(false) {
        tp.a = tp.b = nullptr;
        if (!c->opt_timing) return;
        tp.a = get_event(c);
        tp.b = get_event(c);
        tp.which = which;
        if (tp.a && tp.b) {
            on = hipEventRecord(tp.a, c->stream) == hipSuccess;
        }
    }
    ~ScopedTimer() {
        if (on) {
            (void)hipEventRecord(tp.b, c->stream);
            c->pending.push_back(tp);
        }
    }
};

// layout of c->small (doubles)
inline double* d_aden(mbar_ctx* c) { return c->small; }                      // [2][Kp]
inline double* d_anum(mbar_ctx* c) { return c->small + 2 * c->Kp; }          // [Kp]
inline double* d_f(mbar_ctx* c) { return c->small + 3 * c->Kp; }             // [Kp]
inline double* d_Nk(mbar_ctx* c) { return c->small + 4 * c->Kp; }            // [Kp]
inline double* d_lnNk(mbar_ctx* c) { return c->small + 5 * c->Kp; }          // [Kp]
inline double* d_delta(mbar_ctx* c) { return c->small + 6 * c->Kp; }         // [256]
inline double* d_misc(mbar_ctx* c) { return c->small + 6 * c->Kp + 256; }    // [4*Kp]
inline size_t small_doubles(int64_t Kp) { return (size_t)(10 * Kp + 256); }


// a transport whose collective is enqueued on the compute stream (no host in the loop)
inline bool stream_transport(const mbar_ctx* c) { return c->comm != nullptr || c->loop != nullptr; }

struct GramPlan {
    struct Item { bool diag; int64_t ri, rj; int nbi, nbj; int nblk; size_t off; };
    std::vector<Item> items;
    size_t total_blocks = 0;
};

// ---- shared functions (definitions: see the file list above) ----
void flush_timers(mbar_ctx* c);
int sync_stream(mbar_ctx* c);
int drop_graphs(mbar_ctx* c);
int ensure(mbar_ctx* c, double** p, size_t* have, size_t want);
int refresh_poison(mbar_ctx* c);
bool f_is_finite(const mbar_ctx* c, const double* f, int nf);
bool loop_barrier(mbar_loopback* g);
void loop_break(mbar_loopback* g);
int allreduce_loop(mbar_ctx* c, double* dev, int64_t count, int op);
int allreduce_dev(mbar_ctx* c, double* dev, int64_t count, int op);
int allreduce_host(mbar_ctx* c, double* host, int64_t count, int op);
int agree_with_rank0(mbar_ctx* c, double* v, int64_t count);
bool wide_pitch(const mbar_ctx* c);
int lse_variant_for(const mbar_ctx* c);
bool use_fast(const mbar_ctx* c);
void build_aden(const mbar_ctx* c, const double* f, double* out, int64_t rows);
bool split_sweep_ok(const mbar_ctx* c, int64_t rows);
int run_lse(mbar_ctx* c, int nf, int64_t rows, double* ld0, double* ld1, bool use_offset);
GramPlan gram_plan(int64_t Kp, bool quad = false);
GramPlan gram_plan_pmode(int64_t Kp);
bool use_quad(const mbar_ctx* c);
GramPlan plan_for(const mbar_ctx* c);
int quad_live_blocks(const mbar_ctx* c);
int run_gram(mbar_ctx* c, const double* anum_dev, const double* logden, size_t red_off, const GramPlan& plan, const double* pmat = nullptr);
void gram_operand_sums(const double* G, int64_t K, const double* w, double* out);
void unpack_gram(const GramPlan& plan, const double* blocks, int64_t K, double* G);
// psum[k] = sum_j G[k][j] from the reduced upper-triangular blocks of ONE diagonal panel of nb x 16 states (host copy; block (I, J >= I)
// at ((I nb - I (I - 1) / 2) + (J - I)) * 256, element (r, q) at r * 16 + q): with the rows of p summing to one that is sum_n c_n p_kn.
void gram_row_sums(const double* blocks, int nb, int64_t K, double* psum);
void unpack_gram_to_hessian(const GramPlan& plan, const double* blocks, int64_t K, const double* factor, const int* pos, int m, double* H,
                            int threads);
int ensure_red(mbar_ctx* c, size_t want);
int64_t lse_rows(const mbar_ctx* c);
int eval_core(mbar_ctx* c, const double* f, int nf, unsigned flags, double* ld0, double* ld1, double* psum,
              double* sumlogden, double* gram);
double now_ms();
bool chol_solve(std::vector<double>& A, std::vector<double>& b, int m);
int host_team_size(int m);
bool chol_solve_blocked(const double* A, size_t lda, std::vector<double>& b, int m, int threads);
void host_team_run(int threads, const std::function<void(int)>& fn);  // fn(0) on the caller, fn(1 .. threads-1) on the parked team
int host_team_size(int m);
void jacobi_eigh(std::vector<double> A, int m, std::vector<double>& w, std::vector<double>& V);
void newton_direction(const std::vector<double>& H, const std::vector<double>& g, int m, std::vector<double>& x);
int adaptive_host_loop(mbar_ctx* c, std::vector<double>& f, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                       int check_convergence, double* history, int64_t history_rows, mbar_solve_result& res,
                       std::vector<double>& psum, double& max_delta);
bool device_loop_eligible(const mbar_ctx* c);
int ensure_ad(mbar_ctx* c, int64_t hist_rows);
int agree_all_ok(mbar_ctx* c, bool& ok);
int adaptive_device_loop(mbar_ctx* c, std::vector<double>& f, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                         int check_convergence, double* history, int64_t history_rows, mbar_solve_result& res,
                         std::vector<double>& psum, double& max_delta, bool& handed_back);

}  // namespace host
}  // namespace mbar
