// Host side of libmbar_hip.so: the C ABI declared in include/mbar_hip.h.
//
// Owns the device-resident shard of u_kn, drives the gfx950 kernels of mbar_kernels.hip, performs the
// (tiny) cross-rank all-reduce through RCCL, and runs the solver loops that the reference writes in
// Python (adaptive(): pymbar/mbar_solvers.py:510-667).  No PyTorch, no BLAS/LAPACK: the K x K Newton
// system is solved here by a Cholesky factorisation of the gauge-fixed Hessian with a Jacobi
// pseudo-inverse fallback (minimum-norm semantics of numpy.linalg.lstsq, mbar_solvers.py:582-583).
#include "../../include/mbar_hip.h"
#include "mbar_internal.h"

#include <dlfcn.h>
#include <sched.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace mbar;

namespace {

thread_local std::string g_last_error;

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
RcclApi g_rccl;

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
MemCache g_mem;
std::atomic<int> g_live_contexts{0};
struct DevInfo {
    int num_cu = 256;
    std::string arch;
};
std::mutex g_dev_mu;
std::map<int, DevInfo> g_dev_info;
std::map<int, std::vector<hipStream_t>> g_stream_pool;
inline hipError_t cache_malloc(void** p, size_t bytes) { return g_mem.alloc(p, bytes, false); }
inline hipError_t cache_free(void* p) { return g_mem.release(p); }
inline hipError_t cache_host_malloc(void** p, size_t bytes) { return g_mem.alloc(p, bytes, true); }
inline hipError_t cache_host_free(void* p) { return g_mem.release(p); }

struct TimerPair {
    hipEvent_t a, b;
    int which;
};

}  // namespace

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
    double* part_g = nullptr;       // Gram partial records of the fused-sweep loop (the psum records use `part`)
    size_t part_g_doubles = 0;
    double* cwsq = nullptr;         // sqrt of the per-sample multiplicities (only when weighted; else cw itself serves)
    double* chol = nullptr;         // workspace of the blocked Cholesky Newton solve (129 .. 256 states)
    long long* stamps = nullptr;    // MBAR_DEBUG_STAMPS: phase stamps of k_select_newton (64 launches x 8)
    // P outlives the solve that built it: a later solve on the same matrix whose start lies within the window of the anchor
    // (bootstrap replicates, protocol stages, continuation) starts with ONE fused sweep instead of the build sweep
    std::vector<double> last_psum;  // per-state sums at the f the last adaptive solve returned (empty: none)
    bool P_valid = false;
    std::vector<double> P_a0;       // anchor of the resident probability matrix: aden at the build point (Kp entries)
    // options
    int64_t opt_grid = 0, opt_force_generic = 0, opt_check_finite = 1, opt_sci_batch = 16, opt_timing = 0, opt_graph = 1, opt_small = 1, opt_wide = 1;
    int64_t opt_device_loop = 1, opt_adapt_batch = 8, opt_pmode = 1, opt_fused = 1, opt_quad = 1, opt_device_loop_wide = 1, opt_pcache = 1, opt_merge_select = 1, opt_sci_merged = 1, opt_wide_pmode = 1, opt_quad_trim = 1, opt_light_last = 1, opt_direct_results = 1;
    int64_t opt_small_balanced = 1, opt_sci_pingpong = 1;

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

namespace {

int fail(mbar_ctx* c, int code, const std::string& msg) {
    if (c) c->error = msg;
    g_last_error = msg;
    return code;
}
#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(ctx, MBAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

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
int refresh_poison(mbar_ctx* c);

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
// layout of c->small (doubles)
inline double* d_aden(mbar_ctx* c) { return c->small; }                      // [2][Kp]
inline double* d_anum(mbar_ctx* c) { return c->small + 2 * c->Kp; }          // [Kp]
inline double* d_f(mbar_ctx* c) { return c->small + 3 * c->Kp; }             // [Kp]
inline double* d_Nk(mbar_ctx* c) { return c->small + 4 * c->Kp; }            // [Kp]
inline double* d_lnNk(mbar_ctx* c) { return c->small + 5 * c->Kp; }          // [Kp]
inline double* d_delta(mbar_ctx* c) { return c->small + 6 * c->Kp; }         // [256]
inline double* d_misc(mbar_ctx* c) { return c->small + 6 * c->Kp + 256; }    // [4*Kp]
inline size_t small_doubles(int64_t Kp) { return (size_t)(10 * Kp + 256); }

int allreduce_host(mbar_ctx* c, double* host, int64_t count, int op);
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

// ---- collectives -------------------------------------------------------------------------------
// a transport whose collective is enqueued on the compute stream (no host in the loop)
inline bool stream_transport(const mbar_ctx* c) { return c->comm != nullptr || c->loop != nullptr; }

bool loop_barrier(mbar_loopback* g) {  // rendezvous of the caller threads; false: a peer never came (or failed)
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const uint64_t my = g->gen;
    if (++g->arrived == g->nranks) {
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->gen != my || g->broken; })) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return !g->broken;
}
void loop_break(mbar_loopback* g) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
}
int allreduce_loop(mbar_ctx* c, double* dev, int64_t count, int op) {
    mbar_loopback* g = c->loop;
    const int r = c->rank;
#define LOOPCHK(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            loop_break(g);                                                                               \
            return fail(c, MBAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
        }                                                                                                \
    } while (0)
    if (g->tmp_doubles[r] < (size_t)count) {
        if (g->tmp[r]) LOOPCHK(cache_free(g->tmp[r]));
        g->tmp[r] = nullptr;
        g->tmp_doubles[r] = 0;
        LOOPCHK(cache_malloc((void**)&g->tmp[r], (size_t)count * sizeof(double)));
        g->tmp_doubles[r] = (size_t)count;
    }
    LOOPCHK(hipEventRecord(g->ready[r], c->stream));  // my contribution is complete once this event has happened
    g->src[r] = dev;
    g->cnt[r] = count;
    g->op[r] = op;
    if (!loop_barrier(g)) return fail(c, MBAR_ERR_COMM, "in-process all-reduce: a peer did not arrive");
    LoopSrc ls;
    ls.n = g->nranks;
    for (int q = 0; q < g->nranks; ++q) {
        if (g->cnt[q] != count || g->op[q] != op) {
            loop_break(g);
            return fail(c, MBAR_ERR_COMM, "in-process all-reduce: the ranks disagree on the collective (count / operation)");
        }
        ls.p[q] = g->src[q];
        if (q != r) LOOPCHK(hipStreamWaitEvent(c->stream, g->ready[q], 0));
    }
    LOOPCHK(launch_loop_reduce(c->stream, ls, count, op, g->tmp[r]));  // rank order on every rank: bit-identical results
    LOOPCHK(hipEventRecord(g->done[r], c->stream));
    if (!loop_barrier(g)) return fail(c, MBAR_ERR_COMM, "in-process all-reduce: a peer did not arrive");
    for (int q = 0; q < g->nranks; ++q)  // nobody overwrites its buffer before everybody has read it
        if (q != r) LOOPCHK(hipStreamWaitEvent(c->stream, g->done[q], 0));
    LOOPCHK(hipMemcpyAsync(dev, g->tmp[r], (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
#undef LOOPCHK
    return MBAR_OK;
}

int allreduce_dev(mbar_ctx* c, double* dev, int64_t count, int op) {
    if (c->nranks <= 1 && !c->comm) return MBAR_OK;
    if (c->loop) return allreduce_loop(c, dev, count, op);
    if (c->comm) {
        ncclResult_t r = g_rccl.AllReduce(dev, dev, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax,
                                          c->comm, c->stream);
        if (r != ncclSuccess)
            return fail(c, MBAR_ERR_COMM, std::string("ncclAllReduce: ") +
                                              (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return MBAR_OK;
    }
    if (c->host_reduce) {
        std::vector<double> h((size_t)count);
        HIPCHK(c, hipMemcpyAsync(h.data(), dev, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->host_reduce(h.data(), count, op, c->host_reduce_user) != 0)
            return fail(c, MBAR_ERR_COMM, "host all-reduce callback failed");
        HIPCHK(c, hipMemcpyAsync(dev, h.data(), count * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MBAR_OK;
    }
    return fail(c, MBAR_ERR_STATE, "nranks > 1 but no communicator attached");
}
int allreduce_host(mbar_ctx* c, double* host, int64_t count, int op) {
    if (c->nranks <= 1 && !c->comm) return MBAR_OK;
    if (c->host_reduce && !stream_transport(c)) {
        if (c->host_reduce(host, count, op, c->host_reduce_user) != 0)
            return fail(c, MBAR_ERR_COMM, "host all-reduce callback failed");
        return MBAR_OK;
    }
    double* tmp = d_misc(c);
    if (count > 4 * c->Kp) return fail(c, MBAR_ERR_ARG, "allreduce_host: buffer too large");
    HIPCHK(c, hipMemcpyAsync(tmp, host, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = allreduce_dev(c, tmp, count, op);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(host, tmp, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MBAR_OK;
}

// Make rank 0's copy of a few control values authoritative on every rank (the reduced sums are bit-identical on all
// ranks after an all-reduce, so this is insurance against a desynchronised loop exit, not a correctness need).
int agree_with_rank0(mbar_ctx* c, double* v, int64_t count) {
    if (c->nranks <= 1) return MBAR_OK;
    if (c->rank != 0)
        for (int64_t i = 0; i < count; ++i) v[i] = 0.0;
    return allreduce_host(c, v, count, 0);
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
struct GramPlan {
    struct Item { bool diag; int64_t ri, rj; int nbi, nbj; int nblk; size_t off; };
    std::vector<Item> items;
    size_t total_blocks = 0;
};
GramPlan gram_plan(int64_t Kp, bool quad = false) {
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
int run_gram(mbar_ctx* c, const double* anum_dev, const double* logden, size_t red_off, const GramPlan& plan) {
    const int64_t ntiles = (c->N + TS - 1) / TS;
    if (c->weighted) {  // sum_n c_n p p^T: each operand carries sqrt(c_n), folded into the exponent
        HIPCHK(c, launch_shift_logden(c->stream, logden, c->cw, 0.5, c->N, c->lden_eff));
        logden = c->lden_eff;
    }
    for (const auto& it : plan.items) {
        if (it.diag && it.nbi > 8) {  // one read of the matrix: the four waves of a workgroup split the panel's blocks
            LaunchGeom g = gram_quad_geometry(it.nbi, c->num_cu, ntiles, c->opt_grid);
            g.live_blocks = quad_live_blocks(c);
            const size_t rec = (size_t)it.nblk * 256;
            int rc = ensure(c, &c->part, &c->part_doubles, (size_t)g.nwaves * rec);
            if (rc) return rc;
            rc = ensure(c, &c->scratch, &c->scratch_doubles, ((size_t)g.nwaves / 32 + 1) * rec);
            if (rc) return rc;
            {
                ScopedTimer t(c, MBAR_TIMER_GRAM);
                HIPCHK(c, launch_gram_quad(c->stream, it.nbi, g, c->u, c->ld, c->N, anum_dev + it.ri, logden, c->part));
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
                HIPCHK(c, launch_gram_diag(c->stream, it.nbi, g, c->u, c->ld, c->N, anum_dev + it.ri, logden,
                                           it.ri, gp, nullptr, lo));
            }
            else
                HIPCHK(c, launch_gram_off(c->stream, it.nbj, g, c->u, c->ld, c->N, anum_dev + it.ri, anum_dev + it.rj,
                                          logden, it.ri, it.rj, gp));
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

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- dense K x K helpers (host) ---------------------------------------------------------------
// Cholesky solve of A x = b (A m x m SPD, row-major, destroyed).  Returns false on breakdown.
bool chol_solve(std::vector<double>& A, std::vector<double>& b, int m) {
    // pivots below eps * m * (largest diagonal entry) count as zero, like the singular values numpy.linalg.lstsq drops
    // (mbar_solvers.py:582, rcond = machine precision): a state whose weights underflow leaves a row of H at ~1e-300, and
    // dividing by it would throw the Newton candidate to +-inf where lstsq returns a zero component
    double dmax = 0.0;
    for (int j = 0; j < m; ++j) dmax = std::max(dmax, A[(size_t)j * m + j]);
    const double thr = dmax * std::numeric_limits<double>::epsilon() * m;
    // Right-looking, panels of 4 columns: the trailing update is a rank-4 update whose inner loops run along rows without a
    // reduction, so the compiler vectorises them as they stand (255 unknowns: 0.57 ms against 1.38 ms for the dot-product form
    // -- at 129..256 states, where the loop is host-driven, this solve was most of the time between two sweeps).
    constexpr int P = 4;
    std::vector<double> col((size_t)P * m);
    for (int j0 = 0; j0 < m; j0 += P) {
        const int j1 = std::min(j0 + P, m);
        for (int c = j0; c < j1; ++c) {
            double d = A[(size_t)c * m + c];
            if (!(d > thr) || !std::isfinite(d)) return false;
            d = std::sqrt(d);
            A[(size_t)c * m + c] = d;
            const double inv = 1.0 / d;
            double* cc = col.data() + (size_t)(c - j0) * m;
            for (int i = c + 1; i < m; ++i) cc[i] = (A[(size_t)i * m + c] *= inv);
            for (int i = c + 1; i < m; ++i) {  // the rest of the panel's columns
                const double li = cc[i];
                double* row = A.data() + (size_t)i * m;
                const int kend = std::min(i, j1 - 1);
                for (int k = c + 1; k <= kend; ++k) row[k] -= li * cc[k];
            }
        }
        if (j1 - j0 == P) {  // (a short last panel has no trailing block)
            const double *c0 = col.data(), *c1 = c0 + m, *c2 = c1 + m, *c3 = c2 + m;
            for (int i = j1; i < m; ++i) {
                const double l0 = c0[i], l1 = c1[i], l2 = c2[i], l3 = c3[i];
                double* row = A.data() + (size_t)i * m;
                for (int k = j1; k <= i; ++k) row[k] -= l0 * c0[k] + l1 * c1[k] + l2 * c2[k] + l3 * c3[k];
            }
        }
    }
    for (int i = 0; i < m; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * m + k] * b[k];
        b[i] = s / A[(size_t)i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s -= A[(size_t)k * m + i] * b[k];
        b[i] = s / A[(size_t)i * m + i];
    }
    for (int i = 0; i < m; ++i)
        if (!std::isfinite(b[i])) return false;
    return true;
}

// The same factorisation for the state counts whose K x K solve stays on the host (more than 256 states): blocks of CHOL_BLOCK
// columns, the rows below the diagonal block shared out over a team of host threads in chunks of eight (one cache line of a block
// column).  A row below the block depends on the block's own factor only (phase 1: its entries in the block's columns -- a
// triangular solve against the diagonal block) and then on the finished block columns of the rows above it (phase 2: the
// rank-CHOL_BLOCK update of its trailing entries).  Phase 2 of one block and phase 1 of the next touch the same rows, so a thread
// runs them back to back and a block costs ONE barrier; the caller's thread updates and factors the next diagonal block first and
// publishes it while the others are still in phase 2.  Every entry receives the same operations in the same order whatever the
// number of threads: results do not depend on it.  (The rank-8 row update is where the flops are: compiled a second and third time
// for AVX2 + FMA and AVX-512 and chosen at run time -- the library itself is built for baseline x86-64.)
#define MBAR_ROW_UPDATE8_BODY                                                                                                  \
    const double *c0 = cb, *c1 = cb + ms, *c2 = cb + 2 * ms, *c3 = cb + 3 * ms, *c4 = cb + 4 * ms, *c5 = cb + 5 * ms,         \
                 *c6 = cb + 6 * ms, *c7 = cb + 7 * ms;                                                                         \
    const double l0 = c0[i], l1 = c1[i], l2 = c2[i], l3 = c3[i], l4 = c4[i], l5 = c5[i], l6 = c6[i], l7 = c7[i];               \
    for (int k = k0; k <= k1; ++k)                                                                                             \
        row[k] -= ((l0 * c0[k] + l1 * c1[k]) + (l2 * c2[k] + l3 * c3[k])) + ((l4 * c4[k] + l5 * c5[k]) + (l6 * c6[k] + l7 * c7[k]));
void row_update8_base(double* __restrict__ row, const double* __restrict__ cb, size_t ms, int i, int k0, int k1) {
    MBAR_ROW_UPDATE8_BODY
}
__attribute__((target("avx2,fma")))
void row_update8_avx2(double* __restrict__ row, const double* __restrict__ cb, size_t ms, int i, int k0, int k1) {
    MBAR_ROW_UPDATE8_BODY
}
__attribute__((target("avx512f")))
void row_update8_avx512(double* __restrict__ row, const double* __restrict__ cb, size_t ms, int i, int k0, int k1) {
    MBAR_ROW_UPDATE8_BODY
}
#undef MBAR_ROW_UPDATE8_BODY
constexpr int CHOL_BLOCK = 32;
constexpr int CHOL_BLOCKED_MIN = 320;   // unknowns from which the blocked form is used ...
constexpr int CHOL_THREADED_MIN = 448;  // ... and from which it is worth a team (below: one thread, same code)
int host_team_size(int m) {
    if (m < CHOL_THREADED_MIN) return 1;
    int t = (int)std::thread::hardware_concurrency();
    {   // (the cores this process may actually run on: a container or taskset may leave it fewer than the machine has, and a
        // spinning team larger than that only takes turns)
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) t = std::min(t > 0 ? t : 1 << 20, (int)CPU_COUNT(&set));
    }
    if (const char* e = std::getenv("MBAR_HOST_THREADS")) t = std::atoi(e);
    t = std::max(1, std::min(t, 16));
    return std::min(t, std::max(1, m / 96));
}
bool chol_solve_blocked(std::vector<double>& A, std::vector<double>& b, int m, int threads) {
    double dmax = 0.0;
    for (int j = 0; j < m; ++j) dmax = std::max(dmax, A[(size_t)j * m + j]);
    const double thr = dmax * std::numeric_limits<double>::epsilon() * m;
    constexpr int B = CHOL_BLOCK;
    const size_t ms = ((size_t)m + 7) & ~(size_t)7;  // padded length of a block column: chunks of 8 rows = whole cache lines
    struct Free { void operator()(void* q) const { std::free(q); } };
    std::unique_ptr<double, Free> colmem((double*)std::aligned_alloc(64, 2 * (size_t)B * ms * sizeof(double)));
    if (!colmem) return false;
    double* const colbuf[2] = {colmem.get(), colmem.get() + (size_t)B * ms};  // colbuf[block & 1][c * ms + i] = L[i][j0 + c]
    double Lt[B * B];  // the current diagonal block's factor, transposed: Lt[c * B + k] = L[j0 + k][j0 + c]
    const auto update8 = __builtin_cpu_supports("avx512f") ? row_update8_avx512
                         : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? row_update8_avx2 : row_update8_base;
    const int T = std::max(1, threads);
    const int nblk = (m + B - 1) / B;
    std::atomic<int> diag_ready{-1}, arrived{0}, failed{0};
    auto spin_until = [&](auto&& cond) {
        int spins = 0;
        while (!cond())
            if (++spins > 8192) std::this_thread::yield();
    };
    // rows [i_lo, i_hi) owned by thread t: chunks of eight by absolute row index, dealt round-robin
    auto for_my_rows = [&](int t, int i_lo, auto&& fn) {
        for (int q = i_lo / 8; q * 8 < m; ++q) {
            if (q % T != t) continue;
            for (int i = std::max(q * 8, i_lo); i < std::min(q * 8 + 8, m); ++i) fn(i);
        }
    };
    auto phase2_row = [&](int i, int bi_prev) {  // trailing entries of row i: columns j1(prev) .. i
        const int k0 = (bi_prev + 1) * B;
        double* row = A.data() + (size_t)i * m;
        const double* cb = colbuf[bi_prev & 1];
        for (int c = 0; c < B; c += 8) update8(row, cb + (size_t)c * ms, ms, i, k0, i);
    };
    auto phase1_row = [&](int i, int bi) {  // row i of the triangular solve x L_block^T = A[i, block], column by column
        const int j0 = bi * B, jb = std::min(B, m - j0);
        double* rb = A.data() + (size_t)i * m + j0;
        double* cb = colbuf[bi & 1];
        for (int c = 0; c < jb; ++c) {
            const double v = rb[c] / Lt[c * B + c];
            rb[c] = v;
            cb[(size_t)c * ms + i] = v;
            const double* lt = Lt + c * B;  // lt[k] = L[j0 + k][j0 + c]
            for (int k = c + 1; k < jb; ++k) rb[k] -= v * lt[k];
        }
    };
    auto factor_diag = [&](int bi) -> bool {  // plain column Cholesky of the B x B block, then its transpose for phase 1
        const int j0 = bi * B, j1 = std::min(j0 + B, m);
        for (int c = j0; c < j1; ++c) {
            double* rc_ = A.data() + (size_t)c * m;
            double d = rc_[c];
            for (int k = j0; k < c; ++k) d -= rc_[k] * rc_[k];
            if (!(d > thr) || !std::isfinite(d)) return false;
            d = std::sqrt(d);
            rc_[c] = d;
            for (int i = c + 1; i < j1; ++i) {
                double* ri = A.data() + (size_t)i * m;
                double v = ri[c];
                for (int k = j0; k < c; ++k) v -= ri[k] * rc_[k];
                ri[c] = v / d;
            }
        }
        for (int c = 0; c < j1 - j0; ++c)
            for (int k = c; k < j1 - j0; ++k) Lt[c * B + k] = A[(size_t)(j0 + k) * m + j0 + c];
        return true;
    };
    auto run = [&](int t) {
        for (int bi = 0; bi < nblk; ++bi) {
            const int j1 = std::min((bi + 1) * B, m);
            if (t == 0) {
                if (bi > 0)
                    for (int i = bi * B; i < j1; ++i) phase2_row(i, bi - 1);  // the next diagonal block's rows first
                if (!factor_diag(bi)) {
                    failed.store(1, std::memory_order_release);
                    return;
                }
                diag_ready.store(bi, std::memory_order_release);
            }
            if (j1 >= m) return;  // (the last block has no rows below it)
            if (bi > 0) for_my_rows(t, j1, [&](int i) { phase2_row(i, bi - 1); });
            if (t != 0) {
                spin_until([&]() { return diag_ready.load(std::memory_order_acquire) >= bi || failed.load(std::memory_order_acquire); });
                if (failed.load(std::memory_order_acquire)) return;
            }
            for_my_rows(t, j1, [&](int i) { phase1_row(i, bi); });
            arrived.fetch_add(1, std::memory_order_acq_rel);
            spin_until([&]() { return arrived.load(std::memory_order_acquire) >= T * (bi + 1) || failed.load(std::memory_order_acquire); });
            if (failed.load(std::memory_order_acquire)) return;
        }
    };
    const bool dbg = std::getenv("MBAR_DEBUG_TIMING") != nullptr;
    const double t_begin = dbg ? now_ms() : 0.0;
    {
        std::vector<std::thread> team;
        for (int t = 1; t < T; ++t) team.emplace_back(run, t);
        run(0);
        for (auto& th : team) th.join();
    }
    if (dbg) std::fprintf(stderr, "[mbar] blocked Cholesky m=%d, %d threads: factorisation %.3f ms\n", m, T, now_ms() - t_begin);
    if (failed.load()) return false;
    for (int i = 0; i < m; ++i) {  // L y = b
        double s = b[i];
        const double* row = A.data() + (size_t)i * m;
        for (int k = 0; k < i; ++k) s -= row[k] * b[k];
        b[i] = s / row[i];
    }
    for (int i = m - 1; i >= 0; --i) {  // L^T x = y, along the rows of L
        const double* row = A.data() + (size_t)i * m;
        const double xi = b[i] / row[i];
        b[i] = xi;
        for (int k = 0; k < i; ++k) b[k] -= row[k] * xi;
    }
    for (int i = 0; i < m; ++i)
        if (!std::isfinite(b[i])) return false;
    return true;
}

// Cyclic Jacobi eigendecomposition of a symmetric matrix: A = V diag(w) V^T.
void jacobi_eigh(std::vector<double> A, int m, std::vector<double>& w, std::vector<double>& V) {
    V.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < m; ++i) {
            diag += A[(size_t)i * m + i] * A[(size_t)i * m + i];
            for (int j = i + 1; j < m; ++j) off += A[(size_t)i * m + j] * A[(size_t)i * m + j];
        }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                const double apq = A[(size_t)p * m + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < m; ++k) {
                    const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
                    A[(size_t)k * m + p] = cs * akp - sn * akq;
                    A[(size_t)k * m + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < m; ++k) {
                    const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
                    A[(size_t)p * m + k] = cs * apk - sn * aqk;
                    A[(size_t)q * m + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < m; ++k) {
                    const double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q];
                    V[(size_t)k * m + p] = cs * vkp - sn * vkq;
                    V[(size_t)k * m + q] = sn * vkp + cs * vkq;
                }
            }
    }
    w.resize(m);
    for (int i = 0; i < m; ++i) w[i] = A[(size_t)i * m + i];
}

// Newton direction: x = H^+ g - (H^+ g)[0]  (mbar_solvers.py:582-583).  H is PSD with null vector 1;
// fixing x[0] = 0 and solving the (m-1) x (m-1) SPD system gives the same vector.  If that system is
// not positive definite (disconnected states), fall back to the minimum-norm pseudo-inverse solution.
void newton_direction(const std::vector<double>& H, const std::vector<double>& g, int m, std::vector<double>& x) {
    x.assign(m, 0.0);
    if (m <= 1) return;
    const int r = m - 1;
    std::vector<double> A((size_t)r * r), b(r);
    for (int i = 0; i < r; ++i) {
        b[i] = g[i + 1];
        for (int j = 0; j < r; ++j) A[(size_t)i * r + j] = H[(size_t)(i + 1) * m + (j + 1)];
    }
    if (r >= CHOL_BLOCKED_MIN ? chol_solve_blocked(A, b, r, host_team_size(r)) : chol_solve(A, b, r)) {
        for (int i = 0; i < r; ++i) x[i + 1] = b[i];
        return;
    }
    std::vector<double> w, V;
    jacobi_eigh(H, m, w, V);
    double wmax = 0.0;
    for (double v : w) wmax = std::max(wmax, std::fabs(v));
    const double cut = wmax * std::numeric_limits<double>::epsilon() * m;
    std::vector<double> y(m, 0.0);
    for (int e = 0; e < m; ++e) {
        if (std::fabs(w[e]) <= cut) continue;
        double proj = 0.0;
        for (int k = 0; k < m; ++k) proj += V[(size_t)k * m + e] * g[k];
        proj /= w[e];
        for (int k = 0; k < m; ++k) y[k] += V[(size_t)k * m + e] * proj;
    }
    for (int k = 0; k < m; ++k) x[k] = y[k] - y[0];
}


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
    std::vector<double> gram((size_t)K * K), H((size_t)m * m), g(m), x;
    psum.assign(K, 0.0);
    int cur = 0;  // logden slot of the current f
    // initial gradient (mbar_solvers.py:570)
    int rc = eval_core(c, f.data(), 1, 0, c->logden[cur], nullptr, psum.data(), nullptr, nullptr);
    if (rc) return rc;
    bool done = false;
    const GramPlan plan = plan_for(c);
    const bool dbg = std::getenv("MBAR_DEBUG_TIMING") != nullptr;
    double tA = 0, tH = 0, tB = 0;
    const int64_t it0 = res.iterations;
    const double t0 = now_ms();
    for (int64_t it = it0; it < maxiter && !done; ++it) {
        // ---- pass A: Gram at f with the known logden -> Hessian (mbar_solvers.py:581) ----
        const double t_a0 = now_ms();
        {
            const size_t n_gram = plan.total_blocks * 256, total = n_gram;
            rc = ensure_red(c, total);
            if (rc) return rc;
            std::vector<double> an((size_t)c->Kp);
            build_aden(c, f.data(), an.data(), c->Kp);
            std::copy(an.begin(), an.end(), c->hstage + 2 * c->Kp);
            HIPCHK(c, hipMemcpyAsync(d_anum(c), c->hstage + 2 * c->Kp, an.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            rc = run_gram(c, d_anum(c), c->logden[cur], 0, plan);
            if (rc) return rc;
            rc = allreduce_dev(c, c->red, (int64_t)total, 0);
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->hred, c->red, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            rc = sync_stream(c);
            if (rc) return rc;
            unpack_gram(plan, c->hred, K, gram.data());
        }
        const double t_a1 = now_ms();
        for (int i = 0; i < m; ++i) {
            const int ki = c->sampled[i];
            g[i] = psum[ki] - c->Nk[ki];
            for (int j = 0; j < m; ++j) H[(size_t)i * m + j] = -gram[(size_t)ki * K + c->sampled[j]];
            H[(size_t)i * m + i] += psum[ki];
        }
        newton_direction(H, g, m, x);  // :582-583
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
        rc = eval_core(c, cand.data(), 2, 0, c->logden[sA], c->logden[sB], psum2.data(), nullptr, nullptr);
        if (rc) return rc;
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
        std::fprintf(stderr, "[mbar] adaptive (host loop): %lld it, per it: passA %.3f ms, host solve %.3f ms, passB %.3f ms, total %.3f ms\n",
                     (long long)nit, tA / nit, tH / nit, tB / nit, (now_ms() - t0) / nit);
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
                HIPCHK(c, launch_build_gram(c->stream, nb, gb, c->u, c->ld, c->N, d_aden(c), c->cw, c->weighted ? c->cwsq : c->cw,
                                            c->P, c->logden[0], c->part, c->part_g));
            else
                HIPCHK(c, launch_build_sweep(c->stream, nb, gb, c->u, c->ld, c->N, d_aden(c), c->cw, c->P, c->logden[0], c->part));
        }
        if (fused) {  // per-state sums and the Gram matrix at the anchor: one pair of reduction launches, ONE all-reduce
            HIPCHK(c, launch_reduce2(c->stream, c->part, Kp, c->part_g, (int64_t)rec_g, gb.nwaves, c->scratch, c->red,
                                     c->red + off_gram));
            rc = allreduce_dev(c, c->red, (int64_t)(off_gram + rec_g), 0);
        } else {
            HIPCHK(c, launch_reduce(c->stream, c->part, gb.nwaves, Kp, c->scratch, c->red));
            rc = allreduce_dev(c, c->red, Kp, 0);
        }
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->hred, c->red, (size_t)Kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        rc = sync_stream(c);
        if (rc) return rc;
        for (int64_t k = 0; k < K; ++k) psum[k] = c->hred[k];
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
    q.stamps = nullptr;
    if (std::getenv("MBAR_DEBUG_STAMPS")) {
        if (!c->stamps) HIPCHK(c, hipMalloc((void**)&c->stamps, 65 * 8 * sizeof(long long)));
        HIPCHK(c, hipMemsetAsync(c->stamps, 0, 65 * 8 * sizeof(long long), c->stream));
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
        const int64_t want = std::min(ramp, nbatch < 3 ? std::min(batch, first_batches[nbatch]) : batch);
        ++nbatch;
        ramp = std::min(batch, ramp * 2);
        int64_t nbat = std::min(want, maxiter - it);
        if (use_graph && nbat == batch) {
            rc = prepare_graph();
            if (rc) return rc;
            if (merged && need_newton) {  // (start of the solve / after a pause: the replayed iterations have no solve of their own)
                HIPCHK(c, launch_newton(c->stream, q));
                need_newton = false;
            }
            HIPCHK(c, hipGraphLaunch(c->ad_graph, c->stream));
        } else {
            for (int64_t b = 0; b < nbat; ++b) {
                rc = enqueue_iteration(c->opt_timing != 0);
                if (rc) return rc;
            }
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
        std::vector<long long> st(65 * 8);
        HIPCHK(c, hipMemcpy(st.data(), c->stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i) {
            const long long* p = st.data() + 8 * i;
            if (!p[0] || !p[5]) continue;
            std::fprintf(stderr, "[mbar] k_select_newton launch %d (shader clocks): select %lld, set-up %lld, elimination %lld, solution %lld, candidates %lld, total %lld\n",
                         i, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[5] - p[0]);
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

}  // namespace

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
    if (c->u) (void)cache_free(c->u);
    if (c->logden[0]) (void)cache_free(c->logden[0]);
    if (c->ad) (void)cache_free(c->ad);
    if (c->P) (void)cache_free(c->P);
    if (c->pm_vec) (void)cache_free(c->pm_vec);
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

int mbar_cache_trim(void) {
    g_mem.trim();
    return MBAR_OK;
}

// ---- content digest of a host buffer ------------------------------------------------------------------------------------
// The module-level functions of the reference are pure functions of their arguments (mbar_solvers.py:260-292): a caller may edit
// u_kn in place between two calls.  The Python side keeps device copies of recently seen host matrices and has to know whether
// the bytes behind an address are still the bytes it uploaded; this is that test, at memory speed on all host cores.
// 128 bits: every 1 MiB chunk runs four independent 64-bit lanes acc <- rotl(acc ^ w, 29) * ODD over its 8-byte words (a
// bijection of acc for a fixed word and injective in the word for a fixed acc, so a change of ONE word always changes its
// lane), the lanes fold into two words by maps that are injective in each lane, and the chunk digests are chained in chunk
// order by the same step with two different multipliers.  A single changed element is therefore ALWAYS detected; an arbitrary
// multi-element change escapes with probability ~2^-128.  Not cryptographic (nobody is forging matrices).
namespace {
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
constexpr uint64_t DG_M0 = 0x9E3779B97F4A7C15ull, DG_M1 = 0xC2B2AE3D27D4EB4Full, DG_M2 = 0x165667B19E3779F9ull,
                   DG_M3 = 0xD6E8FEB86659FD93ull;
constexpr int64_t DG_CHUNK = 1 << 20;

void digest_chunk(const unsigned char* p, int64_t n, uint64_t out[2]) {
    uint64_t a0 = DG_M0 ^ (uint64_t)n, a1 = DG_M1, a2 = DG_M2, a3 = DG_M3;
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        a0 = rotl64(a0 ^ w[0], 29) * DG_M1;
        a1 = rotl64(a1 ^ w[1], 29) * DG_M2;
        a2 = rotl64(a2 ^ w[2], 29) * DG_M3;
        a3 = rotl64(a3 ^ w[3], 29) * DG_M0;
    }
    if (i < n) {  // tail: zero-padded (the length is part of the seed)
        uint64_t w[4] = {0, 0, 0, 0};
        std::memcpy(w, p + i, (size_t)(n - i));
        a0 = rotl64(a0 ^ w[0], 29) * DG_M1;
        a1 = rotl64(a1 ^ w[1], 29) * DG_M2;
        a2 = rotl64(a2 ^ w[2], 29) * DG_M3;
        a3 = rotl64(a3 ^ w[3], 29) * DG_M0;
    }
    out[0] = a0 ^ rotl64(a1, 13) ^ rotl64(a2, 29) ^ rotl64(a3, 47);
    out[1] = a0 * DG_M2 + a1 * DG_M3 + a2 * DG_M0 + a3 * DG_M1;
}
}  // namespace

int mbar_host_digest(const void* data, int64_t nbytes, int threads, uint64_t* out2) {
    if ((!data && nbytes > 0) || nbytes < 0 || !out2) return fail(nullptr, MBAR_ERR_ARG, "mbar_host_digest: bad argument");
    const unsigned char* p = (const unsigned char*)data;
    const int64_t nchunks = (nbytes + DG_CHUNK - 1) / DG_CHUNK;
    std::vector<uint64_t> part((size_t)nchunks * 2);
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)nt, (int64_t)64, nchunks / 8}));  // (>= 8 MiB per thread)
    auto work = [&](int t) {
        for (int64_t c = t; c < nchunks; c += nt)
            digest_chunk(p + c * DG_CHUNK, std::min<int64_t>(DG_CHUNK, nbytes - c * DG_CHUNK), &part[(size_t)c * 2]);
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    uint64_t h0 = DG_M3 ^ (uint64_t)nbytes, h1 = DG_M2 + (uint64_t)nbytes;
    for (int64_t c = 0; c < nchunks; ++c) {
        h0 = rotl64(h0 ^ part[(size_t)c * 2], 31) * DG_M0;
        h1 = rotl64(h1 ^ part[(size_t)c * 2 + 1], 27) * DG_M1;
    }
    out2[0] = h0 ^ (h0 >> 32);
    out2[1] = h1 ^ (h1 >> 29);
    return MBAR_OK;
}

int mbar_host_newton_direction(const double* H, const double* g, int m, int threads, double* x) {
    if (!H || !g || !x || m < 1) return fail(nullptr, MBAR_ERR_ARG, "mbar_host_newton_direction: bad argument");
    std::vector<double> Hv(H, H + (size_t)m * m), gv(g, g + m), xv;
    if (threads != 0) {  // (test hook: the blocked factorisation with a given team size, whatever m; < 0: the panels-of-4 form)
        const int r = m - 1;
        std::vector<double> A((size_t)r * r), b(r);
        for (int i = 0; i < r; ++i) {
            b[i] = gv[i + 1];
            for (int j = 0; j < r; ++j) A[(size_t)i * r + j] = Hv[(size_t)(i + 1) * m + (j + 1)];
        }
        if (r > 0 && (threads > 0 ? chol_solve_blocked(A, b, r, threads) : chol_solve(A, b, r))) {
            x[0] = 0.0;
            for (int i = 0; i < r; ++i) x[i + 1] = b[i];
            return MBAR_OK;
        }
    }
    newton_direction(Hv, gv, m, xv);
    std::copy(xv.begin(), xv.end(), x);
    return MBAR_OK;
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
    else if (k == "direct_results") c->opt_direct_results = value;
    else if (k == "sci_merged") c->opt_sci_merged = value;
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
    HIPCHK(c, hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(double), c->u, (size_t)c->ld * sizeof(double),
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

int mbar_ctx_draw_bootstrap_weights(mbar_ctx* c, uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states,
                                    const int64_t* order, int64_t n_global0) {
    if (!c || !cumN) return fail(c, MBAR_ERR_ARG, "NULL argument");
    if (K_states < 1 || replicate < 0 || n_global0 < 0) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: bad argument");
    const int64_t total = cumN[K_states];
    if (cumN[0] != 0 || total < n_global0 + c->N) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: the runs do not cover this shard");
    for (int64_t k = 0; k < K_states; ++k)
        if (cumN[k + 1] < cumN[k]) return fail(c, MBAR_ERR_ARG, "mbar_ctx_draw_bootstrap_weights: cumN must not decrease");
    HIPCHK(c, hipSetDevice(c->device));
    // the layout (runs + order) goes to the device once per layout, not once per replicate
    const size_t words = (size_t)(K_states + 1) + (order ? (size_t)total : 0);
    uint64_t dg[2] = {0, 0};
    {
        uint64_t a[2], b[2] = {0, 0};
        (void)mbar_host_digest(cumN, (int64_t)((K_states + 1) * sizeof(int64_t)), 1, a);
        if (order) (void)mbar_host_digest(order, (int64_t)((size_t)total * sizeof(int64_t)), 0, b);
        dg[0] = a[0] ^ (b[0] * 0x9E3779B97F4A7C15ull) ^ (uint64_t)words;
        dg[1] = a[1] ^ (b[1] * 0xC2B2AE3D27D4EB4Full) ^ (order ? 1u : 0u);
    }
    if (!c->boot_idx || c->boot_idx_words != words || c->boot_layout_digest[0] != dg[0] || c->boot_layout_digest[1] != dg[1]) {
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
    }
    if (!c->lden_eff) {
        HIPCHK(c, cache_malloc((void**)&c->lden_eff, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->lden_eff, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    if (!c->cwsq) {
        HIPCHK(c, cache_malloc((void**)&c->cwsq, (size_t)c->ld * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->cwsq, 0, (size_t)c->ld * sizeof(double), c->stream));
    }
    HIPCHK(c, launch_fill(c->stream, c->cw, 0.0, c->N));
    HIPCHK(c, launch_bootstrap_counts(c->stream, seed, replicate, c->boot_idx, K_states, total, order ? c->boot_idx + K_states + 1 : nullptr,
                                      n_global0, c->N, c->cw));
    HIPCHK(c, launch_sqrt_vec(c->stream, c->cwsq, c->cw, c->N));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->weighted = true;
    c->last_psum.clear();
    return MBAR_OK;
}

int mbar_bootstrap_draws(uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states, const int64_t* order, int64_t* rints_out) {
    if (!cumN || !rints_out || K_states < 1 || replicate < 0 || cumN[0] != 0) return fail(nullptr, MBAR_ERR_ARG, "mbar_bootstrap_draws: bad argument");
    for (int64_t k = 0; k < K_states; ++k) {
        const int64_t start = cumN[k], nk = cumN[k + 1] - start;
        if (nk < 0) return fail(nullptr, MBAR_ERR_ARG, "mbar_bootstrap_draws: cumN must not decrease");
        for (int64_t i = 0; i < nk; ++i) {
            const int64_t pos = start + bootstrap_draw(seed, (uint64_t)replicate, (uint64_t)(start + i), (uint64_t)nk);
            const int64_t slot_sample = order ? order[start + i] : start + i;
            rints_out[slot_sample] = order ? order[pos] : pos;
        }
    }
    return MBAR_OK;
}

int mbar_ctx_weights_from_vec(mbar_ctx* c, double power) {
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

int mbar_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, MBAR_ERR_ARG, "id128 is NULL");
    std::string err;
    if (!g_rccl.load(err)) return fail(nullptr, MBAR_ERR_COMM, err);
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MBAR_ERR_COMM, "ncclGetUniqueId failed");
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return MBAR_OK;
}

int mbar_ctx_comm_init(mbar_ctx* c, const void* id128, int rank, int nranks) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MBAR_ERR_ARG, "bad argument");
    if (c->loop) return fail(c, MBAR_ERR_STATE, "the context has an in-process transport (mbar_ctx_comm_destroy first)");
    std::string err;
    if (!g_rccl.load(err)) return fail(c, MBAR_ERR_COMM, err);
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess)
        return fail(c, MBAR_ERR_COMM, std::string("ncclCommInitRank: ") +
                                          (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    c->comm = comm;
    c->rank = rank;
    c->nranks = nranks;
    c->u_checked = false;  // the NaN / -inf flag of the matrix becomes a cross-rank property
    return MBAR_OK;
}

int mbar_loopback_create(mbar_loopback** out, int nranks) {
    if (!out || nranks < 1 || nranks > 8) return fail(nullptr, MBAR_ERR_ARG, "mbar_loopback_create: 1 <= nranks <= 8");
    mbar_loopback* g = new mbar_loopback();
    g->nranks = nranks;
    g->src.assign(nranks, nullptr);
    g->cnt.assign(nranks, 0);
    g->op.assign(nranks, 0);
    g->ready.assign(nranks, nullptr);
    g->done.assign(nranks, nullptr);
    g->tmp.assign(nranks, nullptr);
    g->tmp_doubles.assign(nranks, 0);
    g->attached.assign(nranks, 0);
    *out = g;
    return MBAR_OK;
}

void mbar_loopback_destroy(mbar_loopback* g) {
    if (!g) return;
    if (g->device >= 0) (void)hipSetDevice(g->device);
    for (auto e : g->ready) if (e) (void)hipEventDestroy(e);
    for (auto e : g->done) if (e) (void)hipEventDestroy(e);
    for (auto t : g->tmp) if (t) (void)cache_free(t);
    delete g;
}

int mbar_ctx_set_loopback(mbar_ctx* c, mbar_loopback* g, int rank) {
    if (!c || !g || rank < 0 || rank >= g->nranks) return fail(c, MBAR_ERR_ARG, "bad argument");
    if (c->comm || c->host_reduce || c->loop) return fail(c, MBAR_ERR_STATE, "the context already has a transport (mbar_ctx_comm_destroy first)");
    HIPCHK(c, hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->device >= 0 && g->device != c->device) return fail(c, MBAR_ERR_ARG, "in-process transport: all contexts must be on one device");
        if (g->attached[rank]) return fail(c, MBAR_ERR_ARG, "in-process transport: rank already taken");
        g->device = c->device;
        g->attached[rank] = 1;
    }
    HIPCHK(c, hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming));
    c->loop = g;
    c->rank = rank;
    c->nranks = g->nranks;
    c->u_checked = false;
    return drop_graphs(c);
}

int mbar_ctx_comm_destroy(mbar_ctx* c) {
    if (!c) return fail(c, MBAR_ERR_ARG, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    if (c->loop) {
        std::lock_guard<std::mutex> lk(c->loop->mu);
        c->loop->attached[c->rank] = 0;
    }
    c->loop = nullptr;
    c->host_reduce = nullptr;
    c->host_reduce_user = nullptr;
    c->rank = 0;
    c->nranks = 1;
    c->u_checked = false;
    return drop_graphs(c);
}

int mbar_ctx_set_host_allreduce(mbar_ctx* c, mbar_allreduce_fn fn, void* user, int rank, int nranks) {
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || (!fn && nranks > 1)) return fail(c, MBAR_ERR_ARG, "bad argument");
    if (c->loop) return fail(c, MBAR_ERR_STATE, "the context has an in-process transport (mbar_ctx_comm_destroy first)");
    if (c->comm) {
        // the host transport REPLACES an RCCL communicator: a rank that kept issuing ncclAllReduce while its peers
        // reduce on the host would deadlock every later sweep
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->u_checked = false;
    c->host_reduce = fn;
    c->host_reduce_user = user;
    c->rank = rank;
    c->nranks = nranks;
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

int mbar_solve_adaptive(mbar_ctx* c, double* f_inout, double tol, int64_t maxiter, int64_t min_sc_iter, double gamma,
                        int check_convergence, double* history, int64_t history_rows, mbar_solve_result* result) {
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
        if (!c->stamps) HIPCHK(c, hipMalloc((void**)&c->stamps, 65 * 8 * sizeof(long long)));
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
