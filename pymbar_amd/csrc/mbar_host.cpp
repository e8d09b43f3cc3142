// Host-only parts of libmbar_hip.so: the state of the caching allocator, the K x K linear algebra of the host-driven loop (Cholesky
// factorisation of the gauge-fixed Hessian -- blocked and threaded from 320 / 448 unknowns -- with a Jacobi pseudo-inverse
// fallback: the minimum-norm semantics of numpy.linalg.lstsq, mbar_solvers.py:582-583), the content digest behind the resident
// cache of host matrices, and the host face of the bootstrap stream.  Nothing here touches a kernel.
#include "mbar_ctx.h"

using namespace mbar;
using namespace mbar::host;

namespace mbar {
namespace host {

thread_local std::string g_last_error;
MemCache g_mem;
std::atomic<int> g_live_contexts{0};
std::mutex g_dev_mu;
std::map<int, DevInfo> g_dev_info;
std::map<int, std::vector<hipStream_t>> g_stream_pool;

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

// The same factorisation for the state counts whose K x K solve stays on the host (more than 256 states): right-looking, blocks of
// CHOL_BLOCK columns, on a 64-byte-aligned copy of the lower triangle whose rows are padded to whole cache lines.  The rows below
// the diagonal block are shared out over a team of host threads in chunks of eight.  A row below the block depends on the block's
// own factor only (phase 1: its entries in the block's columns -- a triangular solve against the diagonal block) and then on the
// finished block columns of the rows above it (phase 2: the rank-CHOL_BLOCK update of its trailing entries).  Phase 2 of one block
// and phase 1 of the next touch the same rows, so a thread runs them back to back and a block costs ONE barrier; the caller's
// thread updates and factors the next diagonal block first and publishes it while the others are still in phase 2.  Every entry
// receives the same operations in the same order whatever the number of threads: results do not depend on it.
// Round 5: both phases work on SEVERAL rows at once.  Phase 1 was a chain of 32 dependent divide-and-update steps per row (2.5
// GFLOP/s: a third of the factorisation's time for a tenth of its arithmetic); eight rows now walk that chain side by side, one
// vector lane each, multiplying by the pivots' reciprocals.  Phase 2 updates four rows against four block columns per pass -- 16
// multipliers stay in registers, every vector of a row and of a column that is loaded is used four times (before: one row against
// eight columns, nine loads per eight multiply-adds, rows not aligned).  511 unknowns, five threads of the MI355X boxes' host:
// 1.15 -> see profiles/r5_host_cholesky.txt.  (The kernels are compiled a second and third time for AVX2 + FMA and AVX-512 and
// chosen at run time -- the library itself is built for baseline x86-64.)
typedef double v8d __attribute__((vector_size(64)));
#define MBAR_LD8(p) ({ v8d v_; std::memcpy(&v_, (p), 64); v_; })
#define MBAR_ST8(p, v) do { v8d v_ = (v); std::memcpy((p), &v_, 64); } while (0)
#define MBAR_BC8(x) ({ const double x_ = (x); v8d{x_, x_, x_, x_, x_, x_, x_, x_}; })
constexpr int CHOL_BLOCK = 32;
// rows i .. i + 3 (row[r]; rows past the matrix: a scratch row, their multipliers are zero), entries k0 .. kend - 1 (multiples of 8;
// the overshoot past a row's diagonal lands in the unused upper triangle): row_r[k] -= sum_c L[i + r][c] L[k][c] over the block's
// columns c, taken from the block-column buffer cb[c * ms + .] in ascending c
#define MBAR_TRAIL4_BODY                                                                                                       \
    for (int c = 0; c < CHOL_BLOCK; c += 4) {                                                                                  \
        v8d l[4][4];                                                                                                           \
        for (int r = 0; r < 4; ++r)                                                                                            \
            for (int q = 0; q < 4; ++q) l[r][q] = MBAR_BC8(r < nr ? cb[(size_t)(c + q) * ms + i + r] : 0.0);                   \
        const double *c0 = cb + (size_t)c * ms, *c1 = c0 + ms, *c2 = c1 + ms, *c3 = c2 + ms;                                   \
        for (int k = k0; k < kend; k += 8) {                                                                                   \
            const v8d x0 = MBAR_LD8(c0 + k), x1 = MBAR_LD8(c1 + k), x2 = MBAR_LD8(c2 + k), x3 = MBAR_LD8(c3 + k);              \
            for (int r = 0; r < 4; ++r) {                                                                                      \
                v8d a = MBAR_LD8(row[r] + k);                                                                                  \
                a -= l[r][0] * x0;                                                                                             \
                a -= l[r][1] * x1;                                                                                             \
                a -= l[r][2] * x2;                                                                                             \
                a -= l[r][3] * x3;                                                                                             \
                MBAR_ST8(row[r] + k, a);                                                                                       \
            }                                                                                                                  \
        }                                                                                                                      \
    }
void trail4_base(double* const* row, const double* cb, size_t ms, int i, int nr, int k0, int kend) { MBAR_TRAIL4_BODY }
__attribute__((target("avx2,fma")))
void trail4_avx2(double* const* row, const double* cb, size_t ms, int i, int nr, int k0, int kend) { MBAR_TRAIL4_BODY }
__attribute__((target("avx512f")))
void trail4_avx512(double* const* row, const double* cb, size_t ms, int i, int nr, int k0, int kend) { MBAR_TRAIL4_BODY }
#undef MBAR_TRAIL4_BODY
// rows i0 .. i0 + nr - 1 (nr <= 8, i0 a multiple of 8) against the diagonal block at column j0 (jb columns; Lt[c * B + k] =
// L[j0 + k][j0 + c], inv[c] = 1 / L[j0 + c][j0 + c]): x L_block^T = A[rows, block], one row per vector lane; the result also goes
// into the block-column buffer (cb[c * ms + row], the transposed copy phase 2 reads along)
#define MBAR_SOLVE8_BODY                                                                                                       \
    v8d xt[CHOL_BLOCK];                                                                                                        \
    for (int c = 0; c < jb; ++c) {                                                                                             \
        double tmp[8];                                                                                                         \
        for (int r = 0; r < 8; ++r) tmp[r] = r < nr ? row0[(size_t)r * S + j0 + c] : 0.0;                                      \
        xt[c] = MBAR_LD8(tmp);                                                                                                 \
    }                                                                                                                          \
    for (int c = 0; c < jb; ++c) {                                                                                             \
        const v8d v = xt[c] * MBAR_BC8(inv[c]);                                                                                \
        xt[c] = v;                                                                                                             \
        const double* lt = Lt + c * CHOL_BLOCK;                                                                                \
        for (int k = c + 1; k < jb; ++k) xt[k] -= v * MBAR_BC8(lt[k]);                                                         \
    }                                                                                                                          \
    for (int c = 0; c < jb; ++c) {                                                                                             \
        double tmp[8];                                                                                                         \
        MBAR_ST8(tmp, xt[c]);                                                                                                  \
        MBAR_ST8(cb + (size_t)c * ms + i0, xt[c]);                                                                             \
        for (int r = 0; r < nr; ++r) row0[(size_t)r * S + j0 + c] = tmp[r];                                                    \
    }
void solve8_base(double* row0, size_t S, int nr, int j0, int jb, const double* Lt, const double* inv, double* cb, size_t ms, int i0) { MBAR_SOLVE8_BODY }
__attribute__((target("avx2,fma")))
void solve8_avx2(double* row0, size_t S, int nr, int j0, int jb, const double* Lt, const double* inv, double* cb, size_t ms, int i0) { MBAR_SOLVE8_BODY }
__attribute__((target("avx512f")))
void solve8_avx512(double* row0, size_t S, int nr, int j0, int jb, const double* Lt, const double* inv, double* cb, size_t ms, int i0) { MBAR_SOLVE8_BODY }
#undef MBAR_SOLVE8_BODY
// The team of host threads behind the blocked factorisation: created on first use and parked on a condition variable between
// calls (the host-driven loop factors once per iteration: creating and joining four to fifteen threads every time cost more than
// the factorisation of 511 unknowns itself).  run(T, fn): fn(0) on the caller's thread, fn(1 .. T-1) on workers; one job at a time.
class HostTeam {
public:
    HostTeam() {
        // fork(): the child has none of the parent's threads and inherits the mutexes in whatever state they were in.  The
        // handlers take both locks around the fork (no job is running, nobody is half-way through the bookkeeping), release them
        // on both sides, and the child starts over with no workers, no placement memory and fresh condition variables (the
        // parent's may have waiters recorded that do not exist in the child).
        pthread_atfork(&HostTeam::atfork_prepare, &HostTeam::atfork_parent, &HostTeam::atfork_child);
    }
    void run(int T, const std::function<void(int)>& fn) {
        if (T <= 1) {
            fn(0);
            return;
        }
        std::lock_guard<std::mutex> one_job(run_mu_);
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (pid_ != getpid()) reset_after_fork();  // (a child forked before the handlers were registered, or the first use)
            while ((int)workers_->size() < T - 1) {
                const int id = (int)workers_->size() + 1;
                workers_->emplace_back([this, id] { worker(id); });
            }
            gather_near_caller();
            job_ = &fn;
            active_ = T;
            remaining_ = T - 1;
            ++generation_;
        }
        cv_go_.notify_all();
        // (the workers hold a pointer to `fn` until the last of them is done: wait for them on EVERY way out of fn(0))
        struct WaitForWorkers {
            HostTeam* t;
            ~WaitForWorkers() {
                std::unique_lock<std::mutex> lk(t->mu_);
                t->cv_done_.wait(lk, [&] { return t->remaining_ == 0; });
                t->job_ = nullptr;
            }
        } wait_for_workers{this};
        fn(0);
    }
    // team size for a caller that would like T threads: no more than the caller's core + the free cores of its L3 domain
    int suggest(int T) {
        if (T <= 1) return T;
        std::lock_guard<std::mutex> lk(mu_);
        (void)refresh_near();
        return near_.empty() ? T : std::min<int>(T, (int)near_.size() + 1);
    }
    ~HostTeam() {
        if (!workers_ || pid_ != getpid()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_go_.notify_all();
        for (auto& w : *workers_) w.join();
        delete workers_;
    }

private:
    // The team exchanges a block column (0.1-0.3 MB) per block: between cores of different L3 domains (8 cores each on the EPYC hosts
    // of the MI355X boxes) that goes through memory.  The workers are therefore kept on the CPUs that share the L3 cache with the
    // CPU the caller is on right now (one worker per physical core while they last); the caller's own thread is never touched.
    // Skipped when the domain cannot be read or the process may not run on enough of its CPUs.  MBAR_HOST_TEAM_AFFINITY=0: off.
    // (call with mu_ held) CPUs for the workers near the caller's current CPU -> near_; true if they changed
    bool refresh_near() {
        static const bool enabled = [] { const char* e = std::getenv("MBAR_HOST_TEAM_AFFINITY"); return !e || std::atoi(e) != 0; }();
        if (!enabled) return false;
        const int cpu = sched_getcpu();
        if (cpu < 0) return false;
        auto read_list = [](const std::string& path, std::vector<int>& out) {
            out.clear();
            FILE* fh = std::fopen(path.c_str(), "r");
            if (!fh) return;
            char buf[4096];
            if (std::fgets(buf, sizeof(buf), fh)) {
                for (char* p = buf; *p;) {
                    char* end;
                    const long a = std::strtol(p, &end, 10);
                    if (end == p) break;
                    long b2 = a;
                    if (*end == '-') b2 = std::strtol(end + 1, &end, 10);
                    for (long v = a; v <= b2 && v < CPU_SETSIZE; ++v) out.push_back((int)v);
                    p = (*end == ',') ? end + 1 : end;
                    if (*end != ',') break;
                }
            }
            std::fclose(fh);
        };
        const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
        std::vector<int> dom;
        read_list(base + "/cache/index3/shared_cpu_list", dom);
        if (dom.empty()) return false;
        const int key = dom.front();
        if (key == near_key_) return false;
        near_key_ = key;
        near_.clear();
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
        // one CPU per physical core first (the first sibling of every core of the domain), the caller's core last
        std::vector<int> firsts, sib;
        for (int cid : dom) {
            if (!CPU_ISSET(cid, &allowed)) continue;
            read_list("/sys/devices/system/cpu/cpu" + std::to_string(cid) + "/topology/thread_siblings_list", sib);
            if (!sib.empty() && sib.front() != cid) continue;
            if (!sib.empty() && std::find(sib.begin(), sib.end(), cpu) != sib.end()) continue;  // (the caller's core)
            firsts.push_back(cid);
        }
        near_ = firsts;
        near_domain_.clear();
        for (int cid : dom)
            if (CPU_ISSET(cid, &allowed)) near_domain_.push_back(cid);
        return true;
    }
    void gather_near_caller() {
        (void)refresh_near();
        if (near_.empty() || (pin_key_ == near_key_ && pinned_workers_ == workers_->size())) return;
        pin_key_ = near_key_;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
        // (the whole domain for every worker, not one CPU each: on a shared host the scheduler must be able to step around a core
        // somebody else keeps busy -- a worker nailed to such a core stalls the whole team at its barriers)
        cpu_set_t domain;
        CPU_ZERO(&domain);
        for (int cid : near_domain_) CPU_SET(cid, &domain);
        for (size_t w = 0; w < workers_->size(); ++w) {
            const cpu_set_t& set = w < near_.size() ? domain : allowed;  // (more workers than cores in the domain: the rest anywhere)
            (void)pthread_setaffinity_np((*workers_)[w].native_handle(), sizeof(set), &set);
        }
        pinned_workers_ = workers_->size();
    }
    std::vector<int> near_, near_domain_;  // free physical cores of the caller's L3 domain (their first CPUs) / all of its allowed CPUs
    int near_key_ = -1, pin_key_ = -2;
    size_t pinned_workers_ = 0;
    void worker(int id) {
        int seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_go_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (id >= active_) continue;
            const std::function<void(int)>* f = job_;
            lk.unlock();
            (*f)(id);
            lk.lock();
            if (--remaining_ == 0) cv_done_.notify_one();
        }
    }
    // (call with mu_ held, or from the fork handler of the child) no workers, no placement memory, counters at rest
    void reset_after_fork() {
        workers_ = new std::vector<std::thread>();  // (the parent's handles are leaked: their threads do not exist here)
        pid_ = getpid();
        pin_key_ = -2;
        near_key_ = -1;
        pinned_workers_ = 0;
        near_.clear();
        near_domain_.clear();
        generation_ = active_ = remaining_ = 0;
        stop_ = false;
        job_ = nullptr;
    }
    static void atfork_prepare();
    static void atfork_parent();
    static void atfork_child();
    std::mutex run_mu_, mu_;
    std::condition_variable cv_go_, cv_done_;
    std::vector<std::thread>* workers_ = nullptr;
    const std::function<void(int)>* job_ = nullptr;
    int generation_ = 0, active_ = 0, remaining_ = 0;
    bool stop_ = false;
    pid_t pid_ = 0;
};
HostTeam g_team;
void HostTeam::atfork_prepare() {
    g_team.run_mu_.lock();
    g_team.mu_.lock();
}
void HostTeam::atfork_parent() {
    g_team.mu_.unlock();
    g_team.run_mu_.unlock();
}
void HostTeam::atfork_child() {
    if (g_team.workers_) g_team.reset_after_fork();
    new (&g_team.cv_go_) std::condition_variable();
    new (&g_team.cv_done_) std::condition_variable();
    g_team.mu_.unlock();
    g_team.run_mu_.unlock();
}

constexpr int CHOL_BLOCKED_MIN = 64;    // unknowns from which the blocked form is used (round 5: it wins from here on, 4.5x at 255) ...
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
    return g_team.suggest(std::min(t, std::max(1, m / 96)));
}
void host_team_run(int threads, const std::function<void(int)>& fn) { g_team.run(threads, fn); }
// A: m x m, row pitch lda, only its lower triangle is read (and nothing of it written); b: in the right-hand side, out the solution.
bool chol_solve_blocked(const double* A, size_t lda, std::vector<double>& b, int m, int threads) {
    double dmax = 0.0;
    for (int j = 0; j < m; ++j) dmax = std::max(dmax, A[(size_t)j * lda + j]);
    const double thr = dmax * std::numeric_limits<double>::epsilon() * m;
    constexpr int B = CHOL_BLOCK;
    // The right-hand side rides along as one more row BELOW the matrix (row mv, in a chunk of its own): what the two phases leave
    // in it is y = L^-1 b -- the forward substitution, done by the team as part of the factorisation.
    const int mv = (m + 7) & ~7;
    const size_t ms = (size_t)mv + 8;  // padded length of a row / of a block column: whole cache lines
    const size_t S = ms;
    struct Free { void operator()(void* q) const { std::free(q); } };
    // [the lower triangle, rows padded][two block-column buffers][one scratch row per thread]
    const int T = std::max(1, threads);
    // (the workspace outlives the call: a fresh 2-8 MB allocation is page-faulted in on first touch, which costs as much as a
    // third of the factorisation -- the host-driven loop calls this once per iteration from the same thread)
    static thread_local std::unique_ptr<double, Free> mem;
    static thread_local size_t mem_doubles = 0;
    const size_t want = (size_t)(mv + 1) * S + 2 * (size_t)B * ms + (size_t)T * S;
    if (mem_doubles < want) {
        mem.reset((double*)std::aligned_alloc(64, ((want * sizeof(double) + 63) / 64) * 64));
        mem_doubles = mem ? want : 0;
        if (!mem) return false;
    }
    double* const L = mem.get();
    double* const colbuf[2] = {L + (size_t)(mv + 1) * S, L + (size_t)(mv + 1) * S + (size_t)B * ms};  // colbuf[block & 1][c * ms + i] = L[i][j0 + c]
    double* const scratch = colbuf[1] + (size_t)B * ms;
    std::memset(colbuf[0], 0, (2 * (size_t)B * ms + (size_t)T * S) * sizeof(double));
    double Lt[B * B], inv[B];  // the current diagonal block's factor, transposed: Lt[c * B + k] = L[j0 + k][j0 + c]; reciprocal pivots
    const bool avx512 = __builtin_cpu_supports("avx512f"), avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    const auto trail4 = avx512 ? trail4_avx512 : avx2 ? trail4_avx2 : trail4_base;
    const auto solve8 = avx512 ? solve8_avx512 : avx2 ? solve8_avx2 : solve8_base;
    const int nblk = (m + B - 1) / B;
    const int vchunk = mv / 8;  // the chunk of the right-hand side's row
    std::atomic<int> diag_ready{-1}, arrived{0}, failed{0}, copied{0};
    auto spin_until = [&](auto&& cond) {
        int spins = 0;
        while (!cond())
            if (++spins > 8192) std::this_thread::yield();
    };
    // chunks of eight rows by absolute row index, dealt round-robin: fn(first row, rows) for thread t's chunks from row i_lo on
    auto for_my_chunks = [&](int t, int i_lo, auto&& fn) {
        for (int q = i_lo / 8; q <= vchunk; ++q)
            if (q % T == t) fn(q * 8, q == vchunk ? 1 : std::min(8, m - q * 8));
    };
    // trailing entries (columns from the end of block bi_prev on) of the rows i0 .. i0 + n - 1, n <= 8, in groups of four
    auto phase2_rows = [&](int t, int i0, int n, int bi_prev) {
        const int k0 = (bi_prev + 1) * B;
        const double* cb = colbuf[bi_prev & 1];
        for (int g = 0; g < n; g += 4) {
            const int i = i0 + g, nr = std::min(4, n - g);
            double* row[4];
            for (int r = 0; r < 4; ++r) row[r] = r < nr ? L + (size_t)(i + r) * S : scratch + (size_t)t * S;
            trail4(row, cb, ms, i, nr, k0, std::min((i + nr + 7) & ~7, mv));
        }
    };
    auto factor_diag = [&](int bi) -> bool {  // plain column Cholesky of the B x B block, then its transpose for phase 1
        const int j0 = bi * B, j1 = std::min(j0 + B, m);
        for (int c = j0; c < j1; ++c) {
            double* rc_ = L + (size_t)c * S;
            double d = rc_[c];
            for (int k = j0; k < c; ++k) d -= rc_[k] * rc_[k];
            if (!(d > thr) || !std::isfinite(d)) return false;
            d = std::sqrt(d);
            rc_[c] = d;
            inv[c - j0] = 1.0 / d;
            for (int i = c + 1; i < j1; ++i) {
                double* ri = L + (size_t)i * S;
                double v = ri[c];
                for (int k = j0; k < c; ++k) v -= ri[k] * rc_[k];
                ri[c] = v * inv[c - j0];
            }
        }
        for (int c = 0; c < j1 - j0; ++c)
            for (int k = c; k < j1 - j0; ++k) Lt[c * B + k] = L[(size_t)(j0 + k) * S + j0 + c];
        return true;
    };
    const bool dbg = std::getenv("MBAR_DEBUG_TIMING") != nullptr;
    double tp_diag = 0, tp_p1 = 0, tp_p2 = 0, tp_wait = 0;  // (thread 0's phases, MBAR_DEBUG_TIMING)
    auto run = [&](int t) {
        const bool tm = dbg && t == 0;
        double q0 = tm ? now_ms() : 0.0, q1;
        auto lap = [&](double& into) { if (tm) { q1 = now_ms(); into += q1 - q0; q0 = q1; } };
        // own chunks of the lower triangle into the aligned copy (first touch by the thread that works on them)
        for_my_chunks(t, 0, [&](int i0, int n) {
            for (int i = i0; i < i0 + n; ++i) {
                double* dst = L + (size_t)i * S;
                const size_t len = i == mv ? (size_t)m : (size_t)(i + 1);
                std::memcpy(dst, i == mv ? b.data() : A + (size_t)i * lda, len * sizeof(double));
                std::memset(dst + len, 0, (S - len) * sizeof(double));
            }
        });
        copied.fetch_add(1, std::memory_order_acq_rel);
        spin_until([&]() { return copied.load(std::memory_order_acquire) >= T; });
        lap(tp_wait);
        for (int bi = 0; bi < nblk; ++bi) {
            const int j1 = std::min((bi + 1) * B, m);
            if (t == 0) {
                if (bi > 0)
                    for (int i0 = bi * B; i0 < j1; i0 += 8) phase2_rows(0, i0, std::min(8, j1 - i0), bi - 1);  // the next diagonal block's rows first
                lap(tp_p2);
                if (!factor_diag(bi)) {
                    failed.store(1, std::memory_order_release);
                    return;
                }
                diag_ready.store(bi, std::memory_order_release);
                lap(tp_diag);
            }
            const int i_lo = j1 >= m ? mv : j1;  // (below the last block: the right-hand side's row only)
            if (bi > 0) for_my_chunks(t, i_lo, [&](int i0, int n) { phase2_rows(t, i0, n, bi - 1); });
            lap(tp_p2);
            if (t != 0) {
                spin_until([&]() { return diag_ready.load(std::memory_order_acquire) >= bi || failed.load(std::memory_order_acquire); });
                if (failed.load(std::memory_order_acquire)) return;
            }
            for_my_chunks(t, i_lo, [&](int i0, int n) { solve8(L + (size_t)i0 * S, S, n, bi * B, j1 - bi * B, Lt, inv, colbuf[bi & 1], ms, i0); });
            lap(tp_p1);
            arrived.fetch_add(1, std::memory_order_acq_rel);
            spin_until([&]() { return arrived.load(std::memory_order_acquire) >= T * (bi + 1) || failed.load(std::memory_order_acquire); });
            lap(tp_wait);
            if (failed.load(std::memory_order_acquire)) return;
        }
    };
    const double t_begin = dbg ? now_ms() : 0.0;
    g_team.run(T, run);
    const double t_fact = dbg ? now_ms() : 0.0;
    if (failed.load()) return false;
    std::memcpy(b.data(), L + (size_t)mv * S, (size_t)m * sizeof(double));  // y = L^-1 b
    for (int i = m - 1; i >= 0; --i) {  // L^T x = y, along the rows of L
        const double* row = L + (size_t)i * S;
        const double xi = b[i] / row[i];
        b[i] = xi;
        for (int k = 0; k < i; ++k) b[k] -= row[k] * xi;
    }
    if (dbg)
        std::fprintf(stderr, "[mbar] blocked Cholesky m=%d, %d threads: factorisation %.3f ms (thread 0: diagonal blocks %.3f, rows against the block %.3f, trailing updates %.3f, copy + waiting %.3f), substitutions %.3f ms\n",
                     m, T, t_fact - t_begin, tp_diag, tp_p1, tp_p2, tp_wait, now_ms() - t_fact);
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
    std::vector<double> b(g.begin() + 1, g.end());
    bool ok;
    if (r >= CHOL_BLOCKED_MIN) {  // (the blocked form works on its own aligned copy of the lower triangle: straight from H)
        ok = chol_solve_blocked(H.data() + m + 1, (size_t)m, b, r, host_team_size(r));
    } else {
        std::vector<double> A((size_t)r * r);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) A[(size_t)i * r + j] = H[(size_t)(i + 1) * m + (j + 1)];
        ok = chol_solve(A, b, r);
    }
    if (ok) {
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



}  // namespace host
}  // namespace mbar

extern "C" {

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
        if (r > 0 && (threads > 0 ? chol_solve_blocked(A.data(), (size_t)r, b, r, threads) : chol_solve(A, b, r))) {
            x[0] = 0.0;
            for (int i = 0; i < r; ++i) x[i + 1] = b[i];
            return MBAR_OK;
        }
    }
    newton_direction(Hv, gv, m, xv);
    std::copy(xv.begin(), xv.end(), x);
    return MBAR_OK;
}

int mbar_bootstrap_draws(uint64_t seed, int64_t replicate, const int64_t* cumN, int64_t K_states, const int64_t* order, int64_t* rints_out) {
    if (!cumN || !rints_out || K_states < 1 || replicate < 0 || cumN[0] != 0) return fail(nullptr, MBAR_ERR_ARG, "mbar_bootstrap_draws: bad argument");
    for (int64_t k = 0; k < K_states; ++k)
        if (cumN[k + 1] < cumN[k]) return fail(nullptr, MBAR_ERR_ARG, "mbar_bootstrap_draws: cumN must not decrease");
    if (order)  // (the positions index the output: a stray entry would write outside it)
        for (int64_t p = 0; p < cumN[K_states]; ++p)
            if (order[p] < 0 || order[p] >= cumN[K_states]) return fail(nullptr, MBAR_ERR_ARG, "mbar_bootstrap_draws: order entry out of range");
    for (int64_t k = 0; k < K_states; ++k) {
        const int64_t start = cumN[k], nk = cumN[k + 1] - start;
        for (int64_t i = 0; i < nk; ++i) {
            const int64_t pos = start + bootstrap_draw(seed, (uint64_t)replicate, (uint64_t)(start + i), (uint64_t)nk);
            const int64_t slot_sample = order ? order[start + i] : start + i;
            rints_out[slot_sample] = order ? order[pos] : pos;
        }
    }
    return MBAR_OK;
}

}  // extern "C"
