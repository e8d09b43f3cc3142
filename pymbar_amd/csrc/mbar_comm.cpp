// Cross-rank reduction of libmbar_hip.so: the RCCL loader (dlopen: the library has no link-time dependency on RCCL), the
// in-process transport between the contexts of several threads on one GPU (tests: it drives the code RCCL drives), the host
// transport (a callback that reduces on the host), and the all-reduce / agreement helpers the sweeps and loops call.
#include "mbar_ctx.h"

using namespace mbar;
using namespace mbar::host;

namespace mbar {
namespace host {

RcclApi g_rccl;

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


}  // namespace host
}  // namespace mbar

extern "C" {

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

}  // extern "C"
