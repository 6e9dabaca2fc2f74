// Multi-GPU forms of the pair-grid entry points (include/pilot_ot.h, section "multi-GPU").
//
// The N^2 pair problems of pilotpy/tools/Trajectory.py:505-515 are independent given the replicated N x K proportions and
// K x K cost, so the grid is PARTITIONED, never exchanged: shard s of G solves rows s, s+G, s+2G, ... (round-robin: the
// per-pair update counts are ragged and, in exact mode with a symmetric cost, only columns >= row are solved, so cyclic
// rows balance both) against all N columns.  The only exchange step is assembling the finished matrix: ONE all-gather
// of the row blocks over RCCL (xGMI), then a device-side row interleave.
//
// Two ways to run it, same kernels and same bytes either way:
//   * pilot_ot_multi_*: ONE process drives G devices (ncclCommInitAll, one plan + one stream per device, grouped
//     ncclAllGather) -- what tl.wasserstein_distance(engine_options={"n_devices": G}) uses;
//   * pilot_ot_comm_*: one process PER device (ncclCommInitRank with a unique id the host language passes around),
//     each calling the single-device *_dev entry points on its own rows -- what bench.py uses under a launcher.
// librccl is loaded lazily (dlopen) by the first call that needs it, so single-GPU users never pay for it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "abi_common.hpp"

namespace {

#define fail(...) pilot::abi_fail(__VA_ARGS__)

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

std::mutex g_rccl_mutex;
void rccl_log_to_stderr();

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);      // first use may come from several host threads at once
    if (g_rccl.h) return PILOT_OT_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    rccl_log_to_stderr();                                   // (before the library reads its environment)
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(PILOT_OT_ERCCL, "librccl not found: %s", dlerror());
    RcclApi a;
    a.h = h;
#define PILOT_SYM(field, name)                                                                   \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                               \
    if (!a.field) { dlclose(h); return fail(PILOT_OT_ERCCL, "librccl lacks %s", name); }
    PILOT_SYM(GetUniqueId, "ncclGetUniqueId")
    PILOT_SYM(CommInitRank, "ncclCommInitRank")
    PILOT_SYM(CommInitAll, "ncclCommInitAll")
    PILOT_SYM(CommDestroy, "ncclCommDestroy")
    PILOT_SYM(AllGather, "ncclAllGather")
    PILOT_SYM(AllReduce, "ncclAllReduce")
    PILOT_SYM(GroupStart, "ncclGroupStart")
    PILOT_SYM(GroupEnd, "ncclGroupEnd")
    PILOT_SYM(CommCount, "ncclCommCount")
    PILOT_SYM(CommUserRank, "ncclCommUserRank")
    PILOT_SYM(GetErrorString, "ncclGetErrorString")
#undef PILOT_SYM
    g_rccl = a;
    return PILOT_OT_OK;
}

#define RCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess)                                                                               \
            return fail(PILOT_OT_ERCCL, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, \
                        __LINE__);                                                                           \
    } while (0)

// rank-major stage (G blocks of n_pad rows; block w row t = grid row w + t*G)  ->  full N x N matrix
__global__ void interleave_rows_kernel(const double *__restrict__ stage, int G, int n_pad, int N,
                                       double *__restrict__ full) {
    const long total = (long)N * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / N), col = (int)(idx % N);
        full[idx] = stage[((size_t)(row % G) * n_pad + row / G) * N + col];
    }
}

int launch_interleave(const double *stage, int G, int n_pad, int N, double *full, hipStream_t s) {
    long blocks = ((long)N * N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(interleave_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, stage, G, n_pad, N, full);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

// RCCL writes its NCCL_DEBUG lines to stdout unless NCCL_DEBUG_FILE names another sink: before librccl is loaded its log is
// pointed at stderr through RCCL's OWN switch (a caller that sets NCCL_DEBUG_FILE itself, or PILOT_OT_KEEP_RCCL_STDOUT=1,
// keeps RCCL's default).  The five-line version banner of RCCL 2.27 is a plain printf that no switch reaches; the library
// no longer swaps fd 1 around communicator creation as round 3 did (process-wide, and it moved whatever another thread wrote
// to stdout in those milliseconds): it only flushes stdio right after the communicator exists, so the banner appears THEN and
// not at exit behind the caller's own output.  A host program whose stdout is a data channel redirects it itself around the
// call (pilot_amd/multi.py::stdout_to_stderr, what bench.py does).
void rccl_log_to_stderr() {
    const char *keep = getenv("PILOT_OT_KEEP_RCCL_STDOUT");
    if (keep && *keep && *keep != '0') return;
    (void)setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);      // (0: an existing setting wins)
}

struct DeviceGuard {   // restores the calling thread's current device
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

}  // namespace

// ================================================================================================
// one process per device
struct pilot_ot_comm {
    ncclComm_t comm;
    int n_ranks, rank, device;
};

PILOT_API int pilot_ot_comm_unique_id(char *uid) {
    if (!uid) return fail(PILOT_OT_EINVAL, "uid is NULL");
    static_assert(sizeof(ncclUniqueId) == PILOT_OT_UNIQUE_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_load();
    if (rc != PILOT_OT_OK) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(uid, &id, sizeof(id));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_comm_init_rank(const char *uid, int n_ranks, int rank, pilot_ot_comm **comm) {
    if (!uid || !comm) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(PILOT_OT_EINVAL, "rank %d of %d", rank, n_ranks);
    int rc = rccl_load();
    if (rc != PILOT_OT_OK) return rc;
    pilot_ot_comm *c = new (std::nothrow) pilot_ot_comm();
    if (!c) return fail(PILOT_OT_EINVAL, "out of host memory");
    c->n_ranks = n_ranks; c->rank = rank; c->comm = nullptr;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { delete c; return fail(PILOT_OT_EHIP, "hipGetDevice: %s", hipGetErrorString(e)); }
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    ncclResult_t r;
    {
        r = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
        fflush(stdout);
    }
    if (r != ncclSuccess) { delete c; return fail(PILOT_OT_ERCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    *comm = c;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_comm_info(pilot_ot_comm *c, int *n_ranks, int *rank) {
    if (!c || !n_ranks || !rank) return fail(PILOT_OT_EINVAL, "NULL pointer");
    RCCL_TRY(g_rccl.CommCount(c->comm, n_ranks));       // what RCCL itself reports for this communicator
    RCCL_TRY(g_rccl.CommUserRank(c->comm, rank));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_comm_destroy(pilot_ot_comm *c) {
    if (!c) return PILOT_OT_OK;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_comm_all_gather_rows(pilot_ot_comm *c, const double *d_local, int n_pad, int N, double *d_stage,
                                            double *d_full, void *stream) {
    if (!c || !d_local || !d_stage || !d_full) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || n_pad != (N + c->n_ranks - 1) / c->n_ranks)
        return fail(PILOT_OT_EINVAL, "n_pad=%d must be ceil(N / n_ranks) = %d", n_pad, (N + c->n_ranks - 1) / c->n_ranks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    RCCL_TRY(g_rccl.AllGather(d_local, d_stage, (size_t)n_pad * N, ncclDouble, c->comm, s));
    return launch_interleave(d_stage, c->n_ranks, n_pad, N, d_full, s);
}

PILOT_API int pilot_ot_comm_all_reduce_max(pilot_ot_comm *c, double *d_vals, int n, void *stream) {
    if (!c || !d_vals || n < 1) return fail(PILOT_OT_EINVAL, "bad argument");
    RCCL_TRY(g_rccl.AllReduce(d_vals, d_vals, (size_t)n, ncclDouble, ncclMax, c->comm, static_cast<hipStream_t>(stream)));
    return PILOT_OT_OK;
}

// ================================================================================================
// one process, G devices
namespace {
struct Shard {
    int device = 0, n_local = 0;
    pilot_ot_plan *plan = nullptr;
    hipStream_t stream = nullptr;
    double *dP = nullptr, *dM = nullptr, *dLocal = nullptr, *dErr = nullptr, *dStage = nullptr, *dFull = nullptr;
    int *dIt = nullptr, *dFl = nullptr;
    hipEvent_t ev_begin = nullptr, ev_grid = nullptr, ev_gather = nullptr, ev_end = nullptr;
    ncclComm_t comm = nullptr;
};

// One host thread per shard.  A call's per-shard work is a handful of stream launches (memset, prep, scatter, two or three
// kernels, events): ~40 us of host time, the same order as a c3 shard's kernel at G = 8, so ONE thread enqueueing shard
// after shard would leave the last device idle for the first seven's enqueue time.  The calling thread posts the job to
// every worker, the workers enqueue on their own streams concurrently, the caller waits for the ENQUEUES (not for the
// GPUs) and then issues the gather.  PILOT_OT_MULTI_SERIAL=1 runs the jobs on the calling thread instead (A/B switch).
class ShardWorkers {
public:
    explicit ShardWorkers(int n) : w_(n) {
        for (int i = 0; i < n; ++i) w_[i].reset(new W());
        for (int i = 0; i < n; ++i) w_[i]->th = std::thread([this, i] { loop(*w_[i]); });
    }
    ~ShardWorkers() {
        for (auto &w : w_) {
            { std::lock_guard<std::mutex> l(w->mu); w->quit = true; }
            w->cv.notify_all();
        }
        for (auto &w : w_) if (w->th.joinable()) w->th.join();
    }
    // run job(s) for every shard s on its worker; returns the first non-zero status (its message becomes the caller's)
    int run(const std::function<int(int)> &job) {
        for (size_t i = 0; i < w_.size(); ++i) {
            W &w = *w_[i];
            { std::lock_guard<std::mutex> l(w.mu); w.job = &job; w.idx = (int)i; w.done = false; }
            w.cv.notify_all();
        }
        int rc = PILOT_OT_OK;
        for (auto &wp : w_) {
            W &w = *wp;
            std::unique_lock<std::mutex> l(w.mu);
            w.cv.wait(l, [&] { return w.done; });
            if (w.rc != PILOT_OT_OK && rc == PILOT_OT_OK) rc = pilot::abi_fail(w.rc, "%s", w.msg.c_str());
        }
        return rc;
    }
private:
    struct W {
        std::thread th; std::mutex mu; std::condition_variable cv;
        const std::function<int(int)> *job = nullptr; int idx = 0; bool done = true, quit = false; int rc = 0; std::string msg;
    };
    static void loop(W &w) {
        for (;;) {
            std::unique_lock<std::mutex> l(w.mu);
            w.cv.wait(l, [&] { return w.quit || (w.job && !w.done); });
            if (w.quit) return;
            const std::function<int(int)> *job = w.job;
            const int idx = w.idx;
            l.unlock();
            const int rc = (*job)(idx);
            l.lock();
            w.rc = rc;
            w.msg = rc != PILOT_OT_OK ? pilot_ot_last_error() : "";
            w.job = nullptr; w.done = true;
            l.unlock();
            w.cv.notify_all();
        }
    }
    std::vector<std::unique_ptr<W>> w_;
};
}  // namespace

struct pilot_ot_multi {
    int N = 0, K = 0, G = 0, n_pad = 0, gather = 0;
    std::vector<Shard> sh;
    bool ran = false, exact = false;
    double max_cost = 1.0;                 // max(M) of the current inputs (set_inputs): precision / fallback decisions
    hipEvent_t ev_copied = nullptr;        // peer-copy gather: shard 0 has read every shard's rows (next call may overwrite them)
    bool copied_valid = false;
    std::unique_ptr<ShardWorkers> workers; // nullptr: jobs run on the calling thread
    int for_each_shard(const std::function<int(int)> &job) {
        if (workers) return workers->run(job);
        for (int s = 0; s < G; ++s) { const int rc = job(s); if (rc != PILOT_OT_OK) return rc; }
        return PILOT_OT_OK;
    }
};

namespace {
void multi_free(pilot_ot_multi *m) {
    if (!m) return;
    m->workers.reset();         // joins the shard threads
    DeviceGuard guard;
    if (m->ev_copied) { (void)hipSetDevice(m->sh[0].device); (void)hipEventDestroy(m->ev_copied); }
    for (Shard &s : m->sh) {
        (void)hipSetDevice(s.device);
        if (s.comm) (void)g_rccl.CommDestroy(s.comm);
        if (s.plan) (void)pilot_ot_plan_destroy(s.plan);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        for (void *p : {(void *)s.dP, (void *)s.dM, (void *)s.dLocal, (void *)s.dErr, (void *)s.dStage, (void *)s.dFull,
                        (void *)s.dIt, (void *)s.dFl})
            if (p) (void)hipFree(p);
        for (hipEvent_t e : {s.ev_begin, s.ev_grid, s.ev_gather, s.ev_end})
            if (e) (void)hipEventDestroy(e);
    }
    delete m;
}
}  // namespace

PILOT_API int pilot_ot_multi_create(int N, int K, const int *devices, int n_shards, int gather, pilot_ot_multi **mp) {
    if (!mp || !devices) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || K <= 0 || n_shards < 1 || n_shards > 64) return fail(PILOT_OT_EINVAL, "N=%d K=%d n_shards=%d out of range", N, K, n_shards);
    if (gather < PILOT_OT_GATHER_AUTO || gather > PILOT_OT_GATHER_COPY) return fail(PILOT_OT_EINVAL, "unknown gather mode %d", gather);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    bool distinct = true;
    for (int s = 0; s < n_shards; ++s) {
        if (devices[s] < 0 || devices[s] >= n_dev) return fail(PILOT_OT_EINVAL, "device %d not visible (%d devices)", devices[s], n_dev);
        for (int t = 0; t < s; ++t) distinct = distinct && devices[t] != devices[s];
    }
    if (gather == PILOT_OT_GATHER_AUTO) gather = distinct ? PILOT_OT_GATHER_RCCL : PILOT_OT_GATHER_COPY;
    if (gather == PILOT_OT_GATHER_RCCL && !distinct)
        return fail(PILOT_OT_EINVAL, "RCCL needs one distinct device per shard (use PILOT_OT_GATHER_COPY for logical shards)");
    pilot_ot_multi *m = new (std::nothrow) pilot_ot_multi();
    if (!m) return fail(PILOT_OT_EINVAL, "out of host memory");
    m->N = N; m->K = K; m->G = n_shards; m->n_pad = (N + n_shards - 1) / n_shards; m->gather = gather;
    m->sh.resize(n_shards);
    DeviceGuard guard;
    const size_t n_loc = (size_t)m->n_pad * N;
    int rc = PILOT_OT_OK;
    for (int s = 0; s < n_shards && rc == PILOT_OT_OK; ++s) {
        Shard &h = m->sh[s];
        h.device = devices[s];
        h.n_local = (N - s + n_shards - 1) / n_shards;
        hipError_t e = hipSetDevice(h.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&h.dP, sizeof(double) * (size_t)N * K);
        if (e == hipSuccess) e = hipMalloc(&h.dM, sizeof(double) * (size_t)K * K);
        if (e == hipSuccess) e = hipMalloc(&h.dLocal, sizeof(double) * n_loc);
        if (e == hipSuccess) e = hipMalloc(&h.dErr, sizeof(double) * n_loc);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h.dIt), sizeof(int) * n_loc);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h.dFl), sizeof(int) * n_loc);
        if (e == hipSuccess) e = hipMemset(h.dLocal, 0, sizeof(double) * n_loc);     // padding rows stay 0
        if (e == hipSuccess && (gather == PILOT_OT_GATHER_RCCL || s == 0)) {
            e = hipMalloc(&h.dStage, sizeof(double) * n_loc * n_shards);
            if (e == hipSuccess) e = hipMalloc(&h.dFull, sizeof(double) * (size_t)N * N);
        }
        if (e == hipSuccess) e = hipEventCreate(&h.ev_begin);
        if (e == hipSuccess) e = hipEventCreate(&h.ev_grid);
        if (e == hipSuccess) e = hipEventCreate(&h.ev_gather);
        if (e == hipSuccess) e = hipEventCreate(&h.ev_end);
        if (e != hipSuccess) { rc = fail(PILOT_OT_EHIP, "shard %d on device %d: %s", s, h.device, hipGetErrorString(e)); break; }
        rc = pilot_ot_plan_create(N, K, &h.plan);
    }
    if (rc == PILOT_OT_OK && gather == PILOT_OT_GATHER_RCCL) {
        rc = rccl_load();
        if (rc == PILOT_OT_OK) {
            std::vector<ncclComm_t> comms(n_shards, nullptr);
            ncclResult_t r;
            {
                r = g_rccl.CommInitAll(comms.data(), n_shards, devices);
                fflush(stdout);
            }
            if (r != ncclSuccess) rc = fail(PILOT_OT_ERCCL, "ncclCommInitAll over %d devices: %s", n_shards, g_rccl.GetErrorString(r));
            else for (int s = 0; s < n_shards; ++s) m->sh[s].comm = comms[s];
        }
    }
    if (rc == PILOT_OT_OK && gather == PILOT_OT_GATHER_COPY) {
        hipError_t e = hipSetDevice(m->sh[0].device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_copied, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(PILOT_OT_EHIP, "event: %s", hipGetErrorString(e));
    }
    if (rc == PILOT_OT_OK && n_shards > 1) {
        const char *serial = pilot::test_switch("PILOT_OT_MULTI_SERIAL");
        if (!(serial && *serial && *serial != '0')) m->workers.reset(new (std::nothrow) ShardWorkers(n_shards));
    }
    if (rc != PILOT_OT_OK) { multi_free(m); return rc; }
    *mp = m;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_rccl_info(pilot_ot_multi *m, int *n_ranks, int *ranks) {
    if (!m || !n_ranks || !ranks) return fail(PILOT_OT_EINVAL, "NULL pointer");
    for (int s = 0; s < m->G; ++s) {
        n_ranks[s] = 0; ranks[s] = -1;             // peer-copy gather: no communicator
        if (!m->sh[s].comm) continue;
        RCCL_TRY(g_rccl.CommCount(m->sh[s].comm, &n_ranks[s]));
        RCCL_TRY(g_rccl.CommUserRank(m->sh[s].comm, &ranks[s]));
    }
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_destroy(pilot_ot_multi *m) {
    multi_free(m);
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_set_inputs(pilot_ot_multi *m, const double *P, const double *M) {
    if (!m || !P || !M) return fail(PILOT_OT_EINVAL, "NULL pointer");
    double mx = 0.0;
    for (size_t t = 0; t < (size_t)m->K * m->K; ++t) mx = M[t] > mx ? M[t] : mx;
    m->max_cost = mx;
    DeviceGuard guard;
    for (Shard &h : m->sh) {
        // (always: an all-zero or NaN-maximum M must not leave a previous call's max_cost on the plan)
        { const int r = pilot_ot_plan_set_max_cost(h.plan, mx > 0.0 && mx < __builtin_inf() ? mx : 1.0); if (r != PILOT_OT_OK) return r; }
        HIP_TRY(hipSetDevice(h.device));
        HIP_TRY(hipMemcpyAsync(h.dP, P, sizeof(double) * (size_t)m->N * m->K, hipMemcpyHostToDevice, h.stream));
        HIP_TRY(hipMemcpyAsync(h.dM, M, sizeof(double) * (size_t)m->K * m->K, hipMemcpyHostToDevice, h.stream));
    }
    for (Shard &h : m->sh) {
        HIP_TRY(hipSetDevice(h.device));
        HIP_TRY(hipStreamSynchronize(h.stream));      // the caller's buffers are free again on return
    }
    return PILOT_OT_OK;
}

namespace {
// assemble the row blocks: full N x N on every device (RCCL) or on the device of shard 0 (peer copies)
int multi_gather(pilot_ot_multi *m, bool mirror) {
    const size_t n_loc = (size_t)m->n_pad * m->N;
    if (m->gather == PILOT_OT_GATHER_RCCL) {
        RCCL_TRY(g_rccl.GroupStart());
        for (Shard &h : m->sh) {
            HIP_TRY(hipSetDevice(h.device));
            HIP_TRY(hipEventRecord(h.ev_gather, h.stream));
        }
        for (Shard &h : m->sh) {
            ncclResult_t r = g_rccl.AllGather(h.dLocal, h.dStage, n_loc, ncclDouble, h.comm, h.stream);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(PILOT_OT_ERCCL, "ncclAllGather: %s", g_rccl.GetErrorString(r)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        for (Shard &h : m->sh) {
            HIP_TRY(hipSetDevice(h.device));
            int rc = launch_interleave(h.dStage, m->G, m->n_pad, m->N, h.dFull, h.stream);
            if (rc == PILOT_OT_OK && mirror) rc = pilot_ot_mirror_upper_dev(h.dFull, m->N, h.stream);
            if (rc != PILOT_OT_OK) return rc;
            HIP_TRY(hipEventRecord(h.ev_end, h.stream));
        }
        return PILOT_OT_OK;
    }
    Shard &h0 = m->sh[0];
    HIP_TRY(hipSetDevice(h0.device));
    for (int s = 1; s < m->G; ++s) HIP_TRY(hipStreamWaitEvent(h0.stream, m->sh[s].ev_grid, 0));
    HIP_TRY(hipEventRecord(h0.ev_gather, h0.stream));      // every shard's rows are ready from here on
    for (int s = 0; s < m->G; ++s) {
        Shard &h = m->sh[s];
        if (h.device == h0.device)
            HIP_TRY(hipMemcpyAsync(h0.dStage + s * n_loc, h.dLocal, sizeof(double) * n_loc, hipMemcpyDeviceToDevice, h0.stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(h0.dStage + s * n_loc, h0.device, h.dLocal, h.device, sizeof(double) * n_loc, h0.stream));
    }
    // the shards' row blocks have been read from here on: the next call's kernels may overwrite them (multi_begin waits)
    HIP_TRY(hipEventRecord(m->ev_copied, h0.stream));
    m->copied_valid = true;
    int rc = launch_interleave(h0.dStage, m->G, m->n_pad, m->N, h0.dFull, h0.stream);
    if (rc == PILOT_OT_OK && mirror) rc = pilot_ot_mirror_upper_dev(h0.dFull, m->N, h0.stream);
    if (rc != PILOT_OT_OK) return rc;
    HIP_TRY(hipEventRecord(h0.ev_end, h0.stream));
    return PILOT_OT_OK;
}

// start of a shard's part of a call (on the shard's thread): its stream must not overwrite dLocal while shard 0's peer
// copies of the PREVIOUS call are still reading it (back-to-back asynchronous calls, peer-copy gather)
int shard_begin(pilot_ot_multi *m, Shard &h) {
    HIP_TRY(hipSetDevice(h.device));
    if (m->gather == PILOT_OT_GATHER_COPY && m->copied_valid && &h != &m->sh[0]) HIP_TRY(hipStreamWaitEvent(h.stream, m->ev_copied, 0));
    HIP_TRY(hipEventRecord(h.ev_begin, h.stream));
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_multi_sinkhorn(pilot_ot_multi *m, double reg, int num_iter_max, double stop_thr, double tau,
                                      int check_period, int precision, double f32_floor_ulps, int cost_is_symmetric) {
    if (!m) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (!(reg > 0.0)) return fail(PILOT_OT_EINVAL, "reg=%g must be positive", reg);
    // Decide ONCE, from max(M) of the current inputs, what every shard runs -- the same decisions pilot_ot_sinkhorn_grid takes
    // on one device (the device entry point itself assumes a cost matrix normalised by its maximum, Trajectory.py:101):
    // exp(-M/reg) outside the f64 range -> POT-literal kernel, whatever precision was asked for.
    precision = pilot_ot_resolve_precision(precision, m->max_cost / reg, m->K, cost_is_symmetric, tau);
    DeviceGuard guard;
    int rc = m->for_each_shard([&](int s) -> int {
        Shard &h = m->sh[s];
        int r = shard_begin(m, h);
        if (r != PILOT_OT_OK) return r;
        r = pilot_ot_sinkhorn_grid_dev(h.plan, h.dP, h.dM, reg, num_iter_max, stop_thr, tau, check_period, precision, f32_floor_ulps,
                                       cost_is_symmetric, s < m->N ? s : m->N /* more shards than rows: empty */, m->N, m->G, h.dLocal,
                                       h.dIt, h.dErr, h.dFl, h.stream);
        if (r != PILOT_OT_OK) return r;
        HIP_TRY(hipEventRecord(h.ev_grid, h.stream));
        return PILOT_OT_OK;
    });
    if (rc != PILOT_OT_OK) return rc;
    m->ran = true; m->exact = false;
    return multi_gather(m, false);
}

PILOT_API int pilot_ot_multi_emd(pilot_ot_multi *m, int cost_is_symmetric) {
    if (!m) return fail(PILOT_OT_EINVAL, "NULL pointer");
    DeviceGuard guard;
    const size_t n_loc = (size_t)m->n_pad * m->N;
    int rc = m->for_each_shard([&](int s) -> int {
        Shard &h = m->sh[s];
        int r = shard_begin(m, h);
        if (r != PILOT_OT_OK) return r;
        HIP_TRY(hipMemsetAsync(h.dLocal, 0, sizeof(double) * n_loc, h.stream));
        HIP_TRY(hipMemsetAsync(h.dIt, 0, sizeof(int) * n_loc, h.stream));
        r = pilot_ot_emd_grid_dev(h.plan, h.dP, h.dM, cost_is_symmetric ? PILOT_OT_EMD_UPPER : PILOT_OT_EMD_ALL, s < m->N ? s : m->N, m->N,
                                  m->G, h.dLocal, h.dIt, h.stream);
        if (r != PILOT_OT_OK) return r;
        HIP_TRY(hipEventRecord(h.ev_grid, h.stream));
        return PILOT_OT_OK;
    });
    if (rc != PILOT_OT_OK) return rc;
    m->ran = true; m->exact = true;
    return multi_gather(m, cost_is_symmetric != 0);
}

PILOT_API int pilot_ot_multi_sync(pilot_ot_multi *m) {
    if (!m) return fail(PILOT_OT_EINVAL, "NULL pointer");
    DeviceGuard guard;
    for (Shard &h : m->sh) {
        HIP_TRY(hipSetDevice(h.device));
        HIP_TRY(hipStreamSynchronize(h.stream));
    }
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_fetch(pilot_ot_multi *m, double *emd, int *iters, double *err, int *flags) {
    if (!m || !emd) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (!m->ran) return fail(PILOT_OT_EINVAL, "nothing has been computed on this multi-GPU plan yet");
    int rc = pilot_ot_multi_sync(m);
    if (rc != PILOT_OT_OK) return rc;
    DeviceGuard guard;
    const int N = m->N, G = m->G;
    HIP_TRY(hipSetDevice(m->sh[0].device));
    HIP_TRY(hipMemcpy(emd, m->sh[0].dFull, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost));
    if (!iters && !err && !flags) return PILOT_OT_OK;
    // per-pair diagnostics are not part of the assembled matrix: fetched shard by shard, interleaved on the host
    const size_t n_loc = (size_t)m->n_pad * N;
    std::vector<int> ti(iters || flags ? n_loc : 0);
    std::vector<double> te(err ? n_loc : 0);
    for (int s = 0; s < G; ++s) {
        Shard &h = m->sh[s];
        HIP_TRY(hipSetDevice(h.device));
        const size_t n_got = (size_t)h.n_local * N;
        auto scatter_i = [&](int *dst) { for (int t = 0; t < h.n_local; ++t) memcpy(dst + (size_t)(s + t * G) * N, ti.data() + (size_t)t * N, sizeof(int) * N); };
        if (iters) { HIP_TRY(hipMemcpy(ti.data(), h.dIt, sizeof(int) * n_got, hipMemcpyDeviceToHost)); scatter_i(iters); }
        if (flags && !m->exact) { HIP_TRY(hipMemcpy(ti.data(), h.dFl, sizeof(int) * n_got, hipMemcpyDeviceToHost)); scatter_i(flags); }
        if (err && !m->exact) {
            HIP_TRY(hipMemcpy(te.data(), h.dErr, sizeof(double) * n_got, hipMemcpyDeviceToHost));
            for (int t = 0; t < h.n_local; ++t) memcpy(err + (size_t)(s + t * G) * N, te.data() + (size_t)t * N, sizeof(double) * N);
        }
    }
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_device_matrix(pilot_ot_multi *m, int shard, double **d_full) {
    if (!m || !d_full || shard < 0 || shard >= m->G) return fail(PILOT_OT_EINVAL, "bad argument");
    if (!m->sh[shard].dFull) return fail(PILOT_OT_EINVAL, "with PILOT_OT_GATHER_COPY only shard 0 holds the assembled matrix");
    *d_full = m->sh[shard].dFull;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_multi_times(pilot_ot_multi *m, float *grid_ms, float *gather_ms) {
    if (!m || !grid_ms || !gather_ms) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (!m->ran) return fail(PILOT_OT_EINVAL, "nothing has been computed on this multi-GPU plan yet");
    int rc = pilot_ot_multi_sync(m);
    if (rc != PILOT_OT_OK) return rc;
    DeviceGuard guard;
    for (int s = 0; s < m->G; ++s) {
        HIP_TRY(hipSetDevice(m->sh[s].device));
        HIP_TRY(hipEventElapsedTime(&grid_ms[s], m->sh[s].ev_begin, m->sh[s].ev_grid));
    }
    // gather = all-gather (or peer copies) + row interleave (+ mirror).  RCCL: a shard's collective also waits for its
    // slower peers, so the collective's own cost is the SMALLEST per-shard (gather start -> matrix assembled) time; peer
    // copies: shard 0's stream, measured from the point where every shard's rows are ready.
    float best = -1.f;
    const int n_meas = m->gather == PILOT_OT_GATHER_RCCL ? m->G : 1;
    for (int s = 0; s < n_meas; ++s) {
        HIP_TRY(hipSetDevice(m->sh[s].device));
        float g = 0.f;
        HIP_TRY(hipEventElapsedTime(&g, m->sh[s].ev_gather, m->sh[s].ev_end));
        if (best < 0.f || g < best) best = g;
    }
    *gather_ms = best;
    return PILOT_OT_OK;
}

// ================================================================================================
// host-buffer convenience: numpy in, numpy out, over several devices (per-thread cached context)
namespace {
// One cached context per calling thread, kept in a process-wide registry: pilot_ot_shutdown() releases ALL of them (it must
// not run concurrently with other calls), and the context of a thread that has exited is released by the next
// host-buffer call of any thread -- never from a thread-exit or process-exit hook, where the HIP runtime may be gone.
struct MultiCtx {
    pilot_ot_multi *m = nullptr;
    std::vector<int> devices;
    int gather = 0;
    bool orphan = false;
};
std::mutex g_ctx_mutex;
std::vector<MultiCtx *> g_ctx_all;
struct CtxOwner {      // thread-exit hook: only marks, no HIP calls
    MultiCtx *c = nullptr;
    ~CtxOwner() {
        if (!c) return;
        std::lock_guard<std::mutex> l(g_ctx_mutex);
        c->orphan = true;
    }
};
thread_local CtxOwner g_owner;

int multi_ctx(int N, int K, const int *devices, int n_devices, int gather, pilot_ot_multi **out) {
    if (!g_owner.c) {
        std::lock_guard<std::mutex> l(g_ctx_mutex);
        for (size_t i = 0; i < g_ctx_all.size();) {          // contexts of threads that are gone
            if (g_ctx_all[i]->orphan) {
                if (g_ctx_all[i]->m) multi_free(g_ctx_all[i]->m);
                delete g_ctx_all[i];
                g_ctx_all.erase(g_ctx_all.begin() + (long)i);
            } else {
                ++i;
            }
        }
        g_owner.c = new MultiCtx();
        g_ctx_all.push_back(g_owner.c);
    }
    MultiCtx &c = *g_owner.c;
    const bool same = c.m && c.m->N == N && c.m->K == K && c.gather == gather && (int)c.devices.size() == n_devices &&
                      memcmp(c.devices.data(), devices, sizeof(int) * n_devices) == 0;
    if (!same) {
        if (c.m) multi_free(c.m);
        c.m = nullptr;
        int rc = pilot_ot_multi_create(N, K, devices, n_devices, gather, &c.m);
        if (rc != PILOT_OT_OK) return rc;
        c.devices.assign(devices, devices + n_devices);
        c.gather = gather;
    }
    *out = c.m;
    return PILOT_OT_OK;
}
}  // namespace

namespace pilot {
void abi_multi_release() {
    std::lock_guard<std::mutex> l(g_ctx_mutex);
    for (MultiCtx *c : g_ctx_all) {
        if (c->m) multi_free(c->m);
        c->m = nullptr;
        c->devices.clear();
    }
    // (the records themselves stay: their owner threads still point at them)
}
}  // namespace pilot

PILOT_API int pilot_ot_sinkhorn_grid_multi(const double *P, int N, int K, const double *M, double reg, int num_iter_max,
                                           double stop_thr, double tau, int check_period, int precision,
                                           double f32_floor_ulps, int cost_is_symmetric, const int *devices,
                                           int n_devices, int gather, double *emd, int *iters, double *err, int *flags) {
    if (!P || !M || !emd || !devices) return fail(PILOT_OT_EINVAL, "NULL pointer");
    // (precision is resolved once for all shards by pilot_ot_multi_sinkhorn, from max(M) of these inputs)
    pilot_ot_multi *m = nullptr;
    int rc = multi_ctx(N, K, devices, n_devices, gather, &m);
    if (rc == PILOT_OT_OK) rc = pilot_ot_multi_set_inputs(m, P, M);
    if (rc == PILOT_OT_OK)
        rc = pilot_ot_multi_sinkhorn(m, reg, num_iter_max, stop_thr, tau, check_period, precision, f32_floor_ulps,
                                     cost_is_symmetric);
    if (rc == PILOT_OT_OK) rc = pilot_ot_multi_fetch(m, emd, iters, err, flags);
    return rc;
}

PILOT_API int pilot_ot_emd_grid_multi(const double *P, int N, int K, const double *M, int cost_is_symmetric,
                                      const int *devices, int n_devices, int gather, double *emd, int *n_aug) {
    if (!P || !M || !emd || !devices) return fail(PILOT_OT_EINVAL, "NULL pointer");
    pilot_ot_multi *m = nullptr;
    int rc = multi_ctx(N, K, devices, n_devices, gather, &m);
    if (rc == PILOT_OT_OK) rc = pilot_ot_multi_set_inputs(m, P, M);
    if (rc == PILOT_OT_OK) rc = pilot_ot_multi_emd(m, cost_is_symmetric);
    if (rc == PILOT_OT_OK) rc = pilot_ot_multi_fetch(m, emd, n_aug, nullptr, nullptr);
    return rc;
}

// ================================================================================================
// cell-level W2 (extension): full N x N grid, rows dealt round-robin over several devices.  Every device
// holds the whole cohort and solves its rows on its own stream (one host thread per device); the matrix is assembled ON THE
// DEVICES like the proportion path's: ONE all-gather of the padded row blocks over RCCL (distinct devices) or peer copies
// into the first device (repeated ids = logical shards), then the row interleave.  Per-pair diagnostics (iters, err) are
// not part of the matrix and are collected shard by shard.
PILOT_API int pilot_ot_cell_w2_grid_multi(const float *X, const long long *offsets, int N, int D, double scale, double reg,
                                          int num_iter_max, double stop_thr, int check_period, double f32_floor_ulps,
                                          const int *devices, int n_devices, double *w2, int *iters, double *err) {
    if (!X || !offsets || !w2 || !devices) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (n_devices < 1 || n_devices > 64) return fail(PILOT_OT_EINVAL, "n_devices=%d out of range", n_devices);
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    bool distinct = true;
    for (int s = 0; s < n_devices; ++s) {
        if (devices[s] < 0 || devices[s] >= n_dev) return fail(PILOT_OT_EINVAL, "device %d not visible (%d devices)", devices[s], n_dev);
        for (int t = 0; t < s; ++t) distinct = distinct && devices[t] != devices[s];
    }
    const bool rccl = distinct && n_devices > 1;
    if (rccl) { const int rc = rccl_load(); if (rc != PILOT_OT_OK) return rc; }
    const int G = n_devices, n_pad = (N + G - 1) / G;
    const size_t n_loc = (size_t)n_pad * N;
    struct CS {
        pilot_ot_cell_cohort *co = nullptr;
        double *dLocal = nullptr, *dStage = nullptr, *dFull = nullptr, *dW = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t ev = nullptr;
        ncclComm_t comm = nullptr;
        size_t n_out = 0;
        int rc = PILOT_OT_OK;
        std::string msg;
    };
    std::vector<CS> cs(G);
    DeviceGuard guard;
    // per device, concurrently: cohort (H2D of all cells), row shard enqueued, rows copied into the padded block
    auto shard_job = [&](int s) {
        CS &c = cs[s];
        auto body = [&]() -> int {
            HIP_TRY(hipSetDevice(devices[s]));
            int rc = pilot_ot_cell_cohort_create(X, offsets, N, D, &c.co);
            if (rc != PILOT_OT_OK) return rc;
            rc = pilot::cell_enqueue_rows(c.co, scale, reg, num_iter_max, stop_thr, check_period, f32_floor_ulps, s < N ? s : N, N, G, &c.n_out);
            if (rc != PILOT_OT_OK) return rc;
            pilot::cell_buffers(c.co, &c.dW, &c.stream);
            HIP_TRY(hipMalloc(&c.dLocal, sizeof(double) * n_loc));
            if (rccl || s == 0) {
                HIP_TRY(hipMalloc(&c.dStage, sizeof(double) * n_loc * G));
                HIP_TRY(hipMalloc(&c.dFull, sizeof(double) * (size_t)N * N));
            }
            HIP_TRY(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
            HIP_TRY(hipMemsetAsync(c.dLocal, 0, sizeof(double) * n_loc, c.stream));
            if (c.n_out) HIP_TRY(hipMemcpyAsync(c.dLocal, c.dW, sizeof(double) * c.n_out, hipMemcpyDeviceToDevice, c.stream));
            HIP_TRY(hipEventRecord(c.ev, c.stream));
            return PILOT_OT_OK;
        };
        c.rc = body();
        if (c.rc != PILOT_OT_OK) c.msg = pilot_ot_last_error();
    };
    {
        std::vector<std::thread> th;
        for (int s = 1; s < G; ++s) th.emplace_back(shard_job, s);
        shard_job(0);
        for (auto &t : th) t.join();
    }
    int rc = PILOT_OT_OK;
    for (int s = 0; s < G && rc == PILOT_OT_OK; ++s)
        if (cs[s].rc != PILOT_OT_OK) rc = fail(cs[s].rc, "%s", cs[s].msg.c_str());
    auto gather = [&]() -> int {
        if (rccl) {
            std::vector<ncclComm_t> comms(G, nullptr);
            ncclResult_t r;
            {
                r = g_rccl.CommInitAll(comms.data(), G, devices);
            }
            if (r != ncclSuccess) return fail(PILOT_OT_ERCCL, "ncclCommInitAll over %d devices: %s", G, g_rccl.GetErrorString(r));
            for (int s = 0; s < G; ++s) cs[s].comm = comms[s];
            RCCL_TRY(g_rccl.GroupStart());
            for (int s = 0; s < G; ++s) {
                ncclResult_t a = g_rccl.AllGather(cs[s].dLocal, cs[s].dStage, n_loc, ncclDouble, cs[s].comm, cs[s].stream);
                if (a != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(PILOT_OT_ERCCL, "ncclAllGather: %s", g_rccl.GetErrorString(a)); }
            }
            RCCL_TRY(g_rccl.GroupEnd());
            for (int s = 0; s < G; ++s) {
                HIP_TRY(hipSetDevice(devices[s]));
                const int ir = launch_interleave(cs[s].dStage, G, n_pad, N, cs[s].dFull, cs[s].stream);
                if (ir != PILOT_OT_OK) return ir;
            }
        } else {
            HIP_TRY(hipSetDevice(devices[0]));
            for (int s = 1; s < G; ++s) HIP_TRY(hipStreamWaitEvent(cs[0].stream, cs[s].ev, 0));
            for (int s = 0; s < G; ++s) {
                if (devices[s] == devices[0])
                    HIP_TRY(hipMemcpyAsync(cs[0].dStage + s * n_loc, cs[s].dLocal, sizeof(double) * n_loc, hipMemcpyDeviceToDevice, cs[0].stream));
                else
                    HIP_TRY(hipMemcpyPeerAsync(cs[0].dStage + s * n_loc, devices[0], cs[s].dLocal, devices[s], sizeof(double) * n_loc, cs[0].stream));
            }
            const int ir = launch_interleave(cs[0].dStage, G, n_pad, N, cs[0].dFull, cs[0].stream);
            if (ir != PILOT_OT_OK) return ir;
        }
        for (int s = 0; s < G; ++s) {
            HIP_TRY(hipSetDevice(devices[s]));
            HIP_TRY(hipStreamSynchronize(cs[s].stream));
        }
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipMemcpy(w2, cs[0].dFull, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost));
        if (iters || err) {
            std::vector<int> ti;
            std::vector<double> te;
            for (int s = 0; s < G; ++s) {
                HIP_TRY(hipSetDevice(devices[s]));
                ti.resize(cs[s].n_out); te.resize(cs[s].n_out);
                const int cr = pilot::cell_collect(cs[s].co, cs[s].n_out, nullptr, iters ? ti.data() : nullptr, err ? te.data() : nullptr, nullptr);
                if (cr != PILOT_OT_OK) return cr;
                for (size_t t = 0; t < cs[s].n_out / (size_t)N; ++t) {
                    const size_t row = (size_t)s + t * G;
                    if (iters) memcpy(iters + row * N, ti.data() + t * N, sizeof(int) * N);
                    if (err) memcpy(err + row * N, te.data() + t * N, sizeof(double) * N);
                }
            }
        }
        return PILOT_OT_OK;
    };
    if (rc == PILOT_OT_OK) rc = gather();
    std::string keep = rc != PILOT_OT_OK ? pilot_ot_last_error() : "";
    for (int s = 0; s < G; ++s) {
        CS &c = cs[s];
        (void)hipSetDevice(devices[s]);
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        if (c.comm) (void)g_rccl.CommDestroy(c.comm);
        for (void *p : {(void *)c.dLocal, (void *)c.dStage, (void *)c.dFull}) if (p) (void)hipFree(p);
        if (c.ev) (void)hipEventDestroy(c.ev);
        if (c.co) pilot_ot_cell_cohort_destroy(c.co);
    }
    if (rc != PILOT_OT_OK) return fail(rc, "%s", keep.c_str());
    return PILOT_OT_OK;
}
