// libpilot_ot.so -- C ABI over the gfx950 kernels (declared in include/pilot_ot.h).
// Host side of the drop-in boundary for pilotpy/tools/Trajectory.py:441-523.
// There is deliberately no CPU implementation in this file: without a HIP device every compute
// entry point returns PILOT_OT_EHIP.
#include <hip/hip_runtime.h>
#include <pthread.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <new>
#include <vector>

#include "abi_common.hpp"
#include "sinkhorn_launch.hpp"
#include "emd_kernels.hpp"
#include "emd_multi_kernels.hpp"
#include "prepass_kernels.hpp"
#include "cellw2_kernels.hpp"
#include "generic_kernels.hpp"
#include "emd_generic_kernel.hpp"

namespace {
thread_local char g_err[512] = "";
}

namespace pilot {
int abi_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace pilot

namespace {

#define fail(...) pilot::abi_fail(__VA_ARGS__)

// ------------------------------------------------------------------------------------------------
// K1: centroid cost matrix (scipy pdist + squareform, Trajectory.py:468-469).  K <= a few hundred,
// D <= a few hundred: one workgroup, one thread per unordered pair, fp64 like scipy.  Far below the
// size where an MFMA contraction pays (K*K*D = 75k FMAs at c3).
// aux: metric-specific extra input (mahalanobis: the D x D inverse covariance VI, computed by the host like scipy does)
__global__ void cost_matrix_kernel(const double *__restrict__ X, int K, int D, int metric, const double *__restrict__ aux,
                                   double *__restrict__ C) {
    extern __shared__ double stat[];  // per-row norm (cosine) or mean + centred norm (correlation); per-dimension variance (seuclidean)
    double *nrm = stat, *mean = stat + K, *var = stat + 2 * K;
    if (metric == PILOT_OT_METRIC_SEUCLIDEAN)       // scipy: V = np.var(X, axis=0, ddof=1)
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            double m = 0.0;
            for (int i = 0; i < K; ++i) m += X[(size_t)i * D + d];
            m /= K;
            double s = 0.0;
            for (int i = 0; i < K; ++i) { const double t = X[(size_t)i * D + d] - m; s += t * t; }
            var[d] = s / (K - 1);
        }
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const double *x = X + (size_t)i * D;
        double m = 0.0;
        if (metric == PILOT_OT_METRIC_CORRELATION) {
            for (int d = 0; d < D; ++d) m += x[d];
            m /= D;
        }
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += (x[d] - m) * (x[d] - m);
        mean[i] = m;
        nrm[i] = sqrt(s);
        C[(size_t)i * K + i] = 0.0;
    }
    __syncthreads();
    const int npairs = K * (K - 1) / 2;
    for (int pidx = threadIdx.x; pidx < npairs; pidx += blockDim.x) {
        // unrank (i < j) from the condensed pdist index
        int i = 0, rem = pidx;
        while (rem >= K - 1 - i) { rem -= K - 1 - i; ++i; }
        const int j = i + 1 + rem;
        const double *u = X + (size_t)i * D, *v = X + (size_t)j * D;
        double out = 0.0;
        switch (metric) {
        case PILOT_OT_METRIC_COSINE:
        case PILOT_OT_METRIC_CORRELATION: {
            const double mu = mean[i], mv = mean[j];
            double dot = 0.0;
            for (int d = 0; d < D; ++d) dot += (u[d] - mu) * (v[d] - mv);
            double c = dot / (nrm[i] * nrm[j]);
            if (fabs(c) > 1.0) c = copysign(1.0, c);  // scipy clips rounding overshoot
            out = 1.0 - c;
            break;
        }
        case PILOT_OT_METRIC_EUCLIDEAN:
        case PILOT_OT_METRIC_MINKOWSKI:          // scipy's default p = 2 (the reference forwards only the name)
        case PILOT_OT_METRIC_SQEUCLIDEAN: {
            double s = 0.0;
            for (int d = 0; d < D; ++d) { const double t = u[d] - v[d]; s += t * t; }
            out = metric == PILOT_OT_METRIC_SQEUCLIDEAN ? s : sqrt(s);
            break;
        }
        case PILOT_OT_METRIC_SEUCLIDEAN: {
            double s = 0.0;
            for (int d = 0; d < D; ++d) { const double t = u[d] - v[d]; s += t * t / var[d]; }
            out = sqrt(s);
            break;
        }
        case PILOT_OT_METRIC_BRAYCURTIS: {
            double s1 = 0.0, s2 = 0.0;
            for (int d = 0; d < D; ++d) { s1 += fabs(u[d] - v[d]); s2 += fabs(u[d] + v[d]); }
            out = s1 / s2;
            break;
        }
        case PILOT_OT_METRIC_CANBERRA: {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                const double den = fabs(u[d]) + fabs(v[d]);
                if (den > 0.0) s += fabs(u[d] - v[d]) / den;          // 0/0 terms count as 0
            }
            out = s;
            break;
        }
        case PILOT_OT_METRIC_HAMMING: {
            int ne = 0;
            for (int d = 0; d < D; ++d) ne += u[d] != v[d];
            out = double(ne) / D;
            break;
        }
        case PILOT_OT_METRIC_CITYBLOCK: {
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += fabs(u[d] - v[d]);
            out = s;
            break;
        }
        // scipy's "boolean" dissimilarities: pdist converts the rows to bool (non-zero = True) and counts agreements
        case PILOT_OT_METRIC_JACCARD: case PILOT_OT_METRIC_YULE: case PILOT_OT_METRIC_RUSSELLRAO: case PILOT_OT_METRIC_SOKALSNEATH:
        case PILOT_OT_METRIC_ROGERSTANIMOTO: case PILOT_OT_METRIC_SOKALMICHENER: case PILOT_OT_METRIC_KULCZYNSKI1: {
            double ntt = 0, ntf = 0, nft = 0, nff = 0;
            for (int d = 0; d < D; ++d) {
                const bool a = u[d] != 0.0, b = v[d] != 0.0;
                ntt += a && b; ntf += a && !b; nft += !a && b; nff += !a && !b;
            }
            const double R = ntf + nft;
            if (metric == PILOT_OT_METRIC_JACCARD) out = (ntt + R) > 0.0 ? R / (ntt + R) : 0.0;
            else if (metric == PILOT_OT_METRIC_YULE) { const double h = ntf * nft; out = h == 0.0 ? 0.0 : 2.0 * h / (ntt * nff + h); }
            else if (metric == PILOT_OT_METRIC_RUSSELLRAO) out = (double(D) - ntt) / double(D);
            else if (metric == PILOT_OT_METRIC_SOKALSNEATH) out = 2.0 * R / (ntt + 2.0 * R);
            else if (metric == PILOT_OT_METRIC_KULCZYNSKI1) out = ntt / R;
            else out = 2.0 * R / (ntt + nff + 2.0 * R);           // rogerstanimoto == sokalmichener
            break;
        }
        case PILOT_OT_METRIC_DICE: {            // (scipy evaluates this one on the values: ntt = sum u v, ...)
            double ntt = 0.0, nd = 0.0;
            for (int d = 0; d < D; ++d) { ntt += u[d] * v[d]; nd += u[d] * (1.0 - v[d]) + (1.0 - u[d]) * v[d]; }
            out = nd / (2.0 * ntt + nd);
            break;
        }
        case PILOT_OT_METRIC_JENSENSHANNON: {
            // no fused multiply-adds in this block: scipy's build (x86-64) rounds every product, and with proportional rows the sign of a
            // sum of +-1e-16 terms -- NaN or not -- follows those roundings (tools/ubench/rcp_f64.hip: m = (p + q) / 2 with p fused in)
#pragma clang fp contract(off)
            double su = 0.0, sv = 0.0;
            bool neg = false;
            for (int d = 0; d < D; ++d) { neg = neg || u[d] < 0.0 || v[d] < 0.0; su += u[d]; sv += v[d]; }
            if (neg || su == 0.0 || sv == 0.0) { out = HUGE_VAL; break; }     // (scipy: inf for a negative entry or an all-zero row)
            // (scipy's build multiplies by the reciprocals of the sums; dividing instead moves a Jensen-Shannon value near zero --
            // proportional rows -- by up to 1e-8 and turns scipy's NaN, the root of a sum that rounded below zero, into 0:
            // tools/fuzz_prepass.py, 20 000 pairs against scipy 1.15.3: 0 differences this way, 3 379 NaN mismatches the other)
            const double ru = 1.0 / su, rv = 1.0 / sv;
            double js = 0.0;
            for (int d = 0; d < D; ++d) {
                const double p = u[d] * ru, q = v[d] * rv, m = (p + q) / 2.0;
                if (p > 0.0) js += p * log(p / m);
                if (q > 0.0) js += q * log(q / m);
            }
            out = sqrt(js / 2.0);
            break;
        }
        case PILOT_OT_METRIC_MAHALANOBIS: {     // sqrt((u - v) VI (u - v)^T)
            double s = 0.0;
            for (int a = 0; a < D; ++a) {
                double t = 0.0;
                for (int b = 0; b < D; ++b) t += (u[b] - v[b]) * aux[(size_t)b * D + a];
                s += t * (u[a] - v[a]);
            }
            out = sqrt(s);
            break;
        }
        default: {  // chebyshev
            double s = 0.0;
            for (int d = 0; d < D; ++d) { const double t = fabs(u[d] - v[d]); s = t > s ? t : s; }
            out = s;
        }
        }
        C[(size_t)i * K + j] = out;
        C[(size_t)j * K + i] = out;
    }
}

constexpr int MAX_K = 128;          // the MFMA pair-grid kernels (8 row-tiles of 16 cell types)
constexpr int GENERIC_MAX_K = 2048;  // the reference-semantics fallback kernel (vectors in LDS)
constexpr int EMD_MAX_K = 256;       // exact-OT kernel: 4 rows / columns per lane
constexpr int WIDE_MAX_K = 256;      // sinkhorn_wide_kernel: 128 < K <= 256, eight waves per 16-pair tile
// (Round 4 had an experiment switch that ran the eight-waves-per-tile kernel below K = 128; it under-sized p_slot / img for the
// wide layout and the experiment is done -- profiles/r04/ab_experiments.md #7 -- so the switch is gone.)
constexpr int CTRL_INTS = pilot::CTRL_INTS;      // control block of a call: see pilot_ot_plan::track_count
// ... followed by the two order histograms and, from a 128-byte boundary, the ticket counters of the fast launch's work queue
constexpr int CTRL_SHARDS_AT = (CTRL_INTS + 2 * pilot::ORDER_NB + 31) / 32 * 32;
constexpr int CTRL_SHARDS_TRACK_AT = CTRL_SHARDS_AT + pilot::QUEUE_SHARDS * pilot::QUEUE_SHARD_STRIDE;      // (the tracking launch's)
constexpr int CTRL_BLOCK_INTS = CTRL_SHARDS_TRACK_AT + pilot::QUEUE_SHARDS * pilot::QUEUE_SHARD_STRIDE;
constexpr int TIMING_RING = 64;

constexpr size_t LDS_BYTES = 160 * 1024;
// exact-EMD kernel: workgroups of pilot::emd_waves(NK) waves, M (+ row minima) in LDS; resident workgroups per CU
static int emd_nk(int K) { return K <= 64 ? 1 : (K <= 128 ? 2 : (K <= 192 ? 3 : 4)); }
// K <= 16: four pairs per wavefront (emd_multi_kernels.hpp).  PILOT_OT_EMD_MULTI=0: the one-pair-per-wave kernel instead (A/B and the
// parity test between the two); =1 / =2 force the flow values into LDS / the global slab (default: LDS up to K = 15, where five or six
// waves per SIMD still fit beside them; profiles/r05/emd_multi_probe.txt)
constexpr int EMD_MULTI_MAX_K = 16;
static int emd_multi_mode(int K) {
    const char *e = pilot::test_switch("PILOT_OT_EMD_MULTI");
    if (e && *e) return atoi(e);
    return K <= 15 ? 1 : 2;
}
static int emd_wgs_per_cu(int K) {
    if (K > 128) return 1;                      // cost matrix in global memory, 3-4 rows per lane: one workgroup per CU
    const size_t lds = pilot::emd_lds_bytes(K);     // (M, row minima; K <= 64: + the per-column source order and its inverse)
    int by_lds = (int)(LDS_BYTES / lds);
    const int by_regs = K <= 64 ? 4 : 2;        // <= 64 VGPRs per lane: 4 x 8 or 2 x 16 waves = 8 per SIMD
    if (by_lds > by_regs) by_lds = by_regs;
    return by_lds < 1 ? 1 : by_lds;
}

}  // namespace

struct pilot_ot_plan {
    int N, K, device;
    double max_cost;   // max(M) of the cost the caller keeps on the device (pilot_ot_plan_set_max_cost; 1 = Trajectory.py:101's
                       // normalised cost): every range decision of a device-resident call is taken on max_cost / reg
    void *img;         // 3 operand images, sized for f64 at this K
    void *p_slot;      // N x KP proportions in accumulator-slot order (f32 or f64; sized for f64)
    int *track_list;   // N x N
    int *track_count;  // [0] track-list length, [1] queue head of the fast launch, [2] queue head of the tracking launch,
                       // [3] queue head of the solo waves, [4..7] split of the ordered list: n_top, (unused copy), n_dup,
                       // n_dup = number of leading exact-duplicate pairs, [8] length of the f64 fallback list (mixed precision
                       // at small reg), [9] its queue head, [10] length of the NaN list (pairs re-solved by the POT-literal
                       // kernel), [11] its queue head
    int *order_list;   // N x N: longest-first work order of the fast launch
    unsigned char *order_bucket;  // N x N
    int *order_hist;   // 2 * ORDER_NB: histogram + scatter cursors
    int *flags_ws;     // per-pair flags when the caller passes none
    size_t flags_ws_n;
    int *emd_counter;  // 1: dynamic pair queue of the exact-EMD kernel
    double *f_slab;    // exact-EMD flow values: one K*K block per resident wave (per 16-lane group of a wave for K <= 16); allocated by the
    size_t f_slab_bytes;  // first exact-mode call that needs it -- a Sinkhorn-only plan never pays for it (164 MB at K = 50)
    double *emdg_slab; // K > 256: flow + label slab per resident workgroup of emd_generic_kernel, then K row minima (lazy)
    int emdg_wgs;
    double *kws;       // generic Sinkhorn kernel: K' and its transpose per workgroup (allocated on first use)
    int generic_wgs;
    float *wide_rec;   // 128 < K <= 256: one record per pair of the grid for sinkhorn_wide_kernel (allocated on first use)
    size_t wide_rec_n; //   pairs it holds
    int *nan_list;     // pairs that ended in NaN (grown on demand)
    size_t nan_list_n;
    int n_cu;
    // event ring for per-launch kernel timing (bench.py roofline)
    int timing;                       // 0 off
    long n_timed;                     // calls recorded so far
    long n_calls;                     // calls seen while timing is on (every `timing`-th one is recorded)
    hipEvent_t ev[TIMING_RING][4];    // [slot]{main begin, main end, track begin, track end}
    // hipGraph replay of a repeated Sinkhorn call (pilot_ot_plan_enable_graph): the launch sequence of one call captured
    // on `gstream` and replayed on the caller's stream while the arguments stay the same
    int graph_mode;                   // 0 off
    int gkey_seen;                    // the key below was used by an ordinary (uncaptured) call: buffers are grown
    struct GraphKey {
        const void *P, *M, *emd, *iters, *err, *flags;
        double reg, stop_thr, tau, floor_ulps, max_cost;
        int num_iter_max, check_period, cfg, mixed, sym, row_begin, n_rows, row_step, debug;
        bool operator==(const GraphKey &o) const {
            return P == o.P && M == o.M && emd == o.emd && iters == o.iters && err == o.err && flags == o.flags && reg == o.reg &&
                   stop_thr == o.stop_thr && tau == o.tau && floor_ulps == o.floor_ulps && max_cost == o.max_cost && num_iter_max == o.num_iter_max &&
                   check_period == o.check_period && cfg == o.cfg && mixed == o.mixed && sym == o.sym && row_begin == o.row_begin &&
                   n_rows == o.n_rows && row_step == o.row_step && debug == o.debug;
        }
    } gkey;
    hipStream_t gstream;
    hipGraphExec_t gexec;
};

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// test switches: a process-wide table set through pilot_ot_test_switch (never from the environment)
namespace {
std::mutex g_switch_mutex;
std::map<std::string, std::string> g_switches;
}  // namespace
namespace pilot {
const char *test_switch(const char *name) {
    static thread_local std::string value;                                 // (the caller's copy: another thread may set the switch meanwhile)
    std::lock_guard<std::mutex> l(g_switch_mutex);
    if (g_switches.empty()) return nullptr;
    auto it = g_switches.find(name);
    if (it == g_switches.end()) return nullptr;
    value = it->second;
    return value.c_str();
}
}  // namespace pilot

PILOT_API int pilot_ot_test_switch(const char *name, const char *value) {
    std::lock_guard<std::mutex> l(g_switch_mutex);
    if (!name) { g_switches.clear(); return PILOT_OT_OK; }
    if (value) g_switches[name] = value; else g_switches.erase(name);
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_version(void) { return PILOT_OT_VERSION; }
PILOT_API const char *pilot_ot_last_error(void) { return g_err; }

PILOT_API int pilot_ot_device_count(int *count) {
    if (!count) return fail(PILOT_OT_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_get_device(int *device) {
    if (!device) return fail(PILOT_OT_EINVAL, "device is NULL");
    HIP_TRY(hipGetDevice(device));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_device_name(char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(PILOT_OT_EINVAL, "bad buffer");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_dev_alloc(void **dptr, unsigned long long bytes) {
    if (!dptr) return fail(PILOT_OT_EINVAL, "dptr is NULL");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return PILOT_OT_OK;
}
PILOT_API int pilot_ot_dev_free(void *dptr) {
    HIP_TRY(hipFree(dptr));
    return PILOT_OT_OK;
}
PILOT_API int pilot_ot_memcpy_h2d(void *dst, const void *src, unsigned long long bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PILOT_OT_OK;
}
PILOT_API int pilot_ot_memcpy_d2h(void *dst, const void *src, unsigned long long bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}
PILOT_API int pilot_ot_stream_sync(void *stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PILOT_OT_OK;
}

// ------------------------------------------------------------------------------------------------
namespace { hipError_t ws_get(int slot, size_t bytes, void **out); }    // the calling thread's pool of temporaries (below)

PILOT_API int pilot_ot_cost_matrix_dev_ex(const double *d_centroids, int K, int D, int metric, const double *d_aux, double *d_cost,
                                          void *stream) {
    if (!d_centroids || !d_cost) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (K <= 0 || D <= 0) return fail(PILOT_OT_EINVAL, "K=%d D=%d must be positive", K, D);
    if (metric < PILOT_OT_METRIC_COSINE || metric > PILOT_OT_METRIC_MAHALANOBIS)
        return fail(PILOT_OT_EINVAL, "unknown metric id %d", metric);
    if (metric == PILOT_OT_METRIC_MAHALANOBIS && !d_aux) return fail(PILOT_OT_EINVAL, "mahalanobis needs the D x D inverse covariance (aux)");
    if (K > 4096 || D > 4096) return fail(PILOT_OT_ENOTSUP, "K=%d D=%d: at most 4096 centroids / dimensions", K, D);
    hipLaunchKernelGGL(cost_matrix_kernel, dim3(1), dim3(1024), sizeof(double) * (2 * K + D),
                       static_cast<hipStream_t>(stream), d_centroids, K, D, metric, d_aux, d_cost);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_cost_matrix_dev(const double *d_centroids, int K, int D, int metric, double *d_cost, void *stream) {
    return pilot_ot_cost_matrix_dev_ex(d_centroids, K, D, metric, nullptr, d_cost, stream);
}

PILOT_API int pilot_ot_cost_matrix_ex(const double *centroids, int K, int D, int metric, const double *aux, double *cost) {
    if (!centroids || !cost) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (K <= 0 || D <= 0) return fail(PILOT_OT_EINVAL, "K=%d D=%d must be positive", K, D);
    const bool has_aux = metric == PILOT_OT_METRIC_MAHALANOBIS;
    if (has_aux && !aux) return fail(PILOT_OT_EINVAL, "mahalanobis needs the D x D inverse covariance (aux)");
    // staging from the calling thread's pool (slots 9 .. 11: the pre-pass calls use 0 .. 8)
    void *dx = nullptr, *dc = nullptr, *da = nullptr;
    HIP_TRY(ws_get(9, sizeof(double) * (size_t)K * D, &dx));
    HIP_TRY(ws_get(10, sizeof(double) * (size_t)K * K, &dc));
    HIP_TRY(hipMemcpy(dx, centroids, sizeof(double) * (size_t)K * D, hipMemcpyHostToDevice));
    if (has_aux) {
        HIP_TRY(ws_get(11, sizeof(double) * (size_t)D * D, &da));
        HIP_TRY(hipMemcpy(da, aux, sizeof(double) * (size_t)D * D, hipMemcpyHostToDevice));
    }
    const int rc = pilot_ot_cost_matrix_dev_ex(static_cast<double *>(dx), K, D, metric, static_cast<double *>(da), static_cast<double *>(dc), nullptr);
    if (rc != PILOT_OT_OK) return rc;
    HIP_TRY(hipMemcpy(cost, dc, sizeof(double) * (size_t)K * K, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_cost_matrix(const double *centroids, int K, int D, int metric, double *cost) {
    return pilot_ot_cost_matrix_ex(centroids, K, D, metric, nullptr, cost);
}

// ------------------------------------------------------------------------------------------------
// range of the fp16-split configuration (PILOT_OT_H_MAX_COST_OVER_REG: experiment switch of tools/f16x2_range_probe.py)
static double h_max_cost_over_reg() {
    const char *e = pilot::test_switch("PILOT_OT_H_MAX_COST_OVER_REG");
    return e && *e ? atof(e) : pilot::H_MAX_COST_OVER_REG;
}

PILOT_API int pilot_ot_auto_precision(double max_cost_over_reg) {
    // f32 keeps every Gibbs-kernel entry exp(-M/reg) a well-scaled normal number only while
    // max(M)/reg stays clear of the f32 exponent range (ln FLT_MIN = -87.3); beyond ~60 the
    // far-transport entries lose bits, so AUTO switches to the f64 kernel.
    // Inside that range the f32 values are iterated with bf16-split products (PILOT_OT_PREC_BF16X3: f32-level rounding on
    // the bf16 matrix pipe, measured 1.3x the f32-input MFMA path).
    // While max(M)/reg <= 16 (PILOT's default reg = 0.1 on the max-normalised cost gives 10) every entry of 2^15 exp(-M/reg)
    // is a fp16 pair good to <= 2^-15.9 relative (22 bits down to 11.8; see H_MAX_COST_OVER_REG) and the products run on 2-way fp16 splits (PILOT_OT_PREC_F16X2: half the MFMAs, a third of
    // the split instructions of BF16X3; same stopping checks, same 1e-7 class distance to the fp64 oracle).
    if (max_cost_over_reg <= h_max_cost_over_reg()) return PILOT_OT_PREC_F16X2;
    return max_cost_over_reg <= 60.0 ? PILOT_OT_PREC_BF16X3 : PILOT_OT_PREC_F64;
}

namespace { bool split_fits_lds(int K, bool sym, int bands); }
PILOT_API int pilot_ot_auto_precision_for(double max_cost_over_reg, int K, int cost_is_symmetric) {
    int prec = pilot_ot_auto_precision(max_cost_over_reg);
    if ((prec == PILOT_OT_PREC_BF16X3 || prec == PILOT_OT_PREC_F16X2) && !split_fits_lds(K, cost_is_symmetric != 0, 1)) prec = PILOT_OT_PREC_F32;
    return prec;
}

// The one place where a requested precision becomes the precision a call runs (host, multi-device and device entry points):
//  * exp(-max(M)/reg) outside the f64 range, or on request -> POT-literal kernel;
//  * AUTO -> F16X2 / BF16X3 by range (F32 where the split images do not fit LDS), AUTO_MIXED beyond the f32 range;
//  * an explicit f32-class precision beyond the f32 range (max(M)/reg > 60) would be off by up to 1e-4 on this path's
//    distributions: it runs AUTO_MIXED as well -- explicit precisions are honoured inside their valid range only;
//  * F16X2 outside its scaled domain (cost range, tau) -> BF16X3.
PILOT_API int pilot_ot_resolve_precision(int precision, double max_cost_over_reg, int K, int cost_is_symmetric, double tau) {
    if (precision == PILOT_OT_PREC_GENERIC || max_cost_over_reg > PILOT_OT_MAX_COST_OVER_REG) return PILOT_OT_PREC_GENERIC;
    const bool f32_class = precision == PILOT_OT_PREC_F32 || precision == PILOT_OT_PREC_BF16X3 || precision == PILOT_OT_PREC_F16X2;
    // (PILOT_OT_RAW_PRECISION=1, tests only: run an explicit f32-class precision outside its range as asked)
    const char *raw = pilot::test_switch("PILOT_OT_RAW_PRECISION");
    const bool promote = f32_class && max_cost_over_reg > 60.0 && !(raw && *raw && *raw != '0');
    if (precision == PILOT_OT_PREC_AUTO || promote) {
        precision = pilot_ot_auto_precision_for(max_cost_over_reg, K, cost_is_symmetric);
        if (precision == PILOT_OT_PREC_F64) precision = PILOT_OT_PREC_AUTO_MIXED;    // f32 first, f64 for the pairs that need it
    }
    if (precision == PILOT_OT_PREC_F16X2 &&
        (max_cost_over_reg > h_max_cost_over_reg() || tau > pilot::H_MAX_TAU || !split_fits_lds(K, cost_is_symmetric != 0, 1)))
        precision = split_fits_lds(K, cost_is_symmetric != 0, 1) ? PILOT_OT_PREC_BF16X3 : PILOT_OT_PREC_F32;
    return precision;
}

PILOT_API int pilot_ot_plan_create(int N, int K, pilot_ot_plan **plan) {
    if (!plan) return fail(PILOT_OT_EINVAL, "plan is NULL");
    if (N <= 0 || K <= 0) return fail(PILOT_OT_EINVAL, "N=%d K=%d must be positive", N, K);
    if (K > GENERIC_MAX_K) return fail(PILOT_OT_ENOTSUP, "K=%d > %d cell types", K, GENERIC_MAX_K);
    if ((long long)N * N > 0x7fffffffLL) return fail(PILOT_OT_ENOTSUP, "N=%d: N*N overflows the pair index", N);
    pilot_ot_plan *pl = new (std::nothrow) pilot_ot_plan();
    if (!pl) return fail(PILOT_OT_EINVAL, "out of host memory");
    pl->N = N; pl->K = K; pl->max_cost = 1.0;
    pl->img = nullptr; pl->p_slot = nullptr; pl->track_list = nullptr; pl->track_count = nullptr;
    pl->emd_counter = nullptr; pl->f_slab = nullptr; pl->f_slab_bytes = 0; pl->emdg_slab = nullptr; pl->emdg_wgs = 0; pl->n_cu = 256; pl->kws = nullptr; pl->generic_wgs = 0; pl->nan_list = nullptr; pl->nan_list_n = 0;
    pl->wide_rec = nullptr; pl->wide_rec_n = 0;
    pl->order_list = nullptr; pl->order_bucket = nullptr; pl->order_hist = nullptr;
    pl->flags_ws = nullptr; pl->flags_ws_n = 0;
    pl->timing = 0; pl->n_timed = 0; pl->n_calls = 0;
    pl->graph_mode = 0; pl->gkey_seen = 0; pl->gstream = nullptr; pl->gexec = nullptr;
    for (int i = 0; i < TIMING_RING; ++i) for (int j = 0; j < 4; ++j) pl->ev[i][j] = nullptr;
    hipError_t e = hipGetDevice(&pl->device);
    if (e == hipSuccess) {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, pl->device) == hipSuccess && n_cu > 0)
            pl->n_cu = n_cu;
    }
    const int kp = ((K + 31) / 32) * 32;
    if (K <= MAX_K) {
        const int rt = (K + 15) / 16;
        size_t img_bytes = pilot::img_elems(pilot::CFG_F64, rt) * sizeof(double);
        const size_t b32 = pilot::img_elems(pilot::CFG_F32, rt) * sizeof(float), bs = pilot::img_elems(pilot::CFG_S32, rt) * sizeof(float);
        const size_t bh = pilot::img_elems(pilot::CFG_H32, rt) * sizeof(float);
        if (b32 > img_bytes) img_bytes = b32;
        if (bs > img_bytes) img_bytes = bs;
        if (bh > img_bytes) img_bytes = bh;
        if (e == hipSuccess) e = hipMalloc(&pl->img, img_bytes);
    } else if (K <= WIDE_MAX_K) {       // the 8-waves-per-tile kernel: the fp16-split operand block at 16 row-tiles
        if (e == hipSuccess) e = hipMalloc(&pl->img, pilot::img_elems(pilot::CFG_H32, 16) * sizeof(float));
    }
    {
        // proportions in slot order + one stop threshold per patient: sized for f64 at the padded K; the fp16-split configuration keeps
        // TWO f32 copies there (plain, and in its scaled domain) -- the same bytes up to K = 128, more with the 16 row-tiles of the
        // eight-waves-per-tile kernel
        size_t p_bytes = sizeof(double) * ((size_t)N * kp + N);
        if (K > MAX_K && K <= WIDE_MAX_K) { const size_t two = 2 * sizeof(float) * ((size_t)N * 256 + N); p_bytes = two > p_bytes ? two : p_bytes; }
        if (e == hipSuccess) e = hipMalloc(&pl->p_slot, p_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->track_list), sizeof(int) * (size_t)N * N);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->track_count), CTRL_BLOCK_INTS * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->order_list), sizeof(int) * (size_t)N * N);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->order_bucket), (size_t)N * N);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->emd_counter), sizeof(int) * pilot::EMD_NQ * pilot::EMD_Q_STRIDE);
    // per-pair work lists of the full grid, so that no grid call allocates (the POT-literal kernel's scratch, 2 K^2 doubles per
    // resident workgroup, is the exception: allocated by the first call that needs that kernel)
    // (two lists of N^2: pairs that ended in NaN, and pairs the f32 passes hand to the f64 pass)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->nan_list), 2 * sizeof(int) * (size_t)N * N);
    if (e == hipSuccess) pl->nan_list_n = (size_t)N * N;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl->flags_ws), sizeof(int) * (size_t)N * N);
    if (e == hipSuccess) pl->flags_ws_n = (size_t)N * N;
    if (e != hipSuccess) {
        pilot_ot_plan_destroy(pl);
        return fail(PILOT_OT_EHIP, "plan allocation failed: %s", hipGetErrorString(e));
    }
    *plan = pl;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_plan_set_max_cost(pilot_ot_plan *pl, double max_cost) {
    if (!pl) return fail(PILOT_OT_EINVAL, "plan is NULL");
    if (!(max_cost > 0.0) || !std::isfinite(max_cost)) return fail(PILOT_OT_EINVAL, "max_cost=%g must be positive and finite", max_cost);
    pl->max_cost = max_cost;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_plan_destroy(pilot_ot_plan *pl) {
    if (!pl) return PILOT_OT_OK;
    if (pl->img) (void)hipFree(pl->img);
    if (pl->p_slot) (void)hipFree(pl->p_slot);
    if (pl->track_list) (void)hipFree(pl->track_list);
    if (pl->track_count) (void)hipFree(pl->track_count);
    if (pl->order_list) (void)hipFree(pl->order_list);
    if (pl->order_bucket) (void)hipFree(pl->order_bucket);
    if (pl->flags_ws) (void)hipFree(pl->flags_ws);
    if (pl->emd_counter) (void)hipFree(pl->emd_counter);
    if (pl->f_slab) (void)hipFree(pl->f_slab);
    if (pl->emdg_slab) (void)hipFree(pl->emdg_slab);
    if (pl->kws) (void)hipFree(pl->kws);
    if (pl->nan_list) (void)hipFree(pl->nan_list);
    if (pl->wide_rec) (void)hipFree(pl->wide_rec);
    if (pl->gexec) (void)hipGraphExecDestroy(pl->gexec);
    if (pl->gstream) (void)hipStreamDestroy(pl->gstream);
    for (int i = 0; i < TIMING_RING; ++i) for (int j = 0; j < 4; ++j) if (pl->ev[i][j]) (void)hipEventDestroy(pl->ev[i][j]);
    delete pl;
    return PILOT_OT_OK;
}

namespace {

int check_grid_args(int N, int K, double reg, int num_iter_max, double stop_thr, double tau, int check_period,
                    int precision, int row_begin, int row_end, int row_step) {
    if (N <= 0 || K <= 0) return fail(PILOT_OT_EINVAL, "N=%d K=%d must be positive", N, K);
    if (!(reg > 0.0) || !std::isfinite(reg)) return fail(PILOT_OT_EINVAL, "reg=%g must be positive and finite", reg);
    if (num_iter_max < 1) return fail(PILOT_OT_EINVAL, "num_iter_max=%d must be >= 1", num_iter_max);
    if (check_period < 1) return fail(PILOT_OT_EINVAL, "check_period=%d must be >= 1", check_period);
    if (!(stop_thr >= 0.0) || !(stop_thr < 1.0)) return fail(PILOT_OT_EINVAL, "stop_thr=%g must be in [0, 1)", stop_thr);
    if (!(tau > 1.0)) return fail(PILOT_OT_EINVAL, "tau=%g must be > 1", tau);
    if (precision < PILOT_OT_PREC_AUTO || precision > PILOT_OT_PREC_F16X2)
        return fail(PILOT_OT_EINVAL, "unknown precision id %d", precision);
    if (row_step < 1 || row_begin < 0 || row_end > N || row_begin > row_end)
        return fail(PILOT_OT_EINVAL, "bad row range [%d, %d) step %d for N=%d", row_begin, row_end, row_step, N);
    if (K > GENERIC_MAX_K) return fail(PILOT_OT_ENOTSUP, "K=%d > %d cell types", K, GENERIC_MAX_K);
    return PILOT_OT_OK;
}

// The MFMA kernels iterate TOTAL scalings against the fixed Gibbs image exp(-M/reg) (POT's log-absorption is value-neutral
// and only its bookkeeping is tracked), so max(M)/reg must stay inside the exponent range of the widest type: beyond ~600
// an f64 Gibbs entry underflows / a total scaling overflows where POT's absorbed kernel would not.  Such calls, and K > 128,
// go to the reference-semantics kernel (generic_kernels.hpp), which rebuilds the absorbed kernel like POT.
constexpr double MAX_COST_OVER_REG = PILOT_OT_MAX_COST_OVER_REG;

// list / list_len (device, nullable): only the listed pairs (NaN hand-over of the fast kernels); queue: zeroed counter
int run_generic(pilot_ot_plan *pl, const double *d_P, const double *d_M, double reg, int num_iter_max, double stop_thr, double tau,
                int check_period, int row_begin, int n_rows, int row_step, double *d_emd, int *d_iters, double *d_err, int *d_flags,
                hipStream_t s, const int *list = nullptr, const int *list_len = nullptr, int *queue = nullptr) {
    const int N = pl->N, K = pl->K;
    const int n_pairs = n_rows * N;
    if (n_pairs == 0) return PILOT_OT_OK;
    // 8 vectors + nsplit rows of partial sums (as many as fit, a power of two <= the waves of a workgroup) + reduction scratch + queue slot
    int nsplit = pilot::GENERIC_WAVES;
    auto lds_for = [&](int ns) { return sizeof(double) * ((8 + (size_t)ns) * (size_t)K + pilot::GENERIC_WAVES) + 16; };
    while (nsplit > 1 && lds_for(nsplit) > LDS_BYTES) nsplit /= 2;
    const size_t lds = lds_for(nsplit);
    if (lds > LDS_BYTES) return fail(PILOT_OT_ENOTSUP, "K=%d does not fit the generic kernel's LDS vectors", K);
    if (!pl->kws) {
        // two workgroups per CU, fewer when K' and its transpose would take more than 8 GB in all
        int wgs = 3 * pl->n_cu;
        const size_t per = sizeof(double) * 2 * (size_t)K * K;
        while (wgs > 1 && per * wgs > ((size_t)8 << 30)) wgs /= 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->kws), per * wgs));
        pl->generic_wgs = wgs;
    }
    if (!d_flags) {
        if ((size_t)n_pairs > pl->flags_ws_n) {
            if (pl->flags_ws) HIP_TRY(hipFree(pl->flags_ws));
            pl->flags_ws = nullptr; pl->flags_ws_n = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->flags_ws), sizeof(int) * (size_t)n_pairs));
            pl->flags_ws_n = (size_t)n_pairs;
        }
        d_flags = pl->flags_ws;
    }
    if (!queue) { queue = pl->track_count; HIP_TRY(hipMemsetAsync(queue, 0, sizeof(int), s)); }
    pilot::GenericParams g;
    g.P = d_P; g.M = d_M; g.N = N; g.K = K; g.n_pairs = n_pairs; g.row_begin = row_begin; g.row_step = row_step;
    g.reg = reg; g.tau = tau; g.stop_thr = stop_thr; g.max_iter = num_iter_max; g.period = check_period;
    g.emd = d_emd; g.iters = d_iters; g.err = d_err; g.flags = d_flags; g.kws = pl->kws; g.queue = queue; g.nsplit = nsplit;
    g.list = list; g.list_len = list_len;
    int wgs = pl->generic_wgs < n_pairs ? pl->generic_wgs : n_pairs;
    if (list && wgs > 64) wgs = 64;          // a hand-over list is short (usually empty)
    if (const char *e = pilot::test_switch("PILOT_OT_GENERIC_WGS")) { const int w = atoi(e); if (w > 0 && w < wgs) wgs = w; }      // experiment switch
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::sinkhorn_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pilot::sinkhorn_generic_kernel, dim3(wgs), dim3(pilot::GENERIC_WG), lds, s, g);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

// resident workgroups per CU of the single-tile stream kernel (mirrors pilot::min_waves_per_simd)
int stream_min_waves(int w /* sizeof(T)/4 */, int RT, bool sym, bool track, int tv, bool split, bool half = false) {
    if (half && !track && RT <= pilot::HALF_OCC4_MAX_RT) return 4;
    if (split) return RT <= (track ? pilot::SPLIT_OCC2_MAX_RT_TRACK : (half ? pilot::HALF_OCC2_MAX_RT : pilot::SPLIT_OCC2_MAX_RT)) ? 2 : 1;
    const int na = RT * 4 * RT * w;
    const bool greg = !split && sym && na <= pilot::GREG_MAX;
    const int regs = (track ? 7 : 5) * RT * 4 * w + 4 * w + 56 + (tv ? 24 * w : 0) + (split ? 3 * ((RT + 1) / 2) * 4 + 24 : 0) +
                     (greg ? na + 2 * tv * ((RT - 1) * 4 + 1) * w : 0);
    return regs <= 128 ? 4 : (regs <= 168 ? 3 : (regs <= 256 ? 2 : 1));
}
// mirrors pilot::solo_in_stream: does the fast launch of this configuration carry the one-wave-per-pair path?
bool stream_has_solo(int w, int RT, bool sym, int tv, bool split, bool half) {
    const int mw = stream_min_waves(w, RT, sym, false, tv, split, half);
    const int budget = mw >= 4 ? 128 : (mw == 3 ? 168 : 256);
    return sym && RT <= 4 && !(half && RT < pilot::HALF_SOLO_MIN_RT) && (64 + 45) * w <= budget;
}

// LDS of one stream-kernel workgroup: operand image(s) + first-product table + tail weights + one ring of finished pairs
// per wave.  The ring gets as many slots (<= 16) as fit while `want` workgroups stay resident per CU; at least 4.
struct StreamLds { size_t bytes; int ring, wgs_per_cu; };
StreamLds stream_lds(size_t fixed, size_t slot_bytes, int want, int min_ring = 4) {
    StreamLds r;
    for (;;) {
        const size_t budget = LDS_BYTES / (size_t)want;
        long ring = budget > fixed ? (long)((budget - fixed) / ((size_t)pilot::WAVES_PER_WG * slot_bytes)) : 0;
        if (ring >= min_ring || want == 1) {
            if (ring > pilot::RING_MAX) ring = pilot::RING_MAX;
            if (ring < 1) ring = 0;
            r.ring = (int)ring; r.wgs_per_cu = want;
            r.bytes = fixed + (size_t)pilot::WAVES_PER_WG * slot_bytes * (size_t)ring;
            return r;
        }
        --want;
    }
}

// does the bf16-split configuration fit LDS at this K (operand image(s) + table + a minimal ring)?
// (the fp16-split configuration needs less for its fast pass and the same for its tracking pass)
bool split_fits_lds(int K, bool sym, int bands = 1) {
    const int RT = (K + 15) / 16, KP = RT * 16;
    const size_t fixed = (size_t)(sym ? 1 : 2) * pilot::form_elems_rt(pilot::CFG_S32, RT) * 4 * bands + (size_t)KP * 4 +
                         (size_t)pilot::WAVES_PER_WG * pilot::HANDOVER_BUF * sizeof(int);
    return fixed + (size_t)4 * pilot::WAVES_PER_WG * (2 * KP + 4) * 4 <= LDS_BYTES;
}

// cfg: pilot::CFG_F32 / CFG_F64 / CFG_S32 / CFG_H32 (all 16-pair tiles: TILE = 16, 4 accumulator registers, 4 lane groups)
// CFG_H32 (fp16-split): the fast pass only; its tracking pass is the bf16-split kernel on the second operand block.
// mixed (cfg == CFG_S32 only): small reg under PILOT_OT_PREC_AUTO -- every pair is first iterated in f32 (bf16-split
// products, tau-tracking kernel); pairs whose plan may touch Gibbs entries outside the f32-safe range, or that went NaN, are
// collected (ring_flush) and solved again by the f64 tracking kernel.
int run_grid(int cfg, pilot_ot_plan *pl, const double *d_P, const double *d_M, double reg, int num_iter_max,
             double stop_thr, double tau, int check_period, double floor_ulps, bool sym, int row_begin,
             int n_rows, int row_step, double *d_emd, int *d_iters, double *d_err, int *d_flags, hipStream_t s,
             bool mixed = false) {
    const bool f64 = cfg == pilot::CFG_F64, half = cfg == pilot::CFG_H32, split = cfg == pilot::CFG_S32 || half;
    const size_t ts = f64 ? sizeof(double) : sizeof(float);
    const int w = (int)(ts / 4);
    constexpr int TILE = 16;
    const int N = pl->N, K = pl->K;
    const int RT = (K + TILE - 1) / TILE;
    const int KP = RT * TILE;
    const char *dbg = pilot::test_switch("PILOT_OT_DEBUG");
    const int debug = dbg ? atoi(dbg) : 0;
    const size_t form = pilot::form_elems_rt(cfg, RT);
    // the per-wave hand-over buffers of the fast kernels
    const size_t hb_bytes = (size_t)pilot::WAVES_PER_WG * pilot::HANDOVER_BUF * sizeof(int);
    size_t fixed = (size_t)(sym ? 1 : 2) * form * ts + (size_t)KP * ts + hb_bytes;   // operand image(s) + first-product table + hand-over buffers
    // K mod 16 in 1..4: the (at most four) cell types of the last row-tile are computed on the VALU (tail_rows)
    int tv = 0;
    // split: skip the dead registers of the last tile (beyond 4 row-tiles those variants run out of registers and spill
    // 600-980 B per lane; the plain variants do not, and measure the same there)
    const int live1 = (split && RT >= 2 && RT <= 4 && K - (RT - 1) * TILE <= 4) ? 1 : 0;
    if (!split) {
        const int n_tail = K - (RT - 1) * TILE;
        // (RT = 8 variants spill: left on the MFMA path)
        if (RT >= 2 && RT <= 7 && n_tail <= 4 && !(debug & 256)) tv = n_tail <= 2 ? 1 : 2;
        const size_t tail_lds = (size_t)(sym ? 1 : 2) * tv * ((RT - 1) * 4 + 1) * 64 * 2 * ts;
        if (tv && fixed + tail_lds + 4 * pilot::WAVES_PER_WG * (2 * KP + 4) * ts > LDS_BYTES) tv = 0;   // no room: MFMA path
        if (tv) fixed += tail_lds;
    }
    const size_t slot_bytes = (size_t)(2 * KP + 4) * ts;
    if (fixed + pilot::WAVES_PER_WG * slot_bytes > LDS_BYTES)
        return fail(PILOT_OT_ENOTSUP, "K=%d with a %ssymmetric cost needs %zu B of LDS (> %zu) in this precision", K, sym ? "" : "non-",
                    fixed + pilot::WAVES_PER_WG * slot_bytes, LDS_BYTES);
    pl->order_hist = pl->track_count + CTRL_INTS;     // one control block, one memset per call
    HIP_TRY(hipMemsetAsync(pl->track_count, 0, CTRL_BLOCK_INTS * sizeof(int), s));
    void *img = pl->img;
    void *Pt = pl->p_slot;
    if (n_rows == 0) return PILOT_OT_OK;

    const int n_pairs = n_rows * N;
    if (!d_flags) {
        if ((size_t)n_pairs > pl->flags_ws_n) {
            if (pl->flags_ws) HIP_TRY(hipFree(pl->flags_ws));
            pl->flags_ws = nullptr; pl->flags_ws_n = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->flags_ws), sizeof(int) * (size_t)n_pairs));
            pl->flags_ws_n = (size_t)n_pairs;
        }
        d_flags = pl->flags_ws;
    }

    pilot::GridParams p;
    p.P = Pt; p.img = img; p.N = N; p.K = K;
    p.n_pairs = n_pairs;
    p.list = nullptr; p.list_len = nullptr;
    p.solo_len = nullptr; p.solo_head = nullptr; p.solo_blocks = 0;
    p.row_begin = row_begin; p.row_step = row_step;
    p.max_iter = num_iter_max; p.period = check_period;
    p.stop_thr = stop_thr; p.tau = tau; p.floor_ulps = floor_ulps;
    p.emd = d_emd; p.iters = d_iters; p.err = d_err; p.flags = d_flags;
    p.track_list = pl->track_list; p.track_count = pl->track_count; p.queue_head = pl->track_count + 1;
    p.queue_shards = pl->track_count + CTRL_SHARDS_AT;       // (the fast launch up to two row-tiles; the list launches below draw from one counter)
    p.ring = 0;
    p.fb_list = nullptr; p.fb_count = nullptr; p.bands = 1;
    // pairs that end in NaN ("Numerical errors" in POT) are collected and re-solved by the POT-literal kernel, which
    // returns the last good iterate like POT does
    if ((size_t)n_pairs > pl->nan_list_n) {
        if (pl->nan_list) HIP_TRY(hipFree(pl->nan_list));
        pl->nan_list = nullptr; pl->nan_list_n = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->nan_list), 2 * sizeof(int) * (size_t)n_pairs));
        pl->nan_list_n = (size_t)n_pairs;
    }
    p.nan_list = pl->nan_list; p.nan_count = pl->track_count + 10;
    p.unequal = pl->track_count + pilot::CTRL_UNEQUAL;
    // Between the fp16-split range and the two-band path (12 < max(M)/reg <= 60) a few pairs per matrix leave the f32 range in
    // the single-band kernels (a scaling jumps past the fp16 domain within one update; products underflow at reg <= 0.025).
    // They used to go to the POT-literal kernel with the other NaN pairs -- one workgroup per pair, 12.5 us per update: 3 to 13
    // pairs cost 12 ms of a 30 ms call at reg 0.025 .. 0.0175.  They are collected like the small-reg path collects its
    // hand-over and solved again by the f64 tracking kernel (symmetric cost, K <= 64: one wave per pair, 1.1 us per update).
    const bool redo64 = !mixed && split && pl->max_cost / reg > 12.0 && !(debug & 4096);
    // From max(M)/reg = 24 on nearly every pair tau-absorbs (c3: 28 % at 20, 94 % at 25) and the fast pass only hands its pairs
    // over after a few dozen wasted updates (3.7 of 11.6 ms at reg 0.04): every pair goes to the tracking kernel at once, as
    // in the two-band path.
    const bool track_all = mixed || (split && !half && pl->max_cost / reg > 24.0 && !(debug & 8192));
    int *const fb_list = pl->nan_list + pl->nan_list_n;
    if (redo64) { p.fb_list = fb_list; p.fb_count = pl->track_count + 8; }
    p.debug = debug;
    const int tiles = (n_pairs + TILE - 1) / TILE;
    // exact duplicates (a == b): one wave per pair in the leading workgroups of the fast launch (symmetric cost, K <= 64)
    // ... while the grid is small.  A wave that iterates ONE pair has the shorter update (K = 50: 0.57 us against 0.77 us for a lone
    // 16-pair wave), which is what a launch with fewer tiles than wave slots waits for (c2; the row shards of a multi-device call);
    // on a full device the 600 diagonal pairs of c3 (165 updates on average, up to 301) on 600 waves of their own are a tail instead:
    // main kernel 0.617 -> 0.584 ms with the duplicates in the tiles (tools/solo_probe.py; crossover between 5 600 and 7 500 tiles at 2 048
    // wave slots).  The rule reads the FULL grid (N x N), not the rows of this call: a row shard and the full grid send the same pair
    // down the same path, so their bits agree.
    const long full_tiles = ((long)N * N + TILE - 1) / TILE;
    const long wave_slots = (long)pl->n_cu * stream_min_waves(w, RT, sym, false, tv, split, half) * pilot::WAVES_PER_WG;
    const bool solo = stream_has_solo(w, RT, sym, tv, split, half) && !(p.debug & 512) && !track_all && full_tiles < 3 * wave_slots;
    int solo_blocks = 0;
    {
        // longest-first work order (see order_bucket_kernel)
        int ob = (n_pairs + 1023) / 1024;
        if (ob > pl->n_cu) ob = pl->n_cu;
        int *split_ctl = pl->track_count + 4;
        const int mode = (solo ? 2 : 0) | ((p.debug & 2) ? 4 : 0);   // bit 2: natural order (experiment)
        HIP_TRY(pilot::launch_prep(cfg, d_M, K, RT, reg, img, d_P, Pt, N, (tv ? 1 : 0) | 2 | (mixed ? 4 : 0), stop_thr, floor_ulps, n_rows, row_begin, row_step,
                                   pl->order_bucket, pl->order_hist, pl->order_list, split_ctl, pl->track_count + 1, mode, ob, s));
        p.list = pl->order_list;
        if (solo) {
            solo_blocks = (n_rows + pilot::WAVES_PER_WG - 1) / pilot::WAVES_PER_WG;     // the diagonal; more duplicates queue up
            if (solo_blocks > pl->n_cu) solo_blocks = pl->n_cu;
            p.solo_len = split_ctl + 3; p.solo_head = pl->track_count + 3; p.solo_blocks = solo_blocks;
        }
    }
    hipEvent_t *ev = (pl->timing > 0 && (pl->n_calls++ % pl->timing) == 0) ? pl->ev[pl->n_timed % TIMING_RING] : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    // (fp16-split configuration, 112 < K <= 128, symmetric cost: four waves per tile, one tile per workgroup, two workgroups per CU)
    const bool quad = half && pilot::quad_covers(K, sym) && !pilot::test_switch("PILOT_OT_NO_QUAD");
    auto launch = [&](int tvv, bool track, int wgs, const StreamLds &L) -> hipError_t {
        p.ring = L.ring;
        if (tvv) return pilot::launch_stream_tv(cfg, tvv, RT, sym, track, dim3(wgs), L.bytes, s, p);
        if (half && !track && quad) return pilot::launch_quad(dim3(wgs), s, p);
        if (half && !track) return pilot::launch_stream_h32(RT, sym, live1, dim3(wgs), L.bytes, s, p);
        if (split) return pilot::launch_stream_s32(RT, sym, track, live1, dim3(wgs), L.bytes, s, p);
        return f64 ? pilot::launch_stream_f64(RT, sym, track, dim3(wgs), L.bytes, s, p)
                   : pilot::launch_stream_f32(RT, sym, track, dim3(wgs), L.bytes, s, p);
    };
    // first pass: throughput kernel (pairs that would tau-absorb are handed to the second pass)
    if (!track_all && quad) {
        int wgs = 2 * pl->n_cu;
        if (wgs > tiles) wgs = tiles;
        HIP_TRY(launch(tv, false, wgs, StreamLds{}));
    } else if (!track_all) {
        int want = stream_min_waves(w, RT, sym, false, tv, split, half);
        // (K <= 4: a third of the pairs tau-absorb and are handed over, and the hand-over's atomics and list stores are what more resident
        // waves contend for -- K = 3 / 4 at N = 600: 0.60 / 0.64 ms at two workgroups per CU, 0.69 / 0.72 at four; from K = 5 on four win)
        if (half && K <= 4 && want > 2) want = 2;
        if ((p.debug >> 4) & 7) want = (p.debug >> 4) & 7;           // experiment: resident workgroups per CU
        // split configurations up to 4 row-tiles flush their ring inline and park U in LDS meanwhile (pilot::parked_flush):
        // one 16-byte line per lane and row-tile
        const bool park = split && RT <= 4;
        // (fp16-split configuration: ring slots and the park area hold packed pieces -- whole k-blocks, so an odd row-tile
        // count rounds up)
        const size_t slot_fast = half ? (size_t)pilot::ring_slot_stride<pilot::CfgH32x16>(RT) * ts : slot_bytes;
        const size_t park_lane = half ? (size_t)pilot::park_lane_elems<pilot::CfgH32x16>(RT) : (size_t)RT * 4;
        const size_t park_bytes = park ? (size_t)pilot::WAVES_PER_WG * park_lane * 64 * ts : 0;
        if (fixed + park_bytes + pilot::WAVES_PER_WG * slot_fast > LDS_BYTES)
            return fail(PILOT_OT_ENOTSUP, "K=%d: operand images + ring + park area exceed LDS", K);
        const StreamLds L = stream_lds(fixed + park_bytes, slot_fast, want);
        int wgs = pl->n_cu * L.wgs_per_cu;
        const int need = (tiles + pilot::WAVES_PER_WG - 1) / pilot::WAVES_PER_WG;
        if (wgs > need) wgs = need;
        wgs += solo_blocks;
        HIP_TRY(launch(tv, false, wgs, L));
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[1], s)); HIP_TRY(hipEventRecord(ev[2], s)); }
    // second pass: pairs in which POT would tau-absorb, with the absorption iterations tracked
    p.list = pl->track_list; p.list_len = pl->track_count; p.queue_head = pl->track_count + 2;
    p.queue_shards = pl->track_count + CTRL_SHARDS_TRACK_AT;     // (used by the tracking kernels up to two row-tiles)
    p.solo_blocks = 0;
    size_t fixed_t = fixed;
    if (half) {     // tracking pass of the fp16-split configuration: the bf16-split kernel on its own operand block
        p.img = static_cast<float *>(img) + pilot::track_img_elems(cfg, RT);
        fixed_t = (size_t)(sym ? 1 : 2) * pilot::form_elems_rt(pilot::CFG_S32, RT) * ts + (size_t)KP * ts + hb_bytes;
    }
    if (track_all) { p.list = pl->order_list; p.list_len = nullptr; }     // EVERY pair goes through the tracking kernel (longest first)
    if (mixed) {
        // small reg: the Gibbs kernel in two exponent bands; pairs that still leave the f32 range are collected in track_list
        // for the f64 pass
        p.bands = 2;
        p.fb_list = pl->track_list; p.fb_count = pl->track_count + 8;
        fixed_t = (size_t)(sym ? 1 : 2) * form * ts * 2 + (size_t)KP * ts + hb_bytes;
    }
    {
        // (the tracking kernel's result need not match the fast kernels' bits: a pair is always solved by one of them)
        const int tv_t = RT <= 4 ? tv : 0;          // larger tracking variants spill with the tail rows
        const StreamLds L = stream_lds(fixed_t, slot_bytes, stream_min_waves(w, RT, sym, true, tv_t, split));
        int wgs_t = pl->n_cu * L.wgs_per_cu;
        const int need = (tiles + pilot::WAVES_PER_WG - 1) / pilot::WAVES_PER_WG;
        if (wgs_t > need) wgs_t = need;
        HIP_TRY(launch(tv_t, true, wgs_t, L));
    }
    if (mixed || redo64) {
        // third pass: the collected pairs in f64 (operand images and proportions rebuilt for the f64 configuration in the
        // same buffers -- the f32 passes are complete in stream order; no ordering, the list is short)
        const int RT64 = RT;
        const size_t form64 = pilot::form_elems_rt(pilot::CFG_F64, RT64);
        const size_t fixed64 = (size_t)(sym ? 1 : 2) * form64 * sizeof(double) + (size_t)KP * sizeof(double);
        const size_t slot64 = (size_t)(2 * KP + 4) * sizeof(double);
        if (fixed64 + pilot::WAVES_PER_WG * slot64 > LDS_BYTES)
            return fail(PILOT_OT_ENOTSUP, "K=%d: the f64 fallback needs %zu B of LDS", K, fixed64 + pilot::WAVES_PER_WG * slot64);
        HIP_TRY(pilot::launch_prep(pilot::CFG_F64, d_M, K, RT64, reg, img, d_P, Pt, N, 2, stop_thr, floor_ulps, 0, row_begin, row_step,
                                   pl->order_bucket, pl->order_hist, pl->order_list, pl->track_count + 4, pl->track_count + 1, 0, 1, s));
        pilot::GridParams q = p;
        q.list = mixed ? pl->track_list : fb_list; q.list_len = pl->track_count + 8; q.queue_head = pl->track_count + 9; q.queue_shards = nullptr;
        q.img = img;                        // (the fp16-split configuration's tracking pass had moved it to its own block)
        q.fb_list = nullptr; q.fb_count = nullptr; q.bands = 1;
        q.nan_list = pl->nan_list; q.nan_count = pl->track_count + 10;
        if (sym && K <= 64 && !(p.debug & 2048)) {
            // the list is short (tens of pairs) and every pair on it runs long: one wave per pair, not 16-pair MFMA tiles
            HIP_TRY(pilot::launch_solo_track_f64(dim3(64), s, q));
        } else {
            const StreamLds L = stream_lds(fixed64, slot64, stream_min_waves(2, RT64, sym, true, 0, false));
            q.ring = L.ring;
            int wgs_t = pl->n_cu * L.wgs_per_cu;
            const int need = (tiles + pilot::WAVES_PER_WG - 1) / pilot::WAVES_PER_WG;
            if (wgs_t > need) wgs_t = need;
            HIP_TRY(pilot::launch_stream_f64(RT64, sym, true, dim3(wgs_t), L.bytes, s, q));
        }
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[3], s)); ++pl->n_timed; }
    if (!(p.debug & 1024)) {
        const int rc = run_generic(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, row_begin, n_rows, row_step, d_emd, d_iters,
                                   d_err, d_flags, s, pl->nan_list, pl->track_count + 10, pl->track_count + 11);
        if (rc != PILOT_OT_OK) return rc;
    }
    return PILOT_OT_OK;
}

// 128 < K <= 256 with a symmetric cost inside the fp16-split range: sinkhorn_wide_kernel (wide_kernels.hpp) on the operand
// block the ordinary prep kernel writes for 16 row-tiles, then the value kernel; hand-overs (tau-absorbing / NaN pairs, or
// every pair when the histograms carry unequal mass) are solved by the POT-literal kernel like those of the stream kernels.
int run_wide(pilot_ot_plan *pl, const double *d_P, const double *d_M, double reg, int num_iter_max, double stop_thr, double tau,
             int check_period, double floor_ulps, int row_begin, int n_rows, int row_step, double *d_emd, int *d_iters, double *d_err,
             int *d_flags, hipStream_t s) {
    const int N = pl->N, K = pl->K, RT = 16;
    // the records are 2 KB per pair: a big grid is solved in row chunks of at most WIDE_CHUNK_PAIRS pairs (1 GB of records),
    // each a complete call of its own (same kernels, same pair -> same bits whatever the chunking)
    long WIDE_CHUNK_PAIRS = 512L * 1024;
    if (const char *e = pilot::test_switch("PILOT_OT_WIDE_CHUNK")) { const long v = atol(e); if (v > 0) WIDE_CHUNK_PAIRS = v; }     // (tests)
    if ((long)n_rows * N > WIDE_CHUNK_PAIRS && n_rows > 1) {
        const int rows_per = (int)(WIDE_CHUNK_PAIRS / N) > 0 ? (int)(WIDE_CHUNK_PAIRS / N) : 1;
        for (int r0 = 0; r0 < n_rows; r0 += rows_per) {
            const int nr = n_rows - r0 < rows_per ? n_rows - r0 : rows_per;
            const size_t off = (size_t)r0 * N;
            const int rc = run_wide(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, floor_ulps, row_begin + r0 * row_step, nr, row_step,
                                    d_emd + off, d_iters ? d_iters + off : nullptr, d_err ? d_err + off : nullptr, d_flags ? d_flags + off : nullptr, s);
            if (rc != PILOT_OT_OK) return rc;
        }
        return PILOT_OT_OK;
    }
    pl->order_hist = pl->track_count + CTRL_INTS;
    HIP_TRY(hipMemsetAsync(pl->track_count, 0, (CTRL_INTS + 2 * pilot::ORDER_NB) * sizeof(int), s));
    if (n_rows == 0) return PILOT_OT_OK;
    const int n_pairs = n_rows * N;
    if (!d_flags) {
        if ((size_t)n_pairs > pl->flags_ws_n) {
            if (pl->flags_ws) HIP_TRY(hipFree(pl->flags_ws));
            pl->flags_ws = nullptr; pl->flags_ws_n = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->flags_ws), sizeof(int) * (size_t)n_pairs));
            pl->flags_ws_n = (size_t)n_pairs;
        }
        d_flags = pl->flags_ws;
    }
    if ((size_t)n_pairs > pl->nan_list_n) {
        if (pl->nan_list) HIP_TRY(hipFree(pl->nan_list));
        pl->nan_list = nullptr; pl->nan_list_n = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->nan_list), 2 * sizeof(int) * (size_t)n_pairs));
        pl->nan_list_n = (size_t)n_pairs;
    }
    if ((size_t)n_pairs > pl->wide_rec_n) {     // (first call of this size: the one allocation of the path)
        if (pl->wide_rec) HIP_TRY(hipFree(pl->wide_rec));
        pl->wide_rec = nullptr; pl->wide_rec_n = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->wide_rec), sizeof(float) * pilot::wide_rec_elems() * (size_t)n_pairs));
        pl->wide_rec_n = (size_t)n_pairs;
    }
    int ob = (n_pairs + 1023) / 1024;
    if (ob > pl->n_cu) ob = pl->n_cu;
    HIP_TRY(pilot::launch_prep(pilot::CFG_H32, d_M, K, RT, reg, pl->img, d_P, pl->p_slot, N, 0, stop_thr, floor_ulps, n_rows, row_begin, row_step,
                               pl->order_bucket, pl->order_hist, pl->order_list, pl->track_count + 4, pl->track_count + 1, 0, ob, s));
    pilot::GridParams p;
    p.P = pl->p_slot; p.img = pl->img; p.N = N; p.K = K;
    p.n_pairs = n_pairs;
    p.list = pl->order_list; p.list_len = nullptr;
    p.solo_len = nullptr; p.solo_head = nullptr; p.solo_blocks = 0;
    p.row_begin = row_begin; p.row_step = row_step;
    p.max_iter = num_iter_max; p.period = check_period;
    p.stop_thr = stop_thr; p.tau = tau; p.floor_ulps = floor_ulps;
    p.emd = d_emd; p.iters = d_iters; p.err = d_err; p.flags = d_flags;
    p.track_list = pl->track_list; p.track_count = pl->track_count; p.queue_head = pl->track_count + 1; p.queue_shards = nullptr;
    p.ring = 0; p.bands = 1;
    p.fb_list = nullptr; p.fb_count = nullptr;
    p.nan_list = pl->nan_list; p.nan_count = pl->track_count + 10;
    p.unequal = pl->track_count + pilot::CTRL_UNEQUAL;
    p.debug = 0;
    hipEvent_t *ev = (pl->timing > 0 && (pl->n_calls++ % pl->timing) == 0) ? pl->ev[pl->n_timed % TIMING_RING] : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    const int tiles = (n_pairs + 15) / 16;
    int wgs = pl->n_cu < tiles ? pl->n_cu : tiles;            // one 512-thread workgroup per CU (230 VGPRs: two waves per SIMD)
    HIP_TRY(pilot::launch_wide(dim3(wgs), s, p, pl->wide_rec));
    if (ev) { HIP_TRY(hipEventRecord(ev[1], s)); HIP_TRY(hipEventRecord(ev[2], s)); }
    int vwgs = (tiles + pilot::WAVES_PER_WG - 1) / pilot::WAVES_PER_WG;
    if (vwgs > 2 * pl->n_cu) vwgs = 2 * pl->n_cu;
    HIP_TRY(pilot::launch_wide_value(dim3(vwgs), s, p, pl->wide_rec));
    if (ev) { HIP_TRY(hipEventRecord(ev[3], s)); ++pl->n_timed; }
    return run_generic(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, row_begin, n_rows, row_step, d_emd, d_iters, d_err,
                       d_flags, s, pl->nan_list, pl->track_count + 10, pl->track_count + 11);
}

}  // namespace

PILOT_API int pilot_ot_sinkhorn_grid_dev(pilot_ot_plan *pl, const double *d_P, const double *d_M, double reg,
                                         int num_iter_max, double stop_thr, double tau, int check_period,
                                         int precision, double f32_floor_ulps, int cost_is_symmetric,
                                         int row_begin, int row_end, int row_step, double *d_emd, int *d_iters,
                                         double *d_err, int *d_flags, void *stream) {
    if (!pl || !d_P || !d_M || !d_emd) return fail(PILOT_OT_EINVAL, "NULL pointer");
    int rc = check_grid_args(pl->N, pl->K, reg, num_iter_max, stop_thr, tau, check_period, precision, row_begin,
                             row_end, row_step);
    if (rc != PILOT_OT_OK) return rc;
    {
        const int n_rows_g = (row_end - row_begin + row_step - 1) / row_step;
        // K beyond the MFMA kernels, a reg beyond the f64 range of exp(-M/reg) (judged by the plan's max_cost), or on request: POT's loop literally, absorbed kernel rebuilt per pair
        // 128 < K <= 256 (the fixed Gibbs image no longer fits one wave's registers and LDS): eight waves per tile while the
        // call is inside the fp16-split range with a symmetric cost; an explicit f64 / POT-literal request, a non-symmetric cost
        // or a smaller reg keep the POT-literal kernel
        if (pl->K > MAX_K && pl->K <= WIDE_MAX_K && cost_is_symmetric && precision != PILOT_OT_PREC_GENERIC && precision != PILOT_OT_PREC_F64 &&
            pl->max_cost / reg <= h_max_cost_over_reg() && tau <= pilot::H_MAX_TAU && !pilot::test_switch("PILOT_OT_NO_WIDE")) {
            if (!(f32_floor_ulps > 0.0)) f32_floor_ulps = 8.0;
            return run_wide(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, f32_floor_ulps, row_begin, n_rows_g, row_step, d_emd,
                            d_iters, d_err, d_flags, static_cast<hipStream_t>(stream));
        }
        if (precision == PILOT_OT_PREC_GENERIC || pl->K > MAX_K || pl->max_cost / reg > MAX_COST_OVER_REG)
            return run_generic(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, row_begin, n_rows_g, row_step, d_emd,
                               d_iters, d_err, d_flags, static_cast<hipStream_t>(stream));
    }
    // (the range is judged by the plan's max_cost / reg: 1 / reg for Trajectory.py:101's normalised cost unless the caller said
    // otherwise with pilot_ot_plan_set_max_cost; the host and multi-device entry points set it from the M they copy in)
    precision = pilot_ot_resolve_precision(precision, pl->max_cost / reg, pl->K, cost_is_symmetric, tau);
    bool mixed = false;
    if (precision == PILOT_OT_PREC_AUTO_MIXED) {
        precision = PILOT_OT_PREC_F64;
        mixed = true;
    }
    // beyond the f32 range AUTO still tries f32 first, pair by pair, where the split images fit and POT's defaults hold
    mixed = mixed && split_fits_lds(pl->K, cost_is_symmetric != 0, 2) && pl->max_cost / reg <= 140.0 && !pilot::test_switch("PILOT_OT_NO_MIXED");
    if (!(f32_floor_ulps > 0.0)) f32_floor_ulps = 8.0;
    const int n_rows = (row_end - row_begin + row_step - 1) / row_step;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cfg = mixed ? pilot::CFG_S32
                          : (precision == PILOT_OT_PREC_F32 ? pilot::CFG_F32
                             : (precision == PILOT_OT_PREC_BF16X3 ? pilot::CFG_S32 : (precision == PILOT_OT_PREC_F16X2 ? pilot::CFG_H32 : pilot::CFG_F64)));
    auto run = [&](hipStream_t on) {
        int r = run_grid(cfg, pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, f32_floor_ulps, cost_is_symmetric != 0,
                         row_begin, n_rows, row_step, d_emd, d_iters, d_err, d_flags, on, mixed);
        // a shape whose operand images do not fit LDS in this precision (non-symmetric cost at large K): the POT-literal
        // kernel takes the whole grid -- the reference has no such limit
        if (r == PILOT_OT_ENOTSUP)
            r = run_generic(pl, d_P, d_M, reg, num_iter_max, stop_thr, tau, check_period, row_begin, n_rows, row_step, d_emd, d_iters, d_err,
                            d_flags, on);
        return r;
    };
    if (!pl->graph_mode || pl->timing || n_rows == 0) return run(s);
    // graph replay: the first call with a new argument set runs as usual (and grows the work buffers), the second one is
    // captured, later ones replay the instantiated graph
    const char *dbg = pilot::test_switch("PILOT_OT_DEBUG");
    const pilot_ot_plan::GraphKey key = {d_P, d_M, d_emd, d_iters, d_err, d_flags, reg, stop_thr, tau, f32_floor_ulps, pl->max_cost, num_iter_max,
                                         check_period, cfg, mixed ? 1 : 0, cost_is_symmetric != 0 ? 1 : 0, row_begin, n_rows, row_step,
                                         dbg ? atoi(dbg) : 0};
    if (pl->gexec && key == pl->gkey) {
        HIP_TRY(hipGraphLaunch(pl->gexec, s));
        return PILOT_OT_OK;
    }
    if (pl->gexec) { (void)hipGraphExecDestroy(pl->gexec); pl->gexec = nullptr; }
    if (!(pl->gkey_seen && key == pl->gkey)) {
        pl->gkey = key; pl->gkey_seen = 1;
        return run(s);
    }
    if (!pl->gstream) HIP_TRY(hipStreamCreateWithFlags(&pl->gstream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(pl->gstream, hipStreamCaptureModeThreadLocal));
    rc = run(pl->gstream);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(pl->gstream, &graph);
    if (rc != PILOT_OT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) { (void)hipGetLastError(); return fail(PILOT_OT_EHIP, "graph capture failed: %s", hipGetErrorString(ce)); }
    const hipError_t ie = hipGraphInstantiate(&pl->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { (void)hipGetLastError(); pl->gexec = nullptr; return fail(PILOT_OT_EHIP, "graph instantiation failed: %s", hipGetErrorString(ie)); }
    HIP_TRY(hipGraphLaunch(pl->gexec, s));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_plan_enable_graph(pilot_ot_plan *pl, int enable) {
    if (!pl) return fail(PILOT_OT_EINVAL, "plan is NULL");
    pl->graph_mode = enable ? 1 : 0;
    if (!enable) {
        if (pl->gexec) { (void)hipGraphExecDestroy(pl->gexec); pl->gexec = nullptr; }
        pl->gkey_seen = 0;
    }
    return PILOT_OT_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-buffer entry points keep one plan + staging buffers per calling thread and reuse them while the
// shape stays the same (a PILOT session calls with one (N, K)); pilot_ot_shutdown() releases them.
namespace {
// Temporaries of the pre-pass host calls come from a per-thread pool that only grows (hipMalloc / hipFree cost about a
// millisecond a pair and hipFree synchronises the device: nine of them were most of a 22 ms medians call); released by
// pilot_ot_shutdown().  Slot i of the pool backs the i-th DevBuf a call declares.
struct WsPool {
    static constexpr int SLOTS = 20;
    void *p[SLOTS] = {};
    size_t cap[SLOTS] = {};
    int device = -1;
    void release() {
        for (int i = 0; i < SLOTS; ++i) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
        device = -1;
    }
    hipError_t get(int slot, size_t bytes, void **out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev != device) { release(); device = dev; }
        if (bytes > cap[slot]) {
            if (p[slot]) (void)hipFree(p[slot]);
            p[slot] = nullptr; cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            e = hipMalloc(&p[slot], want);
            if (e != hipSuccess) return e;
            cap[slot] = want;
        }
        *out = p[slot];
        return hipSuccess;
    }
};
struct HostCtx {
    pilot_ot_plan *plan = nullptr;
    int N = 0, K = 0, device = -1;
    size_t n_out = 0;
    double *dP = nullptr, *dM = nullptr, *dE = nullptr, *dErr = nullptr;
    int *dIt = nullptr, *dFl = nullptr;
    // pinned staging of the results: a D2H copy straight into the caller's pageable arrays makes the driver pin and
    // unpin them on every call (measured: 2 ms -> 25 ms per c3 matrix whenever numpy hands out fresh pages)
    unsigned char *pin = nullptr;
    size_t pin_bytes = 0;
    // events behind the pieces of a large fetch (host_fetch): the copy out of the pinned block starts when the first piece lands
    static constexpr int FETCH_EVENTS = 32;
    hipEvent_t fev[FETCH_EVENTS];
    int n_fev = 0;
    void release() {
        for (int i = 0; i < n_fev; ++i) (void)hipEventDestroy(fev[i]);
        n_fev = 0;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; pin_bytes = 0;
        if (dP) (void)hipFree(dP);
        if (dM) (void)hipFree(dM);
        if (dE) (void)hipFree(dE);
        if (dErr) (void)hipFree(dErr);
        if (dIt) (void)hipFree(dIt);
        if (dFl) (void)hipFree(dFl);
        dP = dM = dE = dErr = nullptr; dIt = dFl = nullptr;
        if (plan) pilot_ot_plan_destroy(plan);
        plan = nullptr; N = K = 0; n_out = 0; device = -1;
    }
    ~HostCtx() {}   // device memory is released by pilot_ot_shutdown() or at process exit
};
// Per-thread caches (host-entry workspace + pre-pass pool) live in a process-wide registry, not in thread_local objects:
// pilot_ot_shutdown() releases the caches of EVERY thread (it must not run concurrently with other calls), and the caches
// of a thread that has exited are released by the next thread that creates its own -- never from a thread-exit or
// process-exit hook, where the HIP runtime may already be gone.
// device time of the calling thread's last pre-pass (pilot_ot_prepass_device_ms): two events on the launch stream
struct PrepassClock {
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool valid = false;
    void release() { for (auto &e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; } valid = false; }
    void start() {
        valid = false;
        if (!ev[0] && (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess)) { release(); return; }
        (void)hipEventRecord(ev[0], nullptr);
    }
    void stop() { if (ev[1]) valid = hipEventRecord(ev[1], nullptr) == hipSuccess; }
};
struct ThreadCtx { HostCtx host; WsPool ws; PrepassClock clock; bool orphan = false; };
std::mutex g_tctx_mutex;
std::vector<ThreadCtx *> g_tctx_all;
struct TctxOwner {
    ThreadCtx *c = nullptr;
    ~TctxOwner() {
        if (!c) return;
        std::lock_guard<std::mutex> l(g_tctx_mutex);
        c->orphan = true;           // (no HIP calls here)
    }
};
thread_local TctxOwner g_tctx_owner;
ThreadCtx &tctx() {
    if (!g_tctx_owner.c) {
        std::lock_guard<std::mutex> l(g_tctx_mutex);
        for (size_t i = 0; i < g_tctx_all.size();) {
            if (g_tctx_all[i]->orphan) {
                g_tctx_all[i]->host.release();
                g_tctx_all[i]->ws.release();
                g_tctx_all[i]->clock.release();
                delete g_tctx_all[i];
                g_tctx_all.erase(g_tctx_all.begin() + (long)i);
            } else {
                ++i;
            }
        }
        g_tctx_owner.c = new ThreadCtx();
        g_tctx_all.push_back(g_tctx_owner.c);
    }
    return *g_tctx_owner.c;
}
#define g_host (tctx().host)
#define g_ws (tctx().ws)
#define g_clock (tctx().clock)
hipError_t ws_get(int slot, size_t bytes, void **out) { return g_ws.get(slot, bytes, out); }
}  // namespace
namespace pilot {
hipError_t ws_buffer(int slot, size_t bytes, void **out) { return ws_get(slot, bytes, out); }
}  // namespace pilot
namespace {

int host_ctx_prepare(int N, int K, size_t n_out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HostCtx &h = g_host;
    if (h.plan && (h.N != N || h.K != K || h.device != dev)) h.release();
    if (!h.plan) {
        int rc = pilot_ot_plan_create(N, K, &h.plan);
        if (rc != PILOT_OT_OK) return rc;
        h.N = N; h.K = K; h.device = dev;
        hipError_t e = hipMalloc(&h.dP, sizeof(double) * (size_t)N * K);
        if (e == hipSuccess) e = hipMalloc(&h.dM, sizeof(double) * (size_t)K * K);
        if (e != hipSuccess) { h.release(); return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e)); }
    }
    if (n_out > h.n_out) {
        if (h.dE) (void)hipFree(h.dE);
        if (h.dErr) (void)hipFree(h.dErr);
        if (h.dIt) (void)hipFree(h.dIt);
        if (h.dFl) (void)hipFree(h.dFl);
        h.dE = h.dErr = nullptr; h.dIt = h.dFl = nullptr; h.n_out = 0;
        hipError_t e = hipMalloc(&h.dE, sizeof(double) * n_out);
        if (e == hipSuccess) e = hipMalloc(&h.dErr, sizeof(double) * n_out);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h.dIt), sizeof(int) * n_out);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h.dFl), sizeof(int) * n_out);
        if (e != hipSuccess) { h.release(); return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e)); }
        h.n_out = n_out;
    }
    const size_t need = n_out * (2 * sizeof(double) + 2 * sizeof(int)) + 4 * 64;
    if (need > h.pin_bytes) {
        if (h.pin) (void)hipHostFree(h.pin);
        h.pin = nullptr; h.pin_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&h.pin), need, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h.pin = nullptr;               // no pinned memory: fall back to direct copies
        } else {
            h.pin_bytes = need;
        }
    }
    return PILOT_OT_OK;
}

// Copies out of the pinned block on a few persistent helper threads (host memory only: they make no HIP call).  One thread moves
// a 2.9 MB matrix into the caller's pageable array at 12 - 17 GB/s, 0.2 of the 0.95 ms of a c3 call through the host entry; a thread
// per call costs more than it saves (thread start + the runtime's per-thread set-up: measured, with a 7 ms outlier).  The pool is
// created by the first large fetch of the process, never destroyed (its threads sleep on a condition variable until the process
// ends), and forgotten by a forked child (pthread_atfork), which copies on its own thread.
struct CopyJob { void *dst; const void *src; size_t bytes; int *left; };      // left: jobs of the caller's batch still to finish (guarded by the pool's mutex)
class CopyPool {
public:
    static constexpr int HELPERS = 2;
    static CopyPool *get() {
        static std::once_flag once;
        std::call_once(once, [] {
            g_pool = new (std::nothrow) CopyPool();
            pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; });      // (the child has no helper threads)
        });
        return g_pool;
    }
    void push(void *dst, const void *src, size_t bytes, int *left) {       // never blocks
        { std::lock_guard<std::mutex> l(m_); ++*left; q_.push_back(CopyJob{dst, src, bytes, left}); }
        cv_.notify_one();
    }
    // the caller's thread copies too (any queued job, its own batch's or another caller's) until its own batch is done
    void finish(int *left) {
        std::unique_lock<std::mutex> l(m_);
        while (*left > 0) {
            if (!q_.empty()) run_one(l);
            else done_.wait(l);
        }
    }
    int helpers() const { return n_started_; }
private:
    CopyPool() {
        for (int t = 0; t < HELPERS; ++t) {
            try { std::thread([this] { loop(); }).detach(); ++n_started_; }
            catch (...) { break; }
        }
    }
    void run_one(std::unique_lock<std::mutex> &l) {
        const CopyJob j = q_.back(); q_.pop_back();
        l.unlock();
        memcpy(j.dst, j.src, j.bytes);
        l.lock();
        if (--*j.left == 0) done_.notify_all();
    }
    void loop() {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            cv_.wait(l, [this] { return !q_.empty(); });
            run_one(l);
        }
    }
    static CopyPool *g_pool;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<CopyJob> q_;
    int n_started_ = 0;
};
CopyPool *CopyPool::g_pool = nullptr;

// device results -> caller's arrays through the pinned staging block.  Small results: one stream sync, then the copies out of the
// block.  From 1 MB on (a 600 x 600 matrix is 2.9 MB) the transfer is cut into a few pieces with an event behind each, and every
// piece that has landed is copied out by the helper threads (and this one) while the next is in flight.
struct Fetch { void *dst; const void *src; size_t bytes; };
int host_fetch(const Fetch *f, int n) {
    HostCtx &h = g_host;
    if (!h.pin) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        for (int i = 0; i < n; ++i)
            if (f[i].dst) HIP_TRY(hipMemcpy(f[i].dst, f[i].src, f[i].bytes, hipMemcpyDeviceToHost));
        return PILOT_OT_OK;
    }
    size_t total = 0;
    for (int i = 0; i < n; ++i) if (f[i].dst) total += f[i].bytes;
    constexpr size_t PIPELINE_FROM = (size_t)1 << 20;
    int want_threads = 1 + CopyPool::HELPERS;
    if (const char *e = pilot::test_switch("PILOT_OT_FETCH_THREADS")) { const int w = atoi(e); if (w >= 1 && w < want_threads) want_threads = w; }   // experiment switch
    CopyPool *pool = (total >= PIPELINE_FROM && want_threads > 1) ? CopyPool::get() : nullptr;
    if (!pool || pool->helpers() == 0) {
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            if (!f[i].dst) continue;
            HIP_TRY(hipMemcpyAsync(h.pin + off, f[i].src, f[i].bytes, hipMemcpyDeviceToHost, nullptr));
            off += (f[i].bytes + 63) & ~(size_t)63;
        }
        HIP_TRY(hipStreamSynchronize(nullptr));
        off = 0;
        for (int i = 0; i < n; ++i) {
            if (!f[i].dst) continue;
            memcpy(f[i].dst, h.pin + off, f[i].bytes);
            off += (f[i].bytes + 63) & ~(size_t)63;
        }
        return PILOT_OT_OK;
    }
    while (h.n_fev < HostCtx::FETCH_EVENTS) {
        HIP_TRY(hipEventCreateWithFlags(&h.fev[h.n_fev], hipEventDisableTiming));
        ++h.n_fev;
    }
    struct Piece { unsigned char *dst; const unsigned char *pin; size_t bytes; };
    Piece pieces[HostCtx::FETCH_EVENTS];
    int n_pieces = 0;
    // pieces of >= 768 KB (every hipMemcpyAsync + hipEventRecord pair is ~10 us of this thread), at most as many as there are events
    size_t piece = (size_t)768 << 10;
    while ((total + piece - 1) / piece + (size_t)n > (size_t)HostCtx::FETCH_EVENTS) piece *= 2;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!f[i].dst) continue;
        for (size_t o = 0; o < f[i].bytes; o += piece) {
            const size_t b = f[i].bytes - o < piece ? f[i].bytes - o : piece;
            HIP_TRY(hipMemcpyAsync(h.pin + off + o, static_cast<const unsigned char *>(f[i].src) + o, b, hipMemcpyDeviceToHost, nullptr));
            HIP_TRY(hipEventRecord(h.fev[n_pieces], nullptr));
            pieces[n_pieces++] = Piece{static_cast<unsigned char *>(f[i].dst) + o, h.pin + off + o, b};
        }
        off += (f[i].bytes + 63) & ~(size_t)63;
    }
    hipError_t err = hipSuccess;
    int left = 0;
    for (int i = 0; i < n_pieces && err == hipSuccess; ++i) {
        err = hipEventSynchronize(h.fev[i]);
        if (err != hipSuccess) break;
        // a landed piece goes to the pool in `want_threads` parts (this thread takes its share in finish())
        const size_t part = ((pieces[i].bytes + want_threads - 1) / want_threads + 4095) & ~(size_t)4095;
        for (size_t o = 0; o < pieces[i].bytes; o += part)
            pool->push(pieces[i].dst + o, pieces[i].pin + o, pieces[i].bytes - o < part ? pieces[i].bytes - o : part, &left);
    }
    pool->finish(&left);
    if (err != hipSuccess) return pilot::abi_fail(PILOT_OT_EHIP, "fetching the results failed: %s", hipGetErrorString(err));
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_shutdown(void) {
    {
        std::lock_guard<std::mutex> l(g_tctx_mutex);
        for (ThreadCtx *c : g_tctx_all) { c->host.release(); c->ws.release(); c->clock.release(); }
    }
    pilot::abi_multi_release();
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_sinkhorn_grid(const double *P, int N, int K, const double *M, double reg, int num_iter_max,
                                     double stop_thr, double tau, int check_period, int precision,
                                     double f32_floor_ulps, int cost_is_symmetric, int row_begin, int row_end,
                                     int row_step, double *emd, int *iters, double *err, int *flags) {
    if (!P || !M || !emd) return fail(PILOT_OT_EINVAL, "NULL pointer");
    int rc = check_grid_args(N, K, reg, num_iter_max, stop_thr, tau, check_period, precision, row_begin, row_end,
                             row_step);
    if (rc != PILOT_OT_OK) return rc;
    double mx = 0.0;
    for (size_t t = 0; t < (size_t)K * K; ++t) mx = M[t] > mx ? M[t] : mx;
    precision = pilot_ot_resolve_precision(precision, mx / reg, K, cost_is_symmetric, tau);     // (max(M) is known here)
    const int n_rows = (row_end - row_begin + row_step - 1) / row_step;
    const size_t n_out = (size_t)n_rows * N;
    if (n_out == 0) return PILOT_OT_OK;

    const bool trace = pilot::test_switch("PILOT_OT_HOST_TRACE") != nullptr;       // (stage stamps on stderr: tools/host_to_host_probe.py)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = now();
    rc = host_ctx_prepare(N, K, n_out);
    if (rc != PILOT_OT_OK) return rc;
    HostCtx &h = g_host;
    hipError_t e = hipMemcpy(h.dP, P, sizeof(double) * (size_t)N * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h.dM, M, sizeof(double) * (size_t)K * K, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "H2D copy failed: %s", hipGetErrorString(e));
    const auto t1 = now();
    h.plan->max_cost = mx > 0.0 ? mx : 1.0;
    rc = pilot_ot_sinkhorn_grid_dev(h.plan, h.dP, h.dM, reg, num_iter_max, stop_thr, tau, check_period, precision,
                                    f32_floor_ulps, cost_is_symmetric, row_begin, row_end, row_step, h.dE,
                                    iters ? h.dIt : nullptr, err ? h.dErr : nullptr, h.dFl, nullptr);
    if (rc != PILOT_OT_OK) return rc;
    const auto t2 = now();
    if (trace) (void)hipStreamSynchronize(nullptr);
    const auto t3 = now();
    const Fetch f[4] = {{emd, h.dE, sizeof(double) * n_out}, {iters, h.dIt, sizeof(int) * n_out},
                        {err, h.dErr, sizeof(double) * n_out}, {flags, h.dFl, sizeof(int) * n_out}};
    rc = host_fetch(f, 4);
    if (trace) fprintf(stderr, "pilot_ot_sinkhorn_grid: prepare + H2D %.0f us, enqueue %.0f us, device %.0f us, fetch %.0f us\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, now()));
    return rc;
}

// ------------------------------------------------------------------------------------------------
PILOT_API int pilot_ot_emd_grid_dev(pilot_ot_plan *pl, const double *d_P, const double *d_M, int mode, int row_begin,
                                    int row_end, int row_step, double *d_emd, int *d_n_aug, void *stream) {
    if (!pl || !d_P || !d_M || !d_emd) return fail(PILOT_OT_EINVAL, "NULL pointer");
    const int N = pl->N, K = pl->K;
    if (mode < PILOT_OT_EMD_ALL || mode > PILOT_OT_EMD_MIRROR) return fail(PILOT_OT_EINVAL, "unknown mode %d", mode);
    if (row_step < 1 || row_begin < 0 || row_end > N || row_begin > row_end)
        return fail(PILOT_OT_EINVAL, "bad row range [%d, %d) step %d for N=%d", row_begin, row_end, row_step, N);
    if (mode == PILOT_OT_EMD_MIRROR && !(row_begin == 0 && row_end == N && row_step == 1))
        return fail(PILOT_OT_EINVAL, "PILOT_OT_EMD_MIRROR needs the full square grid");
    const int n_rows = (row_end - row_begin + row_step - 1) / row_step;
    if (n_rows == 0) return PILOT_OT_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    pilot::EmdParams p;
    p.P = d_P; p.M = d_M; p.N = N; p.K = K;
    p.n_rows = n_rows; p.row_begin = row_begin; p.row_step = row_step;
    p.upper_only = mode != PILOT_OT_EMD_ALL;
    p.emd = d_emd; p.n_aug = d_n_aug; p.f_slab = nullptr; p.queue = pl->emd_counter;
    // the flow slab of the wave-per-pair kernels: allocated ONCE, by the plan's first exact call that needs it, at the largest
    // size any kernel variant of this K asks for (several pairs per wave with the flows in the slab; one pair per wave) -- never
    // freed or regrown by a later call (a call made under stream capture after the first one allocates nothing)
    auto need_slab = [&](size_t bytes) -> int {
        if (!pl->f_slab) {
            size_t most = bytes;
            if (K <= EMD_MAX_K) {
                const size_t one = sizeof(double) * (size_t)K * K * emd_wgs_per_cu(K) * pl->n_cu * pilot::emd_waves(emd_nk(K));
                most = one > most ? one : most;
            }
            if (K <= EMD_MULTI_MAX_K) {
                const pilot::EmdMultiGeom g = pilot::emd_multi_geom(K, false);
                const size_t multi = sizeof(double) * (size_t)K * K * 4 * g.waves * g.wgs_per_cu * pl->n_cu;
                most = multi > most ? multi : most;
            }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->f_slab), most));
            pl->f_slab_bytes = most;
        }
        if (pl->f_slab_bytes < bytes) return fail(PILOT_OT_EHIP, "exact OT: the plan's flow slab (%zu bytes) is smaller than this call needs (%zu)", pl->f_slab_bytes, bytes);
        p.f_slab = pl->f_slab;
        return PILOT_OT_OK;
    };
    HIP_TRY(hipMemsetAsync(pl->emd_counter, 0, sizeof(int) * pilot::EMD_NQ * pilot::EMD_Q_STRIDE, s));
    const long total = (long)n_rows * N;
    if (K > EMD_MAX_K) {
        // beyond the one-wave-per-pair kernel: one workgroup per pair, vectors in LDS, flows in a global slab per resident
        // workgroup (emd_generic_kernel.hpp) -- the reference has no limit on the number of cell types
        if (K > pilot::EMDG_MAX_K) return fail(PILOT_OT_ENOTSUP, "exact OT: K=%d > %d cell types", K, pilot::EMDG_MAX_K);
        const size_t per_wg = sizeof(double) * pilot::emdg_slab_doubles(K);
        const long n_items = p.upper_only ? (long)n_rows * (N - row_begin) - (long)row_step * n_rows * (n_rows - 1) / 2 : total;
        long wgs = 2L * pl->n_cu;
        while (wgs > 1 && per_wg * (size_t)wgs > ((size_t)8 << 30)) wgs /= 2;
        if (!pl->emdg_slab || pl->emdg_wgs < wgs) {          // (first call that needs this kernel: the one allocation of the path)
            if (pl->emdg_slab) HIP_TRY(hipFree(pl->emdg_slab));
            pl->emdg_slab = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->emdg_slab), per_wg * (size_t)wgs + sizeof(double) * (size_t)K));
            pl->emdg_wgs = (int)wgs;
        }
        if (wgs > n_items) wgs = n_items > 0 ? n_items : 1;
        double *rowmin = pl->emdg_slab + pilot::emdg_slab_doubles(K) * (size_t)pl->emdg_wgs;
        hipLaunchKernelGGL(pilot::emd_rowmin_kernel, dim3((K + 255) / 256), dim3(256), 0, s, d_M, K, rowmin);
        p.f_slab = pl->emdg_slab;
        const size_t lds = pilot::emdg_lds_bytes(K);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::emd_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(pilot::emd_generic_kernel, dim3((unsigned)wgs), dim3(pilot::EMDG_WG), lds, s, p, rowmin);
    } else if (K <= EMD_MULTI_MAX_K && emd_multi_mode(K) != 0) {
        // four pairs per wavefront
        const pilot::EmdMultiGeom m = pilot::emd_multi_geom(K, emd_multi_mode(K) != 2);
        if (!m.flds) {      // (flow values in the global slab: a K x K block per 16-lane group of every resident wave)
            const int rc = need_slab(sizeof(double) * (size_t)K * K * 4 * m.waves * m.wgs_per_cu * pl->n_cu);
            if (rc != PILOT_OT_OK) return rc;
        }
        const long groups = (total + (64 / m.G) - 1) / (64 / m.G);
        long wgs = (groups + m.waves - 1) / m.waves;
        const long cap = (long)pl->n_cu * m.wgs_per_cu;
        if (wgs > cap) wgs = cap;
        auto kern = m.flds ? pilot::emd_multi_kernel<16, true> : pilot::emd_multi_kernel<16, false>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m.lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * m.waves), m.lds, s, p);
    } else {
        {
            const int rc = need_slab(sizeof(double) * (size_t)K * K * emd_wgs_per_cu(K) * pl->n_cu * pilot::emd_waves(emd_nk(K)));
            if (rc != PILOT_OT_OK) return rc;
        }
        const size_t lds = pilot::emd_lds_bytes(K);
        if (lds > LDS_BYTES) return fail(PILOT_OT_ENOTSUP, "K=%d does not fit the LDS layout", K);
        const int waves = pilot::emd_waves(emd_nk(K));
        long wgs = (total + waves - 1) / waves;
        const long cap = (long)pl->n_cu * emd_wgs_per_cu(K);
        if (wgs > cap) wgs = cap;
        constexpr bool UL = pilot::emd_ul(128);      // (labels without the column potential: always beyond 64 cell types)
        if (K > 192) {
            hipLaunchKernelGGL((pilot::emd_grid_kernel<4, true, UL>), dim3((unsigned)wgs), dim3(64 * waves), lds, s, p);
        } else if (K > 128) {
            hipLaunchKernelGGL((pilot::emd_grid_kernel<3, true, UL>), dim3((unsigned)wgs), dim3(64 * waves), lds, s, p);
        } else if (K <= 64) {
            auto kern = pilot::emd_ul(K) ? pilot::emd_grid_kernel<1, false, true> : pilot::emd_grid_kernel<1, false, false>;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * waves), lds, s, p);
        } else {
            auto kern = pilot::emd_grid_kernel<2, false, UL>;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * waves), lds, s, p);
        }
    }
    HIP_TRY(hipGetLastError());
    if (mode == PILOT_OT_EMD_MIRROR) {
        hipLaunchKernelGGL(pilot::emd_mirror_kernel, dim3(1024), dim3(256), 0, s, d_emd, N);
        HIP_TRY(hipGetLastError());
    }
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_mirror_upper_dev(double *d_emd, int N, void *stream) {
    if (!d_emd || N <= 0) return fail(PILOT_OT_EINVAL, "bad argument");
    hipLaunchKernelGGL(pilot::emd_mirror_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), d_emd, N);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_emd_grid(const double *P, int N, int K, const double *M, int mode, int row_begin, int row_end,
                                int row_step, double *emd, int *n_aug) {
    if (!P || !M || !emd) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || K <= 0) return fail(PILOT_OT_EINVAL, "N=%d K=%d must be positive", N, K);
    if (row_step < 1 || row_begin < 0 || row_end > N || row_begin > row_end)
        return fail(PILOT_OT_EINVAL, "bad row range [%d, %d) step %d for N=%d", row_begin, row_end, row_step, N);
    const int n_rows = (row_end - row_begin + row_step - 1) / row_step;
    const size_t n_out = (size_t)n_rows * N;
    if (n_out == 0) return PILOT_OT_OK;
    int rc = host_ctx_prepare(N, K, n_out);
    if (rc != PILOT_OT_OK) return rc;
    HostCtx &h = g_host;
    hipError_t e = hipMemcpy(h.dP, P, sizeof(double) * (size_t)N * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h.dM, M, sizeof(double) * (size_t)K * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(h.dE, 0, sizeof(double) * n_out);
    if (e == hipSuccess) e = hipMemset(h.dIt, 0, sizeof(int) * n_out);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    rc = pilot_ot_emd_grid_dev(h.plan, h.dP, h.dM, mode, row_begin, row_end, row_step, h.dE, h.dIt, nullptr);
    if (rc != PILOT_OT_OK) return rc;
    const Fetch f[2] = {{emd, h.dE, sizeof(double) * n_out}, {n_aug, h.dIt, sizeof(int) * n_out}};
    return host_fetch(f, 2);
}

// ------------------------------------------------------------------------------------------------
PILOT_API int pilot_ot_plan_enable_timing(pilot_ot_plan *pl, int enable) {
    if (!pl) return fail(PILOT_OT_EINVAL, "plan is NULL");
    if (enable)
        for (int i = 0; i < TIMING_RING; ++i)
            for (int j = 0; j < 4; ++j)
                if (!pl->ev[i][j]) HIP_TRY(hipEventCreate(&pl->ev[i][j]));
    pl->timing = enable > 0 ? enable : 0;       // n > 1: every n-th call is timed (four event records cost a 0.7 ms call 2 %)
    pl->n_timed = 0; pl->n_calls = 0;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_plan_kernel_times(pilot_ot_plan *pl, int max_n, float *main_ms, float *track_ms, int *n_out) {
    if (!pl || !main_ms || !track_ms || !n_out) return fail(PILOT_OT_EINVAL, "NULL pointer");
    long n = pl->n_timed < TIMING_RING ? pl->n_timed : TIMING_RING;
    if (n > max_n) n = max_n;
    for (long t = 0; t < n; ++t) {
        const long call = pl->n_timed - n + t;
        hipEvent_t *ev = pl->ev[call % TIMING_RING];
        HIP_TRY(hipEventElapsedTime(&main_ms[t], ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&track_ms[t], ev[2], ev[3]));
    }
    *n_out = (int)n;
    return PILOT_OT_OK;
}

// ------------------------------------------------------------------------------------------------
// pre-pass (host-buffer entry points; the inputs are read once, so they are staged per call)
namespace {
struct DevBuf {
    void *p = nullptr;
    int slot;
    explicit DevBuf(int slot_) : slot(slot_) {}
    hipError_t alloc(size_t bytes) { return g_ws.get(slot, bytes ? bytes : 1, &p); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};
int grid_for(long n, int block, int n_cu) {
    long g = (n + block - 1) / block;
    const long cap = (long)n_cu * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
int current_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}
}  // namespace

namespace {
// ---- the device pre-pass, carved out of ONE pooled workspace (slot 1 of the calling thread's pool) ----------------------
// head (cleared by one memset): counts N*K | first_row N | n_k K | cursor K | n_items 1 | global histograms K*D*2*256
// then: prior K | P N*K | segment starts K | select items | select state | centroids K*D | codes | grouped keys
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
struct PrepassWs {
    // what the caller asks for
    bool want_counts = false, want_medians = false;
    long long C = 0; int N = 0, K = 0, D = 0, n_cu = 256; size_t key_bytes = 4;
    int n_code_cols = 1;
    // derived
    long R = 0, max_items = 0;
    size_t o_counts = 0, o_first = 0, o_nk = 0, o_cursor = 0, o_nitems = 0, o_hist = 0, clear_bytes = 0, o_prior = 0, o_P = 0, o_offs = 0,
           o_items = 0, o_st = 0, o_out = 0, o_code = 0, o_y = 0, total = 0;
    void carve() {
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
        o_counts = take(want_counts ? sizeof(unsigned int) * (size_t)N * K : 0);
        o_first = take(want_counts ? sizeof(unsigned int) * (size_t)N : 0);
        o_nk = take(sizeof(unsigned int) * K);
        o_cursor = take(sizeof(unsigned int) * K);
        o_nitems = take(sizeof(unsigned int));
        if (want_medians) {
            // rows per select item (the unit the passes are balanced in; a block takes a run of items): 256, more only to
            // keep the list below 64 K items; a multiple of 4
            R = 256;
            if (C / R > 65536) R = (long)(((C / 65536) + 3) & ~3LL);
            max_items = (long)(C / R) + K + 1;
            o_hist = take(sizeof(unsigned int) * (size_t)K * D * 2 * 256);
        } else {
            o_hist = o;
        }
        clear_bytes = o;
        o_prior = take(want_counts ? sizeof(double) * (size_t)K : 0);
        o_P = take(want_counts ? sizeof(double) * (size_t)N * K : 0);
        o_offs = take(sizeof(unsigned int) * K);
        o_items = take(want_medians ? sizeof(pilot::SelectItem) * (size_t)max_items : 0);
        o_st = take(want_medians ? (key_bytes + 8) * (size_t)K * D * 2 : 0);
        o_out = take(want_medians ? sizeof(double) * (size_t)K * D : 0);
        o_code = take(sizeof(int) * (size_t)C * n_code_cols);
        o_y = take(want_medians ? key_bytes * ((size_t)C + 4 * (size_t)K) * D : 0);
        total = o;
    }
};

// counts + first rows + n_k from device-resident codes, then the proportions: three launches
void launch_counts(const PrepassWs &ws, unsigned char *w, const int *d_cell, const int *d_sample, long long n_total, double regulizer,
                   int normalization, bool want_first) {
    const long nchunks = (long)((ws.C + pilot::COUNT_CHUNK - 1) / pilot::COUNT_CHUNK);
    long grid = 4L * ws.n_cu;
    if (grid > nchunks) grid = nchunks;
    if (grid < 1) grid = 1;
    unsigned int *counts = reinterpret_cast<unsigned int *>(w + ws.o_counts);
    const size_t lds = sizeof(unsigned int) * ((size_t)pilot::COUNT_LDS_BINS + pilot::COUNT_LDS_ROWS + ws.K + 16);
    hipLaunchKernelGGL(pilot::count_kernel, dim3((unsigned)grid), dim3(256), lds, nullptr, d_cell, d_sample, (long)ws.C, ws.N, ws.K, counts,
                       reinterpret_cast<unsigned int *>(w + ws.o_nk), want_first ? reinterpret_cast<unsigned int *>(w + ws.o_first) : nullptr);
    hipLaunchKernelGGL(pilot::prior_kernel, dim3((unsigned)((ws.K + 3) / 4)), dim3(256), 0, nullptr, counts, ws.N, ws.K, (long)n_total, regulizer,
                       reinterpret_cast<double *>(w + ws.o_prior));
    hipLaunchKernelGGL(pilot::proportions_kernel, dim3((unsigned)((ws.N + 3) / 4)), dim3(256), 0, nullptr, counts, ws.N, ws.K,
                       reinterpret_cast<const double *>(w + ws.o_prior), normalization, reinterpret_cast<double *>(w + ws.o_P));
}

// the general median path (prepass_kernels.hpp): [count,] prep, group the rows by type, BITS/8 x (histogram, pick)
template <typename T>
int launch_medians(const PrepassWs &ws, unsigned char *w, const T *dXp, const int *d_cell, bool have_nk) {
    using U = typename pilot::OrderedKey<T>::U;
    using State = pilot::SelectState<U>;
    static_assert(sizeof(State) <= sizeof(U) + 8, "select state larger than its carve");
    const int K = ws.K, D = ws.D;
    const long long C = ws.C;
    unsigned int *d_nk = reinterpret_cast<unsigned int *>(w + ws.o_nk), *d_cursor = reinterpret_cast<unsigned int *>(w + ws.o_cursor),
                 *d_nitems = reinterpret_cast<unsigned int *>(w + ws.o_nitems), *d_hist = reinterpret_cast<unsigned int *>(w + ws.o_hist),
                 *d_offs = reinterpret_cast<unsigned int *>(w + ws.o_offs);
    pilot::SelectItem *d_items = reinterpret_cast<pilot::SelectItem *>(w + ws.o_items);
    State *d_st = reinterpret_cast<State *>(w + ws.o_st);
    double *d_out = reinterpret_cast<double *>(w + ws.o_out);
    U *d_y = reinterpret_cast<U *>(w + ws.o_y);
    const long nb = (long)((C + pilot::GROUP_ROWS_PER_BLOCK - 1) / pilot::GROUP_ROWS_PER_BLOCK);
    if (!have_nk) {
        long g = 2L * ws.n_cu;
        if (g > nb) g = nb;
        hipLaunchKernelGGL(pilot::type_count_kernel, dim3((unsigned)g), dim3(256), sizeof(unsigned int) * K, nullptr, d_cell, (long)C, K, d_nk);
    }
    hipLaunchKernelGGL(pilot::median_prep_kernel, dim3(1), dim3(256), sizeof(unsigned int) * (2 * (size_t)K + 2 + 257), nullptr, d_nk, K,
                       (unsigned int)ws.R, d_offs, d_nitems, d_items);
    hipLaunchKernelGGL(pilot::group_rows_kernel<T>, dim3((unsigned)nb), dim3(256),
                       sizeof(unsigned int) * (2 * (size_t)K + pilot::GROUP_ROWS_PER_BLOCK), nullptr, dXp, D, d_cell, (long)C, K, d_offs, d_cursor, d_y);
    const int Dw_max = D < pilot::SELECT_MAX_DIMS ? D : pilot::SELECT_MAX_DIMS;      // dimensions per histogram launch
    const size_t lds = sizeof(U) * 2 * (size_t)Dw_max + sizeof(unsigned int) * (size_t)Dw_max * 2 * 256;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::select_hist_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // as many histogram blocks as the chip holds at once (LDS-bound), each with an equal run of the work list
    long hist_grid = (long)ws.n_cu * (long)((160 * 1024) / (lds + 256) < (size_t)(2048 / pilot::SELECT_THREADS) ? (160 * 1024) / (lds + 256)
                                                                                                                 : (size_t)(2048 / pilot::SELECT_THREADS));
    if (hist_grid > ws.max_items) hist_grid = ws.max_items;
    if (hist_grid < 1) hist_grid = 1;
    const unsigned pick_blocks = (unsigned)(((size_t)K * D * 64 + 255) / 256);
    for (int shift = pilot::OrderedKey<T>::BITS - 8; shift >= 0; shift -= 8) {
        for (int dbeg = 0; dbeg < D; dbeg += Dw_max) {           // any D: the dimensions in windows that fit the LDS histograms
            const int Dw = D - dbeg < Dw_max ? D - dbeg : Dw_max;
            hipLaunchKernelGGL(pilot::select_hist_kernel<T>, dim3((unsigned)hist_grid), dim3(pilot::SELECT_THREADS),
                               sizeof(U) * 2 * (size_t)Dw + sizeof(unsigned int) * (size_t)Dw * 2 * 256, nullptr, d_y, D, dbeg, Dw,
                               d_nitems, d_items, shift, d_st, d_hist);
        }
        hipLaunchKernelGGL(pilot::select_pick_kernel<T>, dim3(pick_blocks), dim3(256), 0, nullptr, d_nk, K, D, shift, d_st, d_hist, d_out);
    }
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

// small cohorts: one launch, the selection in LDS (small_medians_kernel) -- when every type fits its key buffer and the
// K x D workgroups reading all C codes is a small amount of traffic (PILOT_OT_NO_SMALL_MEDIANS=1: the general path, tests)
bool small_medians_fit(long long C, int D, const int *cell_code, int K, unsigned int *n_max_out) {
    if (!(C > 0 && (double)C * K * D <= 3.2e7) || pilot::test_switch("PILOT_OT_NO_SMALL_MEDIANS")) return false;
    std::vector<unsigned int> n_k((size_t)K, 0u);
    for (long long c = 0; c < C; ++c) { const int k = cell_code[c]; if (k >= 0 && k < K) ++n_k[(size_t)k]; }
    unsigned int n_max = 0;
    for (unsigned int v : n_k) n_max = v > n_max ? v : n_max;
    *n_max_out = n_max;
    return n_max <= (unsigned int)pilot::SMALL_MEDIANS_CAP;
}
template <typename T>
int launch_small_medians(const T *dXp, long long C, int D, const int *d_cell, int K, unsigned int n_max, double *d_out) {
    using U = typename pilot::OrderedKey<T>::U;
    const size_t lds = sizeof(U) * (size_t)(n_max ? n_max : 1);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::small_medians_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(U) * pilot::SMALL_MEDIANS_CAP)));
    hipLaunchKernelGGL(pilot::small_medians_kernel<T>, dim3((unsigned)(K * D)), dim3(256), lds, nullptr, dXp, D, d_cell, (long)C, K, d_out);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_proportions(const int *cell_code, const int *sample_code, long long n_cells, long long n_total,
                                   int N, int K, double regulizer, int normalization, double *P) {
    return pilot_ot_proportions_ex(cell_code, sample_code, n_cells, n_total, N, K, regulizer, normalization, P, nullptr);
}

PILOT_API int pilot_ot_proportions_ex(const int *cell_code, const int *sample_code, long long n_cells, long long n_total,
                                      int N, int K, double regulizer, int normalization, double *P, long long *first_row) {
    if (!cell_code || !sample_code || !P) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (first_row && n_cells > 0xfffffffeLL) return fail(PILOT_OT_ENOTSUP, "n_cells=%lld exceeds the 32-bit row index", n_cells);
    if (N <= 0 || K <= 0 || n_cells < 0 || n_total < 2)
        return fail(PILOT_OT_EINVAL, "N=%d K=%d n_cells=%lld n_total=%lld out of range", N, K, n_cells, n_total);
    if (K > 4096) return fail(PILOT_OT_ENOTSUP, "K=%d > 4096 cell types", K);
    PrepassWs ws;
    ws.want_counts = true; ws.C = n_cells; ws.N = N; ws.K = K; ws.n_cu = current_cu_count(); ws.n_code_cols = 2;
    ws.carve();
    DevBuf buf(1);
    hipError_t e = buf.alloc(ws.total);
    unsigned char *w = buf.as<unsigned char>();
    int *d_cell = reinterpret_cast<int *>(w + ws.o_code), *d_sample = d_cell + n_cells;
    if (e == hipSuccess) e = hipMemcpy(d_cell, cell_code, sizeof(int) * (size_t)n_cells, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sample, sample_code, sizeof(int) * (size_t)n_cells, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(w, 0, ws.clear_bytes, nullptr);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    launch_counts(ws, w, d_cell, d_sample, n_total, regulizer, normalization, first_row != nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(P, w + ws.o_P, sizeof(double) * (size_t)N * K, hipMemcpyDeviceToHost));
    if (first_row) {
        std::vector<unsigned int> fr((size_t)N);
        HIP_TRY(hipMemcpy(fr.data(), w + ws.o_first, sizeof(unsigned int) * (size_t)N, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) first_row[n] = fr[(size_t)n] == 0u ? -1 : (long long)(0xffffffffu - fr[(size_t)n]);
    }
    return PILOT_OT_OK;
}

// the embedding resident on the device: uploaded once (from a helper thread of the host language, beside its own work on
// the label columns), read by pilot_ot_centroid_medians_dev / pilot_ot_prepass_dev
struct pilot_ot_embedding {
    void *dX = nullptr;
    int dtype = 0, D = 0, device = 0;
    long long C = 0;
};

PILOT_API int pilot_ot_embedding_upload(const void *X, int dtype, long long n_cells, int D, pilot_ot_embedding **emb) {
    if (!X || !emb) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (n_cells <= 0 || D <= 0) return fail(PILOT_OT_EINVAL, "n_cells=%lld D=%d must be positive", n_cells, D);
    if (dtype != PILOT_OT_F32 && dtype != PILOT_OT_F64) return fail(PILOT_OT_EINVAL, "unknown dtype id %d", dtype);
    pilot_ot_embedding *e = new (std::nothrow) pilot_ot_embedding();
    if (!e) return fail(PILOT_OT_EINVAL, "out of host memory");
    e->dtype = dtype; e->D = D; e->C = n_cells;
    const size_t bytes = (size_t)n_cells * D * (dtype == PILOT_OT_F32 ? 4 : 8);
    hipError_t he = hipGetDevice(&e->device);
    if (he == hipSuccess) he = hipMalloc(&e->dX, bytes);
    if (he == hipSuccess) he = hipMemcpy(e->dX, X, bytes, hipMemcpyHostToDevice);
    if (he != hipSuccess) { pilot_ot_embedding_destroy(e); return fail(PILOT_OT_EHIP, "embedding upload failed: %s", hipGetErrorString(he)); }
    *emb = e;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_embedding_destroy(pilot_ot_embedding *e) {
    if (!e) return PILOT_OT_OK;
    if (e->dX) (void)hipFree(e->dX);
    delete e;
    return PILOT_OT_OK;
}

namespace {
// dXdev (nullable): the embedding already on the device; else X is copied in
template <typename T>
int centroid_medians_impl(const void *X, const void *dXdev, long long C, int D, const int *cell_code, int K, double *centroids) {
    unsigned int n_max = 0;
    const bool small = small_medians_fit(C, D, cell_code, K, &n_max);
    PrepassWs ws;
    ws.want_medians = !small; ws.C = C; ws.K = K; ws.D = D; ws.n_cu = current_cu_count(); ws.key_bytes = sizeof(T);
    ws.carve();
    DevBuf dX(0), buf(1);
    hipError_t e = dXdev ? hipSuccess : dX.alloc(sizeof(T) * (size_t)C * D);
    if (e == hipSuccess) e = buf.alloc(ws.total + al256(sizeof(double) * (size_t)K * D));
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    unsigned char *w = buf.as<unsigned char>();
    int *d_cell = reinterpret_cast<int *>(w + ws.o_code);
    double *d_out = small ? reinterpret_cast<double *>(w + ws.total) : reinterpret_cast<double *>(w + ws.o_out);
    if (!dXdev) e = hipMemcpy(dX.p, X, sizeof(T) * (size_t)C * D, hipMemcpyHostToDevice);
    const T *dXp = static_cast<const T *>(dXdev ? dXdev : dX.p);
    if (e == hipSuccess) e = hipMemcpy(d_cell, cell_code, sizeof(int) * (size_t)C, hipMemcpyHostToDevice);
    if (e == hipSuccess && !small) e = hipMemsetAsync(w, 0, ws.clear_bytes, nullptr);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    g_clock.start();
    const int rc = small ? launch_small_medians<T>(dXp, C, D, d_cell, K, n_max, d_out) : launch_medians<T>(ws, w, dXp, d_cell, false);
    g_clock.stop();
    if (rc != PILOT_OT_OK) return rc;
    HIP_TRY(hipMemcpy(centroids, d_out, sizeof(double) * (size_t)K * D, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}

// the whole pre-pass from one upload of the two code columns
template <typename T>
int prepass_impl(const pilot_ot_embedding *emb, const int *cell_code, const int *sample_code, long long n_total, int N, int K,
                 double regulizer, int normalization, double *P, long long *first_row, double *centroids) {
    const long long C = emb->C;
    const int D = emb->D;
    unsigned int n_max = 0;
    const bool small = small_medians_fit(C, D, cell_code, K, &n_max);
    PrepassWs ws;
    ws.want_counts = true; ws.want_medians = !small; ws.C = C; ws.N = N; ws.K = K; ws.D = D; ws.n_cu = current_cu_count();
    ws.key_bytes = sizeof(T); ws.n_code_cols = 2;
    ws.carve();
    DevBuf buf(1);
    // results leave in one copy: P | centroids | first rows, packed behind the workspace
    const size_t r_P = sizeof(double) * (size_t)N * K, r_cen = sizeof(double) * (size_t)K * D, r_first = sizeof(unsigned int) * (size_t)N;
    hipError_t e = buf.alloc(ws.total + al256(r_P + r_cen + r_first));
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    unsigned char *w = buf.as<unsigned char>(), *res = w + ws.total;
    int *d_cell = reinterpret_cast<int *>(w + ws.o_code), *d_sample = d_cell + C;
    e = hipMemcpy(d_cell, cell_code, sizeof(int) * (size_t)C, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sample, sample_code, sizeof(int) * (size_t)C, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(w, 0, ws.clear_bytes, nullptr);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    g_clock.start();
    launch_counts(ws, w, d_cell, d_sample, n_total, regulizer, normalization, true);
    const T *dXp = static_cast<const T *>(emb->dX);
    double *d_cen = reinterpret_cast<double *>(res + r_P);
    int rc = small ? launch_small_medians<T>(dXp, C, D, d_cell, K, n_max, d_cen) : launch_medians<T>(ws, w, dXp, d_cell, true);
    g_clock.stop();
    if (rc != PILOT_OT_OK) return rc;
    hipError_t he = hipMemcpyAsync(res, w + ws.o_P, r_P, hipMemcpyDeviceToDevice, nullptr);
    if (he == hipSuccess && !small) he = hipMemcpyAsync(d_cen, w + ws.o_out, r_cen, hipMemcpyDeviceToDevice, nullptr);
    if (he == hipSuccess) he = hipMemcpyAsync(res + r_P + r_cen, w + ws.o_first, r_first, hipMemcpyDeviceToDevice, nullptr);
    std::vector<unsigned char> host(r_P + r_cen + r_first);
    if (he == hipSuccess) he = hipMemcpy(host.data(), res, host.size(), hipMemcpyDeviceToHost);
    if (he != hipSuccess) return fail(PILOT_OT_EHIP, "pre-pass results: %s", hipGetErrorString(he));
    memcpy(P, host.data(), r_P);
    memcpy(centroids, host.data() + r_P, r_cen);
    const unsigned int *fr = reinterpret_cast<const unsigned int *>(host.data() + r_P + r_cen);
    if (first_row) for (int n = 0; n < N; ++n) first_row[n] = fr[n] == 0u ? -1 : (long long)(0xffffffffu - fr[n]);
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_centroid_medians(const void *X, int dtype, long long n_cells, int D, const int *cell_code, int K,
                                        double *centroids) {
    if (!X || !cell_code || !centroids) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (n_cells <= 0 || D <= 0 || K <= 0) return fail(PILOT_OT_EINVAL, "n_cells=%lld D=%d K=%d must be positive", n_cells, D, K);
    if (n_cells > 0xfffffffeLL) return fail(PILOT_OT_ENOTSUP, "n_cells=%lld exceeds the 32-bit row index", n_cells);
    if (K > 4096) return fail(PILOT_OT_ENOTSUP, "K=%d > 4096 cell types", K);
    if (dtype == PILOT_OT_F32) return centroid_medians_impl<float>(X, nullptr, n_cells, D, cell_code, K, centroids);
    if (dtype == PILOT_OT_F64) return centroid_medians_impl<double>(X, nullptr, n_cells, D, cell_code, K, centroids);
    return fail(PILOT_OT_EINVAL, "unknown dtype id %d", dtype);
}

PILOT_API int pilot_ot_centroid_medians_dev(pilot_ot_embedding *e, const int *cell_code, int K, double *centroids) {
    if (!e || !cell_code || !centroids) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (K <= 0) return fail(PILOT_OT_EINVAL, "K=%d must be positive", K);
    if (K > 4096) return fail(PILOT_OT_ENOTSUP, "K=%d > 4096 cell types", K);
    if (e->C > 0xfffffffeLL) return fail(PILOT_OT_ENOTSUP, "n_cells=%lld exceeds the 32-bit row index", e->C);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != e->device) return fail(PILOT_OT_EINVAL, "the embedding lives on device %d, the current device is %d", e->device, dev);
    if (e->dtype == PILOT_OT_F32) return centroid_medians_impl<float>(nullptr, e->dX, e->C, e->D, cell_code, K, centroids);
    return centroid_medians_impl<double>(nullptr, e->dX, e->C, e->D, cell_code, K, centroids);
}

PILOT_API int pilot_ot_prepass_device_ms(float *ms) {
    if (!ms) return fail(PILOT_OT_EINVAL, "NULL pointer");
    PrepassClock &c = g_clock;
    if (!c.valid) return fail(PILOT_OT_EINVAL, "no pre-pass has run on this thread");
    HIP_TRY(hipEventSynchronize(c.ev[1]));
    HIP_TRY(hipEventElapsedTime(ms, c.ev[0], c.ev[1]));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_prepass_dev(pilot_ot_embedding *e, const int *cell_code, const int *sample_code, long long n_total, int N, int K,
                                   double regulizer, int normalization, double *P, long long *first_row, double *centroids) {
    if (!e || !cell_code || !sample_code || !P || !centroids) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || K <= 0 || n_total < 2) return fail(PILOT_OT_EINVAL, "N=%d K=%d n_total=%lld out of range", N, K, n_total);
    if (K > 4096) return fail(PILOT_OT_ENOTSUP, "K=%d > 4096 cell types", K);
    if (e->C > 0xfffffffeLL) return fail(PILOT_OT_ENOTSUP, "n_cells=%lld exceeds the 32-bit row index", e->C);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != e->device) return fail(PILOT_OT_EINVAL, "the embedding lives on device %d, the current device is %d", e->device, dev);
    if (e->dtype == PILOT_OT_F32) return prepass_impl<float>(e, cell_code, sample_code, n_total, N, K, regulizer, normalization, P, first_row, centroids);
    return prepass_impl<double>(e, cell_code, sample_code, n_total, N, K, regulizer, normalization, P, first_row, centroids);
}

// ------------------------------------------------------------------------------------------------
// cell-level W2 (extension)
struct pilot_ot_cell_cohort {
    int N = 0, D = 0, KB = 1, device = 0, n_cu = 256;
    long long C = 0, max_n = 0;
    float *dX = nullptr;           // the cells as given (resident: the operand pieces are rebuilt when scale * reg changes)
    float xb_scale = 0.f;          // operand scale the pieces were built with (0: not built)
    int xb_half = -1;              // ... and their format: 1 two fp16 pieces, 0 three bf16 pieces
    int xb_one_slot = -1;          // ... and whether the last k-slot of every cell holds 1 (cell_setup_kernel)
    float max_abs = 0.f;           // largest |coordinate| of the centred cohort (decides whether fp16 pieces are safe)
    uint4 *dXb = nullptr;          // bf16 operand pieces of every cell (resident)
    float *dnrm = nullptr;
    long long *doffs = nullptr;
    // per-call outputs / queue, grown on demand
    double *dW = nullptr, *dErr = nullptr;
    int *dIt = nullptr, *dQ = nullptr;
    size_t n_out = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

PILOT_API int pilot_ot_cell_cohort_destroy(pilot_ot_cell_cohort *c) {
    if (!c) return PILOT_OT_OK;
    for (void *p : {(void *)c->dX, (void *)c->dXb, (void *)c->dnrm, (void *)c->doffs, (void *)c->dW, (void *)c->dErr, (void *)c->dIt, (void *)c->dQ})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_cell_cohort_create(const float *X, const long long *offsets, int N, int D, pilot_ot_cell_cohort **cohort) {
    if (!X || !offsets || !cohort) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || D <= 0) return fail(PILOT_OT_EINVAL, "N=%d D=%d must be positive", N, D);
    if (D > 64) return fail(PILOT_OT_ENOTSUP, "D=%d > 64 embedding dimensions", D);
    long long max_n = 0;
    for (int i = 0; i < N; ++i) {
        const long long n = offsets[i + 1] - offsets[i];
        if (n <= 0) return fail(PILOT_OT_EINVAL, "patient %d has %lld cells", i, n);
        if (n > max_n) max_n = n;
    }
    const size_t lds = sizeof(float) * (3 * (size_t)max_n + 48);
    if (lds > LDS_BYTES) return fail(PILOT_OT_ENOTSUP, "a patient with %lld cells needs %zu B of LDS (> %zu)", max_n, lds, LDS_BYTES);
    pilot_ot_cell_cohort *c = new (std::nothrow) pilot_ot_cell_cohort();
    if (!c) return fail(PILOT_OT_EINVAL, "out of host memory");
    c->N = N; c->D = D; c->KB = D <= 32 ? 1 : 2; c->C = offsets[N]; c->max_n = max_n;
    c->n_cu = current_cu_count();
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dX), sizeof(float) * (size_t)c->C * D);
    // (+ a zeroed pad: the pipelined column sweep of the fp16-piece kernel reads up to 31 cells past a patient's last one,
    // pilot::CELL_PAD_BYTES in cellw2_kernels.hpp)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dXb), (size_t)c->C * c->KB * 3 * 64 + pilot::CELL_PAD_BYTES);
    if (e == hipSuccess) e = hipMemset(c->dXb, 0, (size_t)c->C * c->KB * 3 * 64 + pilot::CELL_PAD_BYTES);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dnrm), sizeof(float) * (size_t)c->C);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->doffs), sizeof(long long) * (size_t)(N + 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dQ), sizeof(int));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) {
        // |x - y|^2 does not change when every cell is shifted by the same vector, but the f32 cancellation in
        // |x|^2 + |y|^2 - 2 <x, y> does: the cohort is stored centred on its mean (computed in fp64)
        std::vector<double> mean((size_t)D, 0.0);
        for (long long i = 0; i < c->C; ++i)
            for (int d = 0; d < D; ++d) mean[(size_t)d] += (double)X[(size_t)i * D + d];
        for (int d = 0; d < D; ++d) mean[(size_t)d] /= (double)c->C;
        std::vector<float> Xc((size_t)c->C * D);
        float mx = 0.f;
        for (long long i = 0; i < c->C; ++i)
            for (int d = 0; d < D; ++d) {
                const float v = (float)((double)X[(size_t)i * D + d] - mean[(size_t)d]);
                Xc[(size_t)i * D + d] = v;
                mx = fabsf(v) > mx ? fabsf(v) : mx;
            }
        c->max_abs = mx;
        e = hipMemcpy(c->dX, Xc.data(), sizeof(float) * (size_t)c->C * D, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(c->doffs, offsets, sizeof(long long) * (size_t)(N + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) { pilot_ot_cell_cohort_destroy(c); return fail(PILOT_OT_EHIP, "cell cohort setup failed: %s", hipGetErrorString(e)); }
    *cohort = c;
    return PILOT_OT_OK;
}

namespace {
// enqueue one pass over the selected rows on the cohort's stream (asynchronous)
int cell_w2_enqueue(pilot_ot_cell_cohort *c, double scale, double reg, int num_iter_max, double stop_thr, int check_period,
                    double f32_floor_ulps, int row_begin, int row_end, int row_step, size_t *n_out_p) {
    if (!c) return fail(PILOT_OT_EINVAL, "cohort is NULL");
    if (!(scale > 0.0) || !(reg > 0.0)) return fail(PILOT_OT_EINVAL, "scale=%g reg=%g must be positive", scale, reg);
    if (num_iter_max < 1 || check_period < 1) return fail(PILOT_OT_EINVAL, "num_iter_max / check_period must be >= 1");
    if (row_step < 1 || row_begin < 0 || row_end > c->N || row_begin > row_end)
        return fail(PILOT_OT_EINVAL, "bad row range [%d, %d) step %d for N=%d", row_begin, row_end, row_step, c->N);
    const int n_rows = (row_end - row_begin + row_step - 1) / row_step;
    const size_t n_out = (size_t)n_rows * c->N;
    *n_out_p = n_out;
    if (n_out == 0) return PILOT_OT_OK;
    if (!(f32_floor_ulps > 0.0)) f32_floor_ulps = 8.0;
    if (n_out > c->n_out) {
        for (void *p : {(void *)c->dW, (void *)c->dErr, (void *)c->dIt}) if (p) (void)hipFree(p);
        c->dW = c->dErr = nullptr; c->dIt = nullptr; c->n_out = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->dW), sizeof(double) * n_out);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dErr), sizeof(double) * n_out);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->dIt), sizeof(int) * n_out);
        if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
        c->n_out = n_out;
    }
    HIP_TRY(hipMemsetAsync(c->dQ, 0, sizeof(int), c->stream));
    pilot::CellParams p;
    p.Xb = c->dXb; p.C = c->C; p.nrm = c->dnrm; p.offs = c->doffs; p.N = c->N;
    p.n_rows = n_rows; p.row_begin = row_begin; p.row_step = row_step;
    int half = 0;
    const double alpha = 1.0 / (scale * reg);
    p.alpha = (float)alpha;
    p.two_alpha2 = (float)(2.0 * alpha * 1.4426950408889634);
    {
        // operand pieces of sqrt(2 alpha log2 e) * x: a dot product of two operands is the exponent term itself
        const float op_scale = sqrtf(p.two_alpha2);
        p.two_alpha2 = op_scale * op_scale;
        p.dot_unscale = 1.f / p.two_alpha2;
        // two fp16 pieces (half the matrix work) while the scaled coordinates stay far inside fp16's range and above the
        // level where its subnormal spacing (2^-24) would cost accuracy; three bf16 pieces otherwise (PILOT_OT_CELL_BF16=1: always)
        const char *force = pilot::test_switch("PILOT_OT_CELL_BF16");
        half = c->max_abs * op_scale < 3.0e4f && !(force && *force && *force != '0') ? 1 : 0;
        const int one_slot = half && c->D <= 32 * c->KB - 2 && !pilot::test_switch("PILOT_OT_CELL_NO_AUG") ? 1 : 0;
        if (c->xb_scale != op_scale || c->xb_half != half || c->xb_one_slot != one_slot) {
            if (c->xb_half != half) HIP_TRY(hipMemsetAsync(c->dXb, 0, (size_t)c->C * c->KB * 3 * 64 + pilot::CELL_PAD_BYTES, c->stream));   // (the piece count changes the planes)
            hipLaunchKernelGGL(pilot::cell_setup_kernel, dim3(grid_for(c->C * c->KB * 32, 256, c->n_cu)), dim3(256), 0, c->stream, c->dX,
                               (long)c->C, c->D, c->KB, op_scale, half, one_slot, reinterpret_cast<unsigned short *>(c->dXb), c->dnrm);
            HIP_TRY(hipGetLastError());
            c->xb_scale = op_scale;
            c->xb_half = half;
            c->xb_one_slot = one_slot;
        }
    }
    p.inv_scale = (float)(1.0 / scale);
    p.max_iter = num_iter_max; p.period = check_period;
    p.stop_thr = (float)stop_thr; p.floor_ulps = (float)f32_floor_ulps;
    p.max_n = (int)c->max_n;
    p.w2 = c->dW; p.iters = c->dIt; p.err = c->dErr; p.queue = c->dQ;
    const size_t lds = sizeof(float) * (3 * (size_t)c->max_n + 48);
    long wgs = (long)n_out;
    long per_cu = (long)(LDS_BYTES / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    if (wgs > c->n_cu * per_cu) wgs = c->n_cu * per_cu;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    hipError_t le = hipSuccess;
    const bool aug = c->D <= 32 * c->KB - 2 && !pilot::test_switch("PILOT_OT_CELL_NO_AUG");      // two spare k-slots carry h_col - m_row
    auto launch = [&](auto kern) {
        le = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (le == hipSuccess) hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(pilot::CELL_WG), lds, c->stream, p);
    };
    if (half) {
        if (c->KB == 1) { if (aug) launch(pilot::cell_w2_kernel<1, true, true>); else launch(pilot::cell_w2_kernel<1, false, true>); }
        else            { if (aug) launch(pilot::cell_w2_kernel<2, true, true>); else launch(pilot::cell_w2_kernel<2, false, true>); }
    } else {
        if (c->KB == 1) { if (aug) launch(pilot::cell_w2_kernel<1, true>); else launch(pilot::cell_w2_kernel<1, false>); }
        else            { if (aug) launch(pilot::cell_w2_kernel<2, true>); else launch(pilot::cell_w2_kernel<2, false>); }
    }
    HIP_TRY(le);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return PILOT_OT_OK;
}
int cell_w2_collect(pilot_ot_cell_cohort *c, size_t n_out, double *w2, int *iters, double *err, float *kernel_ms) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_out == 0) return PILOT_OT_OK;
    if (w2) HIP_TRY(hipMemcpy(w2, c->dW, sizeof(double) * n_out, hipMemcpyDeviceToHost));
    if (iters) HIP_TRY(hipMemcpy(iters, c->dIt, sizeof(int) * n_out, hipMemcpyDeviceToHost));
    if (err) HIP_TRY(hipMemcpy(err, c->dErr, sizeof(double) * n_out, hipMemcpyDeviceToHost));
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, c->ev0, c->ev1));
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_cell_w2_grid_cohort(pilot_ot_cell_cohort *c, double scale, double reg, int num_iter_max, double stop_thr,
                                           int check_period, double f32_floor_ulps, int row_begin, int row_end, int row_step,
                                           double *w2, int *iters, double *err, float *kernel_ms) {
    if (!c || !w2) return fail(PILOT_OT_EINVAL, "NULL pointer");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != c->device) HIP_TRY(hipSetDevice(c->device));
    size_t n_out = 0;
    int rc = cell_w2_enqueue(c, scale, reg, num_iter_max, stop_thr, check_period, f32_floor_ulps, row_begin, row_end, row_step, &n_out);
    if (rc == PILOT_OT_OK) rc = cell_w2_collect(c, n_out, w2, iters, err, kernel_ms);
    if (dev != c->device) (void)hipSetDevice(dev);
    return rc;
}

PILOT_API int pilot_ot_cell_cohort_pieces(pilot_ot_cell_cohort *c, int *pieces) {
    if (!c || !pieces) return fail(PILOT_OT_EINVAL, "NULL pointer");
    *pieces = c->xb_half < 0 ? 0 : (c->xb_half ? 2 : 3);      // operand pieces of the last call: 2 fp16, 3 bf16, 0 none yet
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_cell_w2_grid(const float *X, const long long *offsets, int N, int D, double scale, double reg,
                                    int num_iter_max, double stop_thr, int check_period, double f32_floor_ulps,
                                    int row_begin, int row_end, int row_step, double *w2, int *iters, double *err) {
    if (!X || !offsets || !w2) return fail(PILOT_OT_EINVAL, "NULL pointer");
    pilot_ot_cell_cohort *c = nullptr;
    int rc = pilot_ot_cell_cohort_create(X, offsets, N, D, &c);
    if (rc != PILOT_OT_OK) return rc;
    rc = pilot_ot_cell_w2_grid_cohort(c, scale, reg, num_iter_max, stop_thr, check_period, f32_floor_ulps, row_begin, row_end,
                                      row_step, w2, iters, err, nullptr);
    pilot_ot_cell_cohort_destroy(c);
    return rc;
}

// internal face of the cohort for the multi-device form (pilot_ot_multi.hip: row shards + device-side all-gather)
namespace pilot {
int cell_enqueue_rows(pilot_ot_cell_cohort *c, double scale, double reg, int num_iter_max, double stop_thr, int check_period,
                      double f32_floor_ulps, int row_begin, int row_end, int row_step, size_t *n_out) {
    return cell_w2_enqueue(c, scale, reg, num_iter_max, stop_thr, check_period, f32_floor_ulps, row_begin, row_end, row_step, n_out);
}
int cell_collect(pilot_ot_cell_cohort *c, size_t n_out, double *w2, int *iters, double *err, float *kernel_ms) {
    return cell_w2_collect(c, n_out, w2, iters, err, kernel_ms);
}
void cell_buffers(pilot_ot_cell_cohort *c, double **d_w2, hipStream_t *stream) { *d_w2 = c->dW; *stream = c->stream; }
}  // namespace pilot
