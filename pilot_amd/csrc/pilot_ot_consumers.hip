// C ABI of the downstream consumers of the distance matrix (include/pilot_ot.h, section "consumers"; kernels and the
// reference call sites they replace: consumer_kernels.hpp).  Host-buffer entry points + device-resident forms.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "abi_common.hpp"
#include "consumer_kernels.hpp"

#define fail(...) pilot::abi_fail(__VA_ARGS__)

namespace {
struct DevMem {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    ~DevMem() { if (p) (void)hipFree(p); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};
}  // namespace

PILOT_API int pilot_ot_row_distances_dev(const double *d_E, int N, int normalize_by_max, int metric, double *d_D,
                                         double *d_max_scratch, void *stream) {
    if (!d_E || !d_D) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    if (metric != PILOT_OT_ROWMETRIC_EUCLIDEAN && metric != PILOT_OT_ROWMETRIC_COSINE) return fail(PILOT_OT_EINVAL, "unknown row metric %d", metric);
    if (normalize_by_max && !d_max_scratch) return fail(PILOT_OT_EINVAL, "normalize_by_max needs an 8-byte device scratch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (normalize_by_max) hipLaunchKernelGGL(pilot::max_reduce_kernel, dim3(1), dim3(256), 0, s, d_E, (long)N * N, d_max_scratch);
    const unsigned g = (unsigned)((N + pilot::RD_TILE - 1) / pilot::RD_TILE);
    hipLaunchKernelGGL(pilot::row_distance_kernel, dim3(g, g), dim3(pilot::RD_TILE * pilot::RD_TILE), 0, s, d_E, N, metric,
                       normalize_by_max ? d_max_scratch : nullptr, d_D);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_row_distances(const double *E, int N, int normalize_by_max, int metric, double *D) {
    if (!E || !D) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    DevMem dE, dD, dM;
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dE.alloc(bytes);
    if (e == hipSuccess) e = dD.alloc(bytes);
    if (e == hipSuccess) e = dM.alloc(8);
    if (e == hipSuccess) e = hipMemcpy(dE.p, E, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    int rc = pilot_ot_row_distances_dev(dE.as<double>(), N, normalize_by_max, metric, dD.as<double>(), dM.as<double>(), nullptr);
    if (rc != PILOT_OT_OK) return rc;
    HIP_TRY(hipMemcpy(D, dD.p, bytes, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_silhouette(const double *D, const int *labels, int N, int n_clusters, double *score, double *samples) {
    if (!D || !labels || !score) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || n_clusters <= 0 || n_clusters > 4096) return fail(PILOT_OT_EINVAL, "N=%d n_clusters=%d out of range", N, n_clusters);
    std::vector<int> sizes(n_clusters, 0);
    for (int i = 0; i < N; ++i) {
        if (labels[i] < 0 || labels[i] >= n_clusters) return fail(PILOT_OT_EINVAL, "label %d of sample %d outside [0, %d)", labels[i], i, n_clusters);
        ++sizes[labels[i]];
    }
    int used = 0;
    for (int c = 0; c < n_clusters; ++c) used += sizes[c] > 0;
    if (used < 2 || used > N - 1)       // sklearn: "Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)"
        return fail(PILOT_OT_EINVAL, "silhouette needs 2 .. N-1 distinct labels, got %d", used);
    DevMem dD, dL, dS, dO;
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = dL.alloc(sizeof(int) * N);
    if (e == hipSuccess) e = dS.alloc(sizeof(int) * n_clusters);
    if (e == hipSuccess) e = dO.alloc(sizeof(double) * N);
    if (e == hipSuccess) e = hipMemcpy(dD.p, D, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dL.p, labels, sizeof(int) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dS.p, sizes.data(), sizeof(int) * n_clusters, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(pilot::silhouette_kernel, dim3(N), dim3(256), sizeof(double) * n_clusters, nullptr, dD.as<double>(), dL.as<int>(),
                       dS.as<int>(), N, n_clusters, dO.as<double>());
    HIP_TRY(hipGetLastError());
    std::vector<double> s(N);
    HIP_TRY(hipMemcpy(s.data(), dO.p, sizeof(double) * N, hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (int i = 0; i < N; ++i) sum += s[i];          // np.mean order
    *score = sum / N;
    if (samples) for (int i = 0; i < N; ++i) samples[i] = s[i];
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_knn_kernel(const double *D, int N, int k, double epsilon, double *Kmat) {
    if (!D || !Kmat) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || k < 1 || !(epsilon > 0.0)) return fail(PILOT_OT_EINVAL, "N=%d k=%d epsilon=%g out of range", N, k, epsilon);
    if (k > N) k = N;
    int np2 = 1;
    while (np2 < N) np2 <<= 1;
    if (sizeof(double) * (size_t)np2 > 160 * 1024) return fail(PILOT_OT_ENOTSUP, "N=%d rows do not fit the LDS sort", N);
    DevMem dD, dK;
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = dK.alloc(bytes);
    if (e == hipSuccess) e = hipMemcpy(dD.p, D, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::knn_kernel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * np2)));
    hipLaunchKernelGGL(pilot::knn_kernel_kernel, dim3(N), dim3(256), sizeof(double) * np2, nullptr, dD.as<double>(), N, np2, k, epsilon,
                       dK.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(Kmat, dK.p, bytes, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}
