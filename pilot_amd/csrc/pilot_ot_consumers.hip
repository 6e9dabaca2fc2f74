// C ABI of the downstream consumers of the distance matrix (include/pilot_ot.h, section "consumers"; kernels and the
// reference call sites they replace: consumer_kernels.hpp).  Device-resident forms (`_dev`: device pointers + a stream, no
// allocation, no synchronisation), fused host chains (the matrix goes up once, only the result comes back) and the plain
// host-buffer forms on top of them.
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "abi_common.hpp"
#include "consumer_kernels.hpp"

#define fail(...) pilot::abi_fail(__VA_ARGS__)

namespace {
// temporaries of the host entry points: slots 12 .. 19 of the calling thread's pool (no hipMalloc / hipFree per call)
struct DevMem {
    void *p = nullptr;
    int slot;
    explicit DevMem(int slot_) : slot(slot_) {}
    hipError_t alloc(size_t bytes) { return pilot::ws_buffer(slot, bytes ? bytes : 1, &p); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};
int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
}  // namespace

PILOT_API int pilot_ot_row_distances_dev(const double *d_E, int N, int normalize_by_max, int metric, double *d_D,
                                         double *d_max_scratch, void *stream) {
    if (!d_E || !d_D) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    if (metric != PILOT_OT_ROWMETRIC_EUCLIDEAN && metric != PILOT_OT_ROWMETRIC_COSINE) return fail(PILOT_OT_EINVAL, "unknown row metric %d", metric);
    if (normalize_by_max && !d_max_scratch) return fail(PILOT_OT_EINVAL, "normalize_by_max needs an 8-byte device scratch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (normalize_by_max) {
        HIP_TRY(hipMemsetAsync(d_max_scratch, 0, sizeof(double), s));
        long blocks = ((long)N * N + 256 * 8 - 1) / (256 * 8);
        blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
        hipLaunchKernelGGL(pilot::max_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_E, (long)N * N, d_max_scratch);
    }
    const unsigned g = (unsigned)((N + pilot::RD_TILE - 1) / pilot::RD_TILE);
    hipLaunchKernelGGL(pilot::row_distance_kernel, dim3(g, g), dim3(pilot::RD_T * pilot::RD_T), 0, s, d_E, N, metric,
                       normalize_by_max ? d_max_scratch : nullptr, d_D);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_silhouette_dev(const double *d_D, const int *d_labels, int N, int n_clusters, int *d_sizes_scratch,
                                      double *d_samples, void *stream) {
    if (!d_D || !d_labels || !d_sizes_scratch || !d_samples) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || n_clusters <= 0 || n_clusters > 4096) return fail(PILOT_OT_EINVAL, "N=%d n_clusters=%d out of range", N, n_clusters);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the row and the labels are staged in LDS while they fit; beyond (N > ~12 700) the kernel reads them from global memory
    // in the same order (round 3 returned ENOTSUP there: ADVICE r03)
    size_t lds = sizeof(double) * ((size_t)N + n_clusters) + sizeof(int) * (size_t)N;
    const char *force = pilot::test_switch("PILOT_OT_SIL_UNSTAGED");       // (tests: the large-N form on a small matrix)
    const int staged = (lds <= 150 * 1024 && !(force && *force && *force != '0')) ? 1 : 0;
    if (!staged) lds = sizeof(double) * (size_t)n_clusters;
    hipLaunchKernelGGL(pilot::label_sizes_kernel, dim3(1), dim3(256), 0, s, d_labels, N, n_clusters, d_sizes_scratch);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::silhouette_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pilot::silhouette_kernel, dim3(N), dim3(256), lds, s, d_D, d_labels, d_sizes_scratch, N, n_clusters, staged, d_samples);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

PILOT_API int pilot_ot_knn_kernel_dev(const double *d_D, int N, int k, double epsilon, double *d_Kmat, void *stream) {
    if (!d_D || !d_Kmat) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || k < 1 || !(epsilon > 0.0)) return fail(PILOT_OT_EINVAL, "N=%d k=%d epsilon=%g out of range", N, k, epsilon);
    if (k > N) k = N;
    const int np2 = next_pow2(N);
    if (sizeof(double) * (size_t)np2 > 150 * 1024) return fail(PILOT_OT_ENOTSUP, "N=%d rows do not fit the LDS sort", N);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pilot::knn_kernel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * np2)));
    hipLaunchKernelGGL(pilot::knn_kernel_kernel, dim3(N), dim3(256), sizeof(double) * np2, static_cast<hipStream_t>(stream), d_D, N, np2, k,
                       epsilon, d_Kmat);
    HIP_TRY(hipGetLastError());
    return PILOT_OT_OK;
}

namespace {
// labels -> validated cluster count the way scikit-learn checks it
int check_labels(const int *labels, int N, int n_clusters) {
    std::vector<int> sizes(n_clusters, 0);
    for (int i = 0; i < N; ++i) {
        if (labels[i] < 0 || labels[i] >= n_clusters) return fail(PILOT_OT_EINVAL, "label %d of sample %d outside [0, %d)", labels[i], i, n_clusters);
        ++sizes[labels[i]];
    }
    int used = 0;
    for (int c = 0; c < n_clusters; ++c) used += sizes[c] > 0;
    if (used < 2 || used > N - 1)       // sklearn: "Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)"
        return fail(PILOT_OT_EINVAL, "silhouette needs 2 .. N-1 distinct labels, got %d", used);
    return PILOT_OT_OK;
}

// the silhouette of a labelling on a distance matrix that is ALREADY on the device: labels up, N scores down
int silhouette_of_device_matrix(const double *d_D, const int *labels, int N, int n_clusters, double *score, double *samples, hipStream_t st) {
    int rc = check_labels(labels, N, n_clusters);
    if (rc != PILOT_OT_OK) return rc;
    DevMem dL(15), dS(16), dO(17);
    hipError_t e = dL.alloc(sizeof(int) * N);
    if (e == hipSuccess) e = dS.alloc(sizeof(int) * n_clusters);
    if (e == hipSuccess) e = dO.alloc(sizeof(double) * N);
    if (e == hipSuccess) e = hipMemcpyAsync(dL.p, labels, sizeof(int) * N, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    rc = pilot_ot_silhouette_dev(d_D, dL.as<int>(), N, n_clusters, dS.as<int>(), dO.as<double>(), st);
    if (rc != PILOT_OT_OK) return rc;
    std::vector<double> s(N);
    HIP_TRY(hipMemcpyAsync(s.data(), dO.p, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double sum = 0.0;
    for (int i = 0; i < N; ++i) sum += s[i];          // np.mean order
    *score = sum / N;
    if (samples) for (int i = 0; i < N; ++i) samples[i] = s[i];
    return PILOT_OT_OK;
}
}  // namespace

PILOT_API int pilot_ot_row_distances(const double *E, int N, int normalize_by_max, int metric, double *D) {
    if (!E || !D) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    DevMem dE(12), dD(13), dM(14);
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
    int rc = check_labels(labels, N, n_clusters);
    if (rc != PILOT_OT_OK) return rc;
    DevMem dD(13);
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = hipMemcpy(dD.p, D, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    return silhouette_of_device_matrix(dD.as<double>(), labels, N, n_clusters, score, samples, nullptr);
}

PILOT_API int pilot_ot_knn_kernel(const double *D, int N, int k, double epsilon, double *Kmat) {
    if (!D || !Kmat) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    DevMem dD(13), dK(18);
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = dK.alloc(bytes);
    if (e == hipSuccess) e = hipMemcpy(dD.p, D, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    int rc = pilot_ot_knn_kernel_dev(dD.as<double>(), N, k, epsilon, dK.as<double>(), nullptr);
    if (rc != PILOT_OT_OK) return rc;
    HIP_TRY(hipMemcpy(Kmat, dK.p, bytes, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}

// ---- fused chains: the matrix goes to the device ONCE (or is there already), the row distances never leave it --------------
// Sil_computing (pilotpy/tools/Trajectory.py:592-612 on EMD / EMD.max(), ploting.py:324): silhouette of `labels` with the rows of
// E as the points.  E_is_device: E is a device pointer (the pair grid's output, e.g. pilot_ot_multi_device_matrix).
PILOT_API int pilot_ot_silhouette_of_rows(const double *E, int E_is_device, int N, int normalize_by_max, int metric, const int *labels,
                                          int n_clusters, double *score, double *samples) {
    if (!E || !labels || !score) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0 || n_clusters <= 0 || n_clusters > 4096) return fail(PILOT_OT_EINVAL, "N=%d n_clusters=%d out of range", N, n_clusters);
    DevMem dE(12), dD(13), dM(14);
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = dM.alloc(8);
    if (e == hipSuccess && !E_is_device) { e = dE.alloc(bytes); if (e == hipSuccess) e = hipMemcpy(dE.p, E, bytes, hipMemcpyHostToDevice); }
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    int rc = pilot_ot_row_distances_dev(E_is_device ? E : dE.as<double>(), N, normalize_by_max, metric, dD.as<double>(), dM.as<double>(), nullptr);
    if (rc != PILOT_OT_OK) return rc;
    return silhouette_of_device_matrix(dD.as<double>(), labels, N, n_clusters, score, samples, nullptr);
}

// the dense part of pl.trajectory (pilotpy/plot/ploting.py:95-110): E / max(E) -> Euclidean row distances -> pydiffmap's k-nearest-
// neighbour Gaussian kernel.  D_out (nullable) receives the row distances as well.
PILOT_API int pilot_ot_diffusion_kernel_of_rows(const double *E, int E_is_device, int N, int k, double epsilon, double *D_out, double *Kmat) {
    if (!E || !Kmat) return fail(PILOT_OT_EINVAL, "NULL pointer");
    if (N <= 0) return fail(PILOT_OT_EINVAL, "N=%d must be positive", N);
    DevMem dE(12), dD(13), dK(18), dM(14);
    const size_t bytes = sizeof(double) * (size_t)N * N;
    hipError_t e = dD.alloc(bytes);
    if (e == hipSuccess) e = dK.alloc(bytes);
    if (e == hipSuccess) e = dM.alloc(8);
    if (e == hipSuccess && !E_is_device) { e = dE.alloc(bytes); if (e == hipSuccess) e = hipMemcpy(dE.p, E, bytes, hipMemcpyHostToDevice); }
    if (e != hipSuccess) return fail(PILOT_OT_EHIP, "device staging failed: %s", hipGetErrorString(e));
    int rc = pilot_ot_row_distances_dev(E_is_device ? E : dE.as<double>(), N, 1, PILOT_OT_ROWMETRIC_EUCLIDEAN, dD.as<double>(), dM.as<double>(), nullptr);
    if (rc == PILOT_OT_OK) rc = pilot_ot_knn_kernel_dev(dD.as<double>(), N, k, epsilon, dK.as<double>(), nullptr);
    if (rc != PILOT_OT_OK) return rc;
    if (D_out) HIP_TRY(hipMemcpy(D_out, dD.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(Kmat, dK.p, bytes, hipMemcpyDeviceToHost));
    return PILOT_OT_OK;
}
