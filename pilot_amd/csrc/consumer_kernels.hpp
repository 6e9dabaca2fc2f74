// Downstream consumers of the finished N x N distance matrix, kept on the device.
//
// What the reference does with adata.uns['EMD'] right after wasserstein_distance:
//   * pl.trajectory         (pilotpy/plot/ploting.py:95-110): EMD / EMD.max(), then pydiffmap's DiffusionMap.from_sklearn(
//                           epsilon, alpha, k).fit_transform(EMD): the ROWS of the normalised matrix are the data points;
//                           kernel = exp(-d^2 / (4 epsilon)) on every row's k nearest rows (Euclidean d);
//   * Sil_computing         (pilotpy/tools/Trajectory.py:592-612, called from ploting.py:324 on EMD / EMD.max()) and
//     pl.select_best_sil    (ploting.py:425-431): sklearn.metrics.silhouette_score(EMD, labels, metric='cosine'): again the
//                           rows are the points, under the cosine (default) or Euclidean metric.
// Three kernels cover the dense part of both: row-to-row distances (an N x N x N contraction), the mean silhouette of a
// labelling, and the k-nearest-neighbour Gaussian kernel matrix.  All fp64 like scikit-learn.  The contraction is 2.2e8
// multiply-adds at N = 600 and 8e9 at N = 2000 -- a millisecond or two on the vector units beside a 38 ms pair grid, so it is a
// plain LDS-tiled kernel, not an MFMA one.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

// max of a buffer of non-negative doubles -> out[0] (preset to 0): the bit patterns of non-negative doubles order like
// unsigned integers, so the per-workgroup maxima meet in one atomicMax on the 64-bit pattern.  (A distance matrix; a
// negative entry can only lose against the preset 0, like in max(E) of a matrix with a zero diagonal.)
static __global__ void max_reduce_kernel(const double *__restrict__ X, long n, double *__restrict__ out) {
    __shared__ double part[256];
    double m = 0.0;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) m = X[t] > m ? X[t] : m;
    part[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] = part[threadIdx.x] > part[threadIdx.x + s] ? part[threadIdx.x] : part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(part[0]));
}

// D[i][j] = distance between rows i and j of X (N x N) * scale, scale = 1 / *inv_scale_src when given (EMD / EMD.max()).
// metric 0: Euclidean, accumulated as sum (x_ik - x_jk)^2 (no Gram cancellation, matches scipy cdist);
// metric 1: cosine, 1 - x.y / (|x| |y|), clipped to [0, 2], exact 0 on the diagonal (sklearn cosine_distances).
// An N x N x N contraction in fp64 (8e9 multiply-adds at N = 2000; the f64 matrix pipe has the vector pipe's rate, so this is a
// vector kernel): one 64 x 64 output tile per 256-thread workgroup, 4 x 4 outputs per thread (rows ti + 16 a, columns
// tj + 16 b: conflict-free LDS reads, 8 operands for 16 multiply-adds), k swept through LDS in chunks of 16.  Every output is
// accumulated over k in index order, whatever the tiling.
constexpr int RD_TILE = 64, RD_KC = 16, RD_T = 16, RD_R = 4;
static __global__ void __launch_bounds__(RD_T * RD_T)
row_distance_kernel(const double *__restrict__ X, int N, int metric, const double *__restrict__ max_src, double *__restrict__ D) {
    __shared__ double A[RD_KC][RD_TILE + 1], B[RD_KC][RD_TILE + 1];          // [k][row]: a thread's 4 rows are 16 apart
    const int ti = threadIdx.x / RD_T, tj = threadIdx.x % RD_T;
    const int i0 = blockIdx.y * RD_TILE, j0 = blockIdx.x * RD_TILE;
    const double scale = max_src ? 1.0 / max_src[0] : 1.0;
    double acc[RD_R][RD_R], na[RD_R], nb[RD_R];
#pragma unroll
    for (int a = 0; a < RD_R; ++a) {
        na[a] = 0.0; nb[a] = 0.0;
#pragma unroll
        for (int b = 0; b < RD_R; ++b) acc[a][b] = 0.0;
    }
    for (int k0 = 0; k0 < N; k0 += RD_KC) {
        for (int t = threadIdx.x; t < RD_TILE * RD_KC; t += RD_T * RD_T) {
            const int r = t / RD_KC, k = t % RD_KC;                          // (consecutive threads: consecutive k of one row)
            A[k][r] = (i0 + r < N && k0 + k < N) ? X[(size_t)(i0 + r) * N + k0 + k] * scale : 0.0;
            B[k][r] = (j0 + r < N && k0 + k < N) ? X[(size_t)(j0 + r) * N + k0 + k] * scale : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < RD_KC; ++k) {
            double av[RD_R], bv[RD_R];
#pragma unroll
            for (int a = 0; a < RD_R; ++a) { av[a] = A[k][ti + RD_T * a]; bv[a] = B[k][tj + RD_T * a]; }
            if (metric == 0) {
#pragma unroll
                for (int a = 0; a < RD_R; ++a)
#pragma unroll
                    for (int b = 0; b < RD_R; ++b) { const double d = av[a] - bv[b]; acc[a][b] += d * d; }
            } else {
#pragma unroll
                for (int a = 0; a < RD_R; ++a) {
                    na[a] += av[a] * av[a]; nb[a] += bv[a] * bv[a];
#pragma unroll
                    for (int b = 0; b < RD_R; ++b) acc[a][b] += av[a] * bv[b];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < RD_R; ++a)
#pragma unroll
        for (int b = 0; b < RD_R; ++b) {
            const int i = i0 + ti + RD_T * a, j = j0 + tj + RD_T * b;
            if (i < N && j < N) {
                double out;
                if (metric == 0) out = sqrt(acc[a][b]);
                else {
                    out = 1.0 - acc[a][b] / (sqrt(na[a]) * sqrt(nb[b]));
                    out = out < 0.0 ? 0.0 : (out > 2.0 ? 2.0 : out);
                }
                D[(size_t)i * N + j] = i == j ? 0.0 : out;
            }
        }
}

// sklearn.metrics.silhouette_samples on a precomputed distance matrix: for sample i, a = mean distance to the other members
// of its cluster, b = smallest mean distance to another cluster, s = (b - a) / max(a, b), 0 for a singleton cluster.
// One workgroup per sample.  The row and the labels are staged in LDS; thread c then adds up cluster c's distances IN INDEX
// ORDER -- the order of sklearn's np.bincount(labels, weights=row) -- so the sums are reproducible bit for bit (LDS float
// atomics would make them depend on the arrival order).  LDS: N doubles + N ints + C doubles.  s_out[i] = s_i.
// staged == 0 (a row that does not fit LDS, N beyond ~12 700): the row and the labels are read from global memory in the
// same order instead -- the same sums bit for bit, only C doubles of LDS, no limit on N.
static __global__ void silhouette_kernel(const double *__restrict__ D, const int *__restrict__ labels, const int *__restrict__ sizes,
                                         int N, int C, int staged, double *__restrict__ s_out) {
    extern __shared__ double sil_smem[];
    double *csum = sil_smem;
    const int i = blockIdx.x;
    const double *row = D + (size_t)i * N;
    const int *lab = labels;
    if (staged) {
        double *srow = csum + C;
        int *slab = reinterpret_cast<int *>(srow + N);
        for (int j = threadIdx.x; j < N; j += blockDim.x) { srow[j] = row[j]; slab[j] = labels[j]; }
        row = srow; lab = slab;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0.0;
        if (sizes[c] > 0)
            for (int j = 0; j < N; ++j) if (lab[j] == c) s += row[j];       // (LDS broadcast reads)
        csum[c] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int li = lab[i];
        double s = 0.0;
        if (sizes[li] > 1) {
            const double a = csum[li] / (sizes[li] - 1);
            double b = __builtin_inf();
            for (int c = 0; c < C; ++c)
                if (c != li && sizes[c] > 0) { const double m = csum[c] / sizes[c]; b = m < b ? m : b; }
            const double mx = a > b ? a : b;
            s = mx > 0.0 ? (b - a) / mx : 0.0;
        }
        s_out[i] = s;
    }
}

// Row i of the k-nearest-neighbour Gaussian kernel: Kmat[i][j] = exp(-D[i][j]^2 / (4 epsilon)) for the k nearest rows of row i
// (the point itself, at distance 0, counts as its own first neighbour, as in sklearn's kneighbors_graph on the fitted
// data), 0 elsewhere.  EXACTLY k entries per row, like sklearn's kneighbors: rows tied at the k-th distance are taken in
// index order (sklearn's own choice among exact ties is whatever its partial sort leaves -- unspecified; the smallest
// indices are a fixed, reproducible choice of the same size).  One workgroup per row: the k-th smallest value by a bitonic
// sort of the row in LDS, then the ties counted in index order.
static __global__ void knn_kernel_kernel(const double *__restrict__ D, int N, int NP2, int k, double epsilon, double *__restrict__ Kmat) {
    extern __shared__ double srt[];           // NP2 (next power of two >= N), padded with +inf
    __shared__ int tie_room;
    const int i = blockIdx.x;
    for (int t = threadIdx.x; t < NP2; t += blockDim.x) srt[t] = t < N ? D[(size_t)i * N + t] : __builtin_inf();
    __syncthreads();
    for (int size = 2; size <= NP2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < NP2 / 2; t += blockDim.x) {
                const int lo = (t / stride) * 2 * stride + t % stride, hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const double a = srt[lo], b = srt[hi];
                if ((a > b) == up) { srt[lo] = b; srt[hi] = a; }
            }
            __syncthreads();
        }
    const int kk = k - 1 < N ? k - 1 : N - 1;
    const double thr = srt[kk];
    if (threadIdx.x == 0) {                    // entries strictly below the k-th value: the first position of thr in the sorted row
        int lo = 0, hi = kk;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (srt[mid] < thr) lo = mid + 1; else hi = mid; }
        tie_room = kk + 1 - lo;                // how many entries EQUAL to thr belong to the k nearest
    }
    __syncthreads();
    const int room = tie_room;
    // ties in index order: a tied entry j is kept when fewer than `room` tied entries precede it.  Wave 0 walks the row in
    // chunks of 64 with a ballot prefix count (N <= a few thousand: a handful of iterations).
    if (threadIdx.x < 64) {
        int seen = 0;
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int j = j0 + (int)threadIdx.x;
            const double d = j < N ? D[(size_t)i * N + j] : __builtin_inf();
            const bool tie = j < N && d == thr;
            const unsigned long long m = __ballot(tie);
            const int rank = seen + (int)__popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (j < N) Kmat[(size_t)i * N + j] = (d < thr || (tie && rank < room)) ? exp(-d * d / (4.0 * epsilon)) : 0.0;
            seen += (int)__popcll(m);
        }
    }
}

// per-cluster sizes from the labels (one workgroup; N <= a few thousand)
static __global__ void label_sizes_kernel(const int *__restrict__ labels, int N, int C, int *__restrict__ sizes) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) sizes[c] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const int l = labels[j];
        if (l >= 0 && l < C) atomicAdd(&sizes[l], 1);
    }
}

}  // namespace pilot
