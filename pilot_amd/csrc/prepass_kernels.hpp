// Device pre-pass of pilotpy.tl.wasserstein_distance:
//   * proportions  -- Cluster_Representations, pilotpy/tools/Trajectory.py:377-436: (sample, cell type) histogram
//                     + prior_k = regulizer * n_k / (C - 1) smoothing, fp64, operation for operation like the
//                     reference so the result is bit-identical;
//   * centroids    -- cost_matrix, Trajectory.py:462-466: per-cell-type column-wise MEDIAN of the C x D embedding,
//                     exact (radix select on the order-preserving integer image of the floats, all D dimensions
//                     of a cell type in one sweep so every embedding row is read as one contiguous line).
// Both are HBM-bound integer/byte work: coalesced row reads, LDS-privatised histograms, no MFMA.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

// ---- proportions --------------------------------------------------------------------------------------------
// first_row (nullable, N entries preset to 0xffffffff): smallest row number of every sample -- the row whose status value
// return_real_labels reports (Trajectory.py:617-642).  The plain read in front of the atomic keeps all but the first few
// cells of a sample off the atomic unit.
__global__ void count_kernel(const int *__restrict__ cell_code, const int *__restrict__ sample_code, long C, int K,
                             unsigned int *__restrict__ counts /* N*K */, unsigned int *__restrict__ first_row) {
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < C; c += (long)gridDim.x * blockDim.x) {
        const int k = cell_code[c], s = sample_code[c];
        if (k >= 0 && s >= 0) atomicAdd(&counts[(size_t)s * K + k], 1u);
        if (first_row && s >= 0 && (unsigned int)c < __builtin_nontemporal_load(&first_row[s])) atomicMin(&first_row[s], (unsigned int)c);
    }
}

// one workgroup; every fp64 operation in the order the reference performs it (Trajectory.py:405-430)
__global__ void proportions_kernel(const unsigned int *__restrict__ counts, int N, int K, long n_total,
                                   double regulizer, int normalization, double *__restrict__ P) {
    extern __shared__ double sh[];       // prior[K], then sum_prior
    double *prior = sh;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        unsigned long long nk = 0;
        for (int n = 0; n < N; ++n) nk += counts[(size_t)n * K + k];
        double pr = double(nk) / double(n_total - 1);      // :407   n_k / (C - 1)
        prior[k] = pr * regulizer;                          // :409
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;                                     // Python sum(prior): int 0 + p0 + p1 + ...
        for (int k = 0; k < K; ++k) s += prior[k];
        sh[K] = s;
    }
    __syncthreads();
    const double sum_prior = sh[K];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double rs = 0.0;                                    // sum(counts[n]): integers, exact in any order
        for (int k = 0; k < K; ++k) rs += double(counts[(size_t)n * K + k]);
        for (int k = 0; k < K; ++k) {
            const double c = double(counts[(size_t)n * K + k]);
            P[(size_t)n * K + k] = normalization ? (c + prior[k]) / (rs + sum_prior) : c;   // :430
        }
    }
}

// ---- medians ------------------------------------------------------------------------------------------------
template <typename T> struct OrderedKey;
template <> struct OrderedKey<float> {
    using U = unsigned int;
    static constexpr int BITS = 32;
    __device__ static inline U enc(float x) {
        const U u = __float_as_uint(x);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // total order of finite floats as unsigned ints
    }
    __device__ static inline float dec(U k) {
        const U u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __uint_as_float(u);
    }
};
template <> struct OrderedKey<double> {
    using U = unsigned long long;
    static constexpr int BITS = 64;
    __device__ static inline U enc(double x) {
        const U u = (U)__double_as_longlong(x);
        return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    }
    __device__ static inline double dec(U k) {
        const U u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)u);
    }
};

// cells grouped by type: perm[offset[k] + i] = index of the i-th cell of type k (order inside a type is irrelevant)
__global__ void type_count_kernel(const int *__restrict__ cell_code, long C, int K, unsigned int *__restrict__ n_k) {
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < C; c += (long)gridDim.x * blockDim.x) {
        const int k = cell_code[c];
        if (k >= 0 && k < K) atomicAdd(&n_k[k], 1u);
    }
}
__global__ void type_offsets_kernel(const unsigned int *__restrict__ n_k, int K, unsigned int *__restrict__ offs /* K+1 */) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned int run = 0;
        for (int k = 0; k < K; ++k) { offs[k] = run; run += n_k[k]; }
        offs[K] = run;
    }
}
__global__ void type_scatter_kernel(const int *__restrict__ cell_code, long C, int K, const unsigned int *__restrict__ offs,
                                    unsigned int *__restrict__ cursor, unsigned int *__restrict__ perm) {
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < C; c += (long)gridDim.x * blockDim.x) {
        const int k = cell_code[c];
        if (k >= 0 && k < K) perm[offs[k] + atomicAdd(&cursor[k], 1u)] = (unsigned int)c;
    }
}

// Radix-select state per (type k, dimension d, query q): q = 0 -> rank floor((n-1)/2), q = 1 -> rank floor(n/2)
// (the two middle elements; equal ranks when n is odd).  prefix = key bits fixed so far, rank = rank of the
// wanted element among the keys that share the prefix.
template <typename U> struct SelectState { U prefix; unsigned int rank; };

template <typename T>
__global__ void select_init_kernel(const unsigned int *__restrict__ n_k, int K, int D,
                                   SelectState<typename OrderedKey<T>::U> *__restrict__ st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * D * 2) return;
    const int q = idx & 1, k = idx / (2 * D);
    const unsigned int n = n_k[k];
    st[idx].prefix = 0;
    st[idx].rank = n ? (q == 0 ? (n - 1) / 2 : n / 2) : 0;
}

// one radix pass (8 bits at `shift`): histogram of the digit over the keys matching each query's prefix.
// grid = (splits, K); every workgroup sweeps a slice of the rows of type k, the dimensions [dbeg, dbeg + Dw) at once
// (the LDS histograms hold Dw <= SELECT_MAX_DIMS dimensions; wider embeddings take several launches per pass).
constexpr int SELECT_MAX_DIMS = 64;
template <typename T>
__global__ void select_hist_kernel(const T *__restrict__ X, int D, int dbeg, int Dw, const unsigned int *__restrict__ perm,
                                   const unsigned int *__restrict__ offs, int shift,
                                   const SelectState<typename OrderedKey<T>::U> *__restrict__ st,
                                   unsigned int *__restrict__ hist /* K * D * 2 * 256 */) {
    using OK = OrderedKey<T>;
    using U = typename OK::U;
    extern __shared__ unsigned int lh[];                   // Dw * 2 * 256
    const int k = blockIdx.y;
    const unsigned int beg = offs[k], end = offs[k + 1];
    const int nbins = Dw * 2 * 256;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const U himask = (shift + 8 >= OK::BITS) ? U(0) : (~U(0) << (shift + 8));
    // a wave handles one row at a time: lane d (< D) takes dimension d -> the row is one coalesced read
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64, nwaves = blockDim.x / 64;
    const unsigned int per = (end - beg + gridDim.x - 1) / gridDim.x;
    const unsigned int r0 = beg + blockIdx.x * per, r1 = (r0 + per < end) ? r0 + per : end;
    for (int d0 = 0; d0 < Dw; d0 += 64) {
        const int dl = d0 + lane, d = dbeg + dl;
        U p0 = 0, p1 = 0;
        if (dl < Dw) { p0 = st[(k * D + d) * 2 + 0].prefix; p1 = st[(k * D + d) * 2 + 1].prefix; }
        for (unsigned int r = r0 + wave; r < r1; r += nwaves) {
            const size_t row = perm[r];
            if (dl < Dw) {
                const U key = OK::enc(X[row * D + d]);
                const unsigned int digit = (unsigned int)(key >> shift) & 255u;
                if ((key & himask) == p0) atomicAdd(&lh[((dl * 2 + 0) << 8) + digit], 1u);
                if ((key & himask) == p1) atomicAdd(&lh[((dl * 2 + 1) << 8) + digit], 1u);
            }
        }
    }
    __syncthreads();
    unsigned int *gh = hist + ((size_t)k * D + dbeg) * 2 * 256;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// pick the digit that contains the wanted rank; fix it in the prefix; re-zero the histogram for the next pass
template <typename T>
__global__ void select_pick_kernel(int K, int D, int shift, SelectState<typename OrderedKey<T>::U> *__restrict__ st,
                                   unsigned int *__restrict__ hist) {
    using U = typename OrderedKey<T>::U;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (k, d, q)
    if (idx >= K * D * 2) return;
    unsigned int *h = hist + (size_t)idx * 256;
    unsigned int rank = st[idx].rank, run = 0;
    int digit = 255;
    for (int b = 0; b < 256; ++b) {
        const unsigned int c = h[b];
        if (rank < run + c) { digit = b; break; }
        run += c;
    }
    for (int b = 0; b < 256; ++b) h[b] = 0;
    st[idx].prefix |= (U(digit) << shift);
    st[idx].rank = rank - run;
}

// median = mean of the two middle elements, computed in the data's own dtype like numpy/pandas, then widened
template <typename T>
__global__ void select_finish_kernel(const unsigned int *__restrict__ n_k, int K, int D,
                                     const SelectState<typename OrderedKey<T>::U> *__restrict__ st,
                                     double *__restrict__ centroids /* K x D */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (k, d)
    if (idx >= K * D) return;
    const int k = idx / D;
    if (n_k[k] == 0) { centroids[idx] = __longlong_as_double(0x7ff8000000000000ll); return; }   // NaN, like an empty slice
    const T lo = OrderedKey<T>::dec(st[idx * 2 + 0].prefix), hi = OrderedKey<T>::dec(st[idx * 2 + 1].prefix);
    const T m = (n_k[k] & 1u) ? lo : T((lo + hi) * T(0.5));
    centroids[idx] = double(m);
}

// SMALL cohorts (the reference test's own: 24 227 cells, 14 clusters, 14 features): the general path above is sixteen launches,
// three memsets and two staging copies -- 1.1 ms for a few hundred microseconds' worth of data, a quarter of the whole
// tl.wasserstein_distance call (profiles/r05/e2e_tl_wasserstein_distance_kidney.txt).  Here ONE launch does it: workgroup (k, d)
// gathers the order-preserving keys of cell type k's values of dimension d into LDS (every workgroup reads all C codes: the host
// takes this path only while C x K x D is small) and radix-selects the two middle elements there, 8 bits per pass, both
// queries at once.  Same keys, same ranks, same final arithmetic as select_*_kernel: the same bits.
constexpr int SMALL_MEDIANS_CAP = 8192;                 // cells of one type the LDS key buffer holds
template <typename T>
__global__ void __launch_bounds__(256) small_medians_kernel(const T *__restrict__ X, int D, const int *__restrict__ cell_code, long C, int K,
                                                            double *__restrict__ centroids /* K x D */) {
    using OK = OrderedKey<T>;
    using U = typename OK::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char small_medians_smem[];
    U *keys = reinterpret_cast<U *>(small_medians_smem);
    __shared__ unsigned int cnt, hist[2][256], rnk[2];
    __shared__ U pref[2];
    const int k = (int)blockIdx.x / D, d = (int)blockIdx.x % D;
    if (threadIdx.x == 0) cnt = 0u;
    __syncthreads();
    for (long c = threadIdx.x; c < C; c += blockDim.x)
        if (cell_code[c] == k) {
            const unsigned int pos = atomicAdd(&cnt, 1u);
            if (pos < (unsigned int)SMALL_MEDIANS_CAP) keys[pos] = OK::enc(X[(size_t)c * D + d]);
        }
    __syncthreads();
    const unsigned int n = cnt;
    if (n == 0u || n > (unsigned int)SMALL_MEDIANS_CAP) {       // an empty slice: NaN like pandas (beyond the cap: the host never sends such a cohort here)
        if (threadIdx.x == 0) centroids[(size_t)k * D + d] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    if (threadIdx.x == 0) { pref[0] = pref[1] = U(0); rnk[0] = (n - 1u) / 2u; rnk[1] = n / 2u; }
    for (int shift = OK::BITS - 8; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) (&hist[0][0])[i] = 0u;
        __syncthreads();
        const U himask = (shift + 8 >= OK::BITS) ? U(0) : (~U(0) << (shift + 8));
        const U p0 = pref[0], p1 = pref[1];
        for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
            const U key = keys[i];
            const unsigned int digit = (unsigned int)(key >> shift) & 255u;
            if ((key & himask) == p0) atomicAdd(&hist[0][digit], 1u);
            if ((key & himask) == p1) atomicAdd(&hist[1][digit], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int q = threadIdx.x;
            unsigned int rank = rnk[q], run = 0u;
            int digit = 255;
            for (int b = 0; b < 256; ++b) {
                const unsigned int c = hist[q][b];
                if (rank < run + c) { digit = b; break; }
                run += c;
            }
            pref[q] |= (U(digit) << shift);
            rnk[q] = rank - run;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const T lo = OK::dec(pref[0]), hi = OK::dec(pref[1]);
        const T m = (n & 1u) ? lo : T((lo + hi) * T(0.5));
        centroids[(size_t)k * D + d] = double(m);
    }
}

}  // namespace pilot
