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
// One pass over the two code columns: counts[s][k] (cells with both labels), n_k[k] (cells with a type label: what the
// median pass groups), first_row[s] (nullable: smallest row number c of every sample -- the row whose status value
// return_real_labels reports, Trajectory.py:617-642 -- kept as 0xffffffff - c under atomicMax, so that 0 = "no cell" and one
// memset clears every output of the pass).
// A cohort is stored sample after sample, so the 2048 cells a block takes at a time belong to one or two samples: 3.6 M
// global atomics on a handful of hot addresses (round 5: 1.2 ms for 14 MB of codes) become LDS atomics on a window of
// sample rows [s_min, s_max] of the block's chunk plus one global add per non-empty bin; a chunk whose samples span more
// rows than the window holds falls back to global atomics, bin for bin the same integers.
constexpr int COUNT_CHUNK = 2048;        // cells per block and step
constexpr int COUNT_LDS_BINS = 8192;     // (sample window) x K counters in LDS
constexpr int COUNT_LDS_ROWS = 1024;     // widest sample window
__global__ void __launch_bounds__(256) count_kernel(const int *__restrict__ cell_code, const int *__restrict__ sample_code, long C, int N, int K,
                                                    unsigned int *__restrict__ counts /* N*K */, unsigned int *__restrict__ n_k /* K, nullable */,
                                                    unsigned int *__restrict__ first_row) {
    extern __shared__ unsigned int ck_sm[];                  // bins[COUNT_LDS_BINS] | first[COUNT_LDS_ROWS] | nk[K] | red[16]
    unsigned int *bins = ck_sm, *first = bins + COUNT_LDS_BINS, *nk = first + COUNT_LDS_ROWS;
    int *red = reinterpret_cast<int *>(nk + K);
    constexpr int PER = COUNT_CHUNK / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < K; k += 256) nk[k] = 0u;
    const long nchunks = (C + COUNT_CHUNK - 1) / COUNT_CHUNK;
    for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const long c0 = ch * COUNT_CHUNK;
        int kk[PER], ss[PER];
        int smin = 0x7fffffff, smax = -1;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const long c = c0 + i * 256 + threadIdx.x;
            kk[i] = -1; ss[i] = -1;
            if (c < C) { kk[i] = cell_code[c]; ss[i] = sample_code[c]; }
            if (kk[i] >= K) kk[i] = -1;
            if (ss[i] >= N) ss[i] = -1;
            if (ss[i] >= 0) { smin = ss[i] < smin ? ss[i] : smin; smax = ss[i] > smax ? ss[i] : smax; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int a = __shfl_xor(smin, o, 64), b = __shfl_xor(smax, o, 64);
            smin = a < smin ? a : smin; smax = b > smax ? b : smax;
        }
        __syncthreads();                                     // (the previous chunk's flush has read bins / first / red)
        if (lane == 0) { red[wave] = smin; red[4 + wave] = smax; }
        __syncthreads();
        smin = min(min(red[0], red[1]), min(red[2], red[3]));
        smax = max(max(red[4], red[5]), max(red[6], red[7]));
        const int rows = smax >= smin ? smax - smin + 1 : 0;
        const bool in_lds = rows <= COUNT_LDS_ROWS && (long)rows * K <= COUNT_LDS_BINS;
        if (in_lds) {
            for (int i = threadIdx.x; i < rows * K; i += 256) bins[i] = 0u;
            for (int i = threadIdx.x; i < rows; i += 256) first[i] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = kk[i], sidx = ss[i];
            const unsigned int c = 0xffffffffu - (unsigned int)(c0 + i * 256 + threadIdx.x);
            if (k >= 0) atomicAdd(&nk[k], 1u);
            if (in_lds) {
                if (k >= 0 && sidx >= 0) atomicAdd(&bins[(sidx - smin) * K + k], 1u);
                if (first_row && sidx >= 0) atomicMax(&first[sidx - smin], c);
            } else {
                if (k >= 0 && sidx >= 0) atomicAdd(&counts[(size_t)sidx * K + k], 1u);
                if (first_row && sidx >= 0 && c > __builtin_nontemporal_load(&first_row[sidx])) atomicMax(&first_row[sidx], c);
            }
        }
        __syncthreads();
        if (in_lds) {
            for (int i = threadIdx.x; i < rows * K; i += 256) if (bins[i]) atomicAdd(&counts[(size_t)smin * K + i], bins[i]);
            if (first_row)
                for (int i = threadIdx.x; i < rows; i += 256) if (first[i]) atomicMax(&first_row[smin + i], first[i]);
        }
    }
    __syncthreads();
    if (n_k) for (int k = threadIdx.x; k < K; k += 256) if (nk[k]) atomicAdd(&n_k[k], nk[k]);
}

// every fp64 operation in the order the reference performs it (Trajectory.py:405-430).  Two launches: the priors
// (a wave per cell type sums its column of the counts: integers, exact in any order), then a wave per sample.
__global__ void __launch_bounds__(256) prior_kernel(const unsigned int *__restrict__ counts, int N, int K, long n_total, double regulizer,
                                                    double *__restrict__ prior /* K */) {
    const int lane = threadIdx.x & 63, k = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
    if (k >= K) return;
    unsigned long long nk = 0;
    for (int n = lane; n < N; n += 64) nk += counts[(size_t)n * K + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nk += __shfl_xor(nk, o, 64);
    if (lane == 0) {
        const double pr = double(nk) / double(n_total - 1);          // :407   n_k / (C - 1)
        prior[k] = pr * regulizer;                                    // :409
    }
}

__global__ void __launch_bounds__(256) proportions_kernel(const unsigned int *__restrict__ counts, int N, int K,
                                                          const double *__restrict__ prior, int normalization, double *__restrict__ P) {
    __shared__ double sum_prior_sh;
    if (threadIdx.x == 0) {
        double s = 0.0;                                     // Python sum(prior): ((0 + p0) + p1) + ... in this order
        for (int k = 0; k < K; ++k) s += prior[k];
        sum_prior_sh = s;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, n = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
    if (n >= N) return;
    const double sum_prior = sum_prior_sh;
    unsigned long long r = 0;                               // sum(counts[n]): integers, exact in any order
    for (int k = lane; k < K; k += 64) r += counts[(size_t)n * K + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
    const double rs = double(r);
    for (int k = lane; k < K; k += 64) {
        const double c = double(counts[(size_t)n * K + k]);
        P[(size_t)n * K + k] = normalization ? (c + prior[k]) / (rs + sum_prior) : c;   // :430
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

// ---- general path: rows grouped by type, then an 8-bit radix select over contiguous memory -----------------------------------
// Round 5's general path gathered one 120-byte row per wave and step through a permutation (30 of 64 lanes busy, every load
// behind the load of its index) and took 20.5 ms for 1.8 M x 30 floats: 0.5 % of the HBM roof.  Now:
//   1. type_count_kernel      n_k (a block counts its cells in LDS, then K global adds);
//   2. median_prep_kernel     one block: segment starts (every type starts on a row that is a multiple of 4, so a block's
//                             first element is 16-byte aligned) and the work list of (type, row chunk) items for the select
//                             passes;
//   3. group_rows_kernel      ONE read of the embedding, ONE write of its order-preserving keys with the rows of a type
//                             contiguous (a block ranks its 1024 cells per type in LDS, reserves a run per type with one
//                             global add, and copies its rows; reads fully coalesced, writes in whole rows);
//   4. select_hist_kernel  x BITS/8: a block streams contiguous rows of ONE type with 16-byte loads, lane = 4 consecutive
//                             elements, i.e. every lane busy whatever D is; the 256-bin histograms of all D dimensions sit in
//                             LDS, so the lanes of a wave spread over D histograms (2 - 3 lanes on a bin, not 64); the chip
//                             holds every block at once and each takes an equal share of the rows;
//      select_pick_kernel     a wave per (type, dimension): the digit that holds the wanted rank; the last one writes the
//                             medians.
// HBM traffic per call: C*D*s read + written once, then BITS/8 reads of C*D*s -- the model DESIGN.md prices the pass against.
// Tuning, measured on 1.8 M x 30 floats (tools/prepass_variants.sh, profiles/r06/prepass_variants.txt): group_rows 1024 rows a
// block 154 us (512: 165, 256: 187 -- more blocks queue on the cursor adds; 2048: 200, 4096: 278 -- too few waves in flight);
// select_hist 1024 threads a block 48.8 us (512: 51.6 here, 67.9 with one block per 64 K keys instead of equal runs of the work
// list; 256: 107), plain loads (non-temporal: +3 us), four 16-byte loads in flight per lane (2: 49.4, 8: 72).
constexpr int GROUP_ROWS_PER_BLOCK = 1024;
constexpr int SELECT_MAX_DIMS = 64;                      // dimensions per histogram window (LDS: 2 x 64 x 256 counters = 128 KB)
constexpr int SELECT_THREADS = 1024;                     // select_hist_kernel: two such blocks per CU at D = 30 (62 KB of LDS each)

__global__ void __launch_bounds__(256) type_count_kernel(const int *__restrict__ cell_code, long C, int K, unsigned int *__restrict__ n_k) {
    extern __shared__ unsigned int tc_bins[];            // K
    for (int k = threadIdx.x; k < K; k += blockDim.x) tc_bins[k] = 0u;
    __syncthreads();
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += (long)gridDim.x * blockDim.x) {
        const int k = cell_code[c];
        if (k >= 0 && k < K) atomicAdd(&tc_bins[k], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) if (tc_bins[k]) atomicAdd(&n_k[k], tc_bins[k]);
}

// Radix-select state per (type k, dimension d, query q): q = 0 -> rank floor((n-1)/2), q = 1 -> rank floor(n/2)
// (the two middle elements; equal ranks when n is odd).  prefix = key bits fixed so far, rank = rank of the
// wanted element among the keys that share the prefix.
template <typename U> struct SelectState { U prefix; unsigned int rank; };
struct SelectItem { unsigned int k, r0, r1; };          // rows [r0, r1) of the grouped array, all of type k

// exclusive scan of v[0..n) in LDS by one 256-thread block; returns the total (tmp: 257 words of LDS)
__device__ inline unsigned int block_exclusive_scan(unsigned int *v, int n, unsigned int *tmp) {
    const int per = (n + 255) / 256, a = (int)threadIdx.x * per, b = a + per < n ? a + per : n;
    unsigned int s = 0;
    for (int i = a; i < b; ++i) s += v[i];
    tmp[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int i = 0; i < 256; ++i) { const unsigned int x = tmp[i]; tmp[i] = run; run += x; }
        tmp[256] = run;
    }
    __syncthreads();
    unsigned int run = tmp[threadIdx.x];
    for (int i = a; i < b; ++i) { const unsigned int x = v[i]; v[i] = run; run += x; }
    __syncthreads();
    return tmp[256];
}

__global__ void __launch_bounds__(256) median_prep_kernel(const unsigned int *__restrict__ n_k, int K, unsigned int R,
                                                          unsigned int *__restrict__ offs /* K */, unsigned int *__restrict__ n_items,
                                                          SelectItem *__restrict__ items) {
    extern __shared__ unsigned int mp_sm[];              // start[K + 1] | chunk[K + 1] | tmp[257]
    unsigned int *start = mp_sm, *chunk = mp_sm + (K + 1), *tmp = chunk + (K + 1);
    for (int k = threadIdx.x; k < K; k += 256) {
        const unsigned int n = n_k[k];
        start[k] = (n + 3u) & ~3u;                       // rows reserved: the next type starts on a multiple of 4 again
        chunk[k] = (n + R - 1u) / R;
    }
    __syncthreads();
    block_exclusive_scan(start, K, tmp);
    const unsigned int total = block_exclusive_scan(chunk, K, tmp);
    if (threadIdx.x == 0) *n_items = total;
    for (int k = threadIdx.x; k < K; k += 256) offs[k] = start[k];
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < total; i += 256) {
        int lo = 0, hi = K;                               // the last k with chunk[k] <= i (empty types share a value: take the last)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk[mid] <= i) lo = mid; else hi = mid; }
        const unsigned int n = n_k[lo], r0 = start[lo] + (i - chunk[lo]) * R;
        const unsigned int end = start[lo] + n;
        items[i] = SelectItem{(unsigned int)lo, r0, r0 + R < end ? r0 + R : end};
    }
}

template <typename T>
__global__ void __launch_bounds__(256) group_rows_kernel(const T *__restrict__ X, int D, const int *__restrict__ cell_code, long C, int K,
                                                         const unsigned int *__restrict__ offs, unsigned int *__restrict__ cursor /* K, zeroed */,
                                                         typename OrderedKey<T>::U *__restrict__ Y) {
    using OK = OrderedKey<T>;
    extern __shared__ unsigned int gr_sm[];              // cnt[K] | base[K] | dest[GROUP_ROWS_PER_BLOCK]
    unsigned int *cnt = gr_sm, *base = gr_sm + K, *dest = base + K;
    constexpr int PER = GROUP_ROWS_PER_BLOCK / 256;
    for (int k = threadIdx.x; k < K; k += 256) cnt[k] = 0u;
    __syncthreads();
    const long c0 = (long)blockIdx.x * GROUP_ROWS_PER_BLOCK;
    const int rows = (int)(C - c0 < GROUP_ROWS_PER_BLOCK ? C - c0 : GROUP_ROWS_PER_BLOCK);
    int kk[PER];
    unsigned int lr[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int r = i * 256 + (int)threadIdx.x;
        kk[i] = -1;
        if (r < rows) { const int k = cell_code[c0 + r]; if (k >= 0 && k < K) { kk[i] = k; lr[i] = atomicAdd(&cnt[k], 1u); } }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) if (cnt[k]) base[k] = offs[k] + atomicAdd(&cursor[k], cnt[k]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) dest[i * 256 + (int)threadIdx.x] = kk[i] >= 0 ? base[kk[i]] + lr[i] : 0xffffffffu;
    __syncthreads();
    // the block's rows are one contiguous piece of X: element e = r * D + d, no division in the loop; four loads in flight
    const T *src = X + (size_t)c0 * D;
    const unsigned int total = (unsigned int)rows * (unsigned int)D;
    unsigned int r = threadIdx.x / (unsigned int)D, d = threadIdx.x % (unsigned int)D;
    const unsigned int dr = 256u / (unsigned int)D, dd = 256u % (unsigned int)D;
    constexpr int UNR = 4;
    for (unsigned int e = threadIdx.x; e < total; e += 256u * UNR) {
        T v[UNR];
        unsigned int rr[UNR], dv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned int idx = e + 256u * u;
            v[u] = idx < total ? __builtin_nontemporal_load(&src[idx]) : T(0);
            rr[u] = idx < total ? r : 0u; dv[u] = d;
            r += dr; d += dd;
            if (d >= (unsigned int)D) { d -= (unsigned int)D; ++r; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned int row = dest[rr[u]];
            if (e + 256u * u < total && row != 0xffffffffu) Y[(size_t)row * D + dv[u]] = OK::enc(v[u]);
        }
    }
}

// one radix pass (8 bits at `shift`) over the dimension window [dbeg, dbeg + Dw).  The work list is cut into gridDim.x
// equal runs of items; consecutive items of one type are adjacent rows, so a block streams ONE contiguous range per type
// it meets (usually one or two) and flushes its LDS histograms when the type changes.
// VEC keys per 16-byte load when the window is the whole row (a range then starts 16-byte aligned: segment starts and
// item lengths are multiples of 4 rows).
template <typename T>
__global__ void __launch_bounds__(SELECT_THREADS) select_hist_kernel(const typename OrderedKey<T>::U *__restrict__ Y, int D, int dbeg, int Dw,
                                                          const unsigned int *__restrict__ n_items, const SelectItem *__restrict__ items,
                                                          int shift, const SelectState<typename OrderedKey<T>::U> *__restrict__ st,
                                                          unsigned int *__restrict__ hist /* K * D * 2 * 256 */) {
    using OK = OrderedKey<T>;
    using U = typename OK::U;
    constexpr int VEC = 16 / (int)sizeof(U);
    const unsigned int total_items = *n_items;
    const unsigned int i0 = (unsigned int)(((unsigned long long)blockIdx.x * total_items) / gridDim.x),
                       i1 = (unsigned int)(((unsigned long long)(blockIdx.x + 1) * total_items) / gridDim.x);
    if (i0 >= i1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_raw[];
    U *pre = reinterpret_cast<U *>(sh_raw);                                 // [2][Dw] prefixes of the two queries
    unsigned int *lh = reinterpret_cast<unsigned int *>(pre + 2 * Dw);      // [2][Dw][256]
    const int nbins = Dw * 2 * 256;
    const bool first = shift + 8 >= OK::BITS;                               // (no state yet: nothing is fixed, every key counts)
    const U himask = first ? U(0) : (~U(0) << (shift + 8));
    const unsigned int uDw = (unsigned int)Dw;
    for (int i = threadIdx.x; i < nbins; i += SELECT_THREADS) lh[i] = 0u;
    auto tally = [&](U key, unsigned int dl) {
        const unsigned int digit = (unsigned int)(key >> shift) & 255u;
        const U hi = key & himask, p0 = pre[dl], p1 = pre[uDw + dl];
        if (hi == p0) atomicAdd(&lh[(dl << 8) + digit], 1u);
        else if (hi == p1) atomicAdd(&lh[((uDw + dl) << 8) + digit], 1u);   // (the second histogram only where the queries parted)
    };
    unsigned int i = i0;
    while (i < i1) {
        // the run of items of one type: rows [r0, r1)
        const SelectItem head = items[i];
        const int k = (int)head.k;
        unsigned int r1 = head.r1;
        for (++i; i < i1; ++i) { const SelectItem nx = items[i]; if ((int)nx.k != k) break; r1 = nx.r1; }
        const unsigned int r0 = head.r0;
        for (int j = threadIdx.x; j < 2 * Dw; j += SELECT_THREADS) pre[j] = first ? U(0) : st[((size_t)k * D + dbeg + (j % Dw)) * 2 + (j / Dw)].prefix;
        __syncthreads();
        if (Dw == D) {
            const U *src = Y + (size_t)r0 * D;
            const size_t total = (size_t)(r1 - r0) * D, nvec = total / VEC;
            using V = __attribute__((ext_vector_type(VEC))) U;
            const V *src4 = reinterpret_cast<const V *>(src);
            unsigned int dl = (unsigned int)(((size_t)threadIdx.x * VEC) % uDw);
            const unsigned int step = ((unsigned int)SELECT_THREADS * VEC) % uDw;
            size_t v = threadIdx.x;
            constexpr int UNR = 4;
            for (; v + (size_t)(UNR - 1) * SELECT_THREADS < nvec; v += (size_t)UNR * SELECT_THREADS) {
                V x[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) x[u] = src4[v + (size_t)u * SELECT_THREADS];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    unsigned int d2 = dl;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { tally(x[u][j], d2); d2 = d2 + 1u == uDw ? 0u : d2 + 1u; }
                    dl += step; if (dl >= uDw) dl -= uDw;
                }
            }
            for (; v < nvec; v += SELECT_THREADS) {
                const V x = src4[v];
                unsigned int d2 = dl;
#pragma unroll
                for (int j = 0; j < VEC; ++j) { tally(x[j], d2); d2 = d2 + 1u == uDw ? 0u : d2 + 1u; }
                dl += step; if (dl >= uDw) dl -= uDw;
            }
            for (size_t e = nvec * VEC + threadIdx.x; e < total; e += SELECT_THREADS) tally(src[e], (unsigned int)(e % uDw));
        } else {
            // a window of a wide row: element (r, dl) at Y[(r0 + r) * D + dbeg + dl]
            const unsigned int rows = r1 - r0;
            unsigned int r = threadIdx.x / uDw, dl = threadIdx.x % uDw;
            const unsigned int dr = (unsigned int)SELECT_THREADS / uDw, dd = (unsigned int)SELECT_THREADS % uDw;
            while (r < rows) {
                tally(Y[(size_t)(r0 + r) * D + dbeg + dl], dl);
                r += dr; dl += dd;
                if (dl >= uDw) { dl -= uDw; ++r; }
            }
        }
        __syncthreads();
        unsigned int *gh0 = hist + (((size_t)k * D + dbeg) * 2) * 256;      // [d][q][256] in global memory
        for (int j = threadIdx.x; j < nbins; j += SELECT_THREADS) {
            const unsigned int c = lh[j];
            if (c) {
                const int q = j / (Dw * 256), dl = (j >> 8) % Dw;
                atomicAdd(&gh0[((size_t)dl * 2 + q) * 256 + (j & 255)], c);
                lh[j] = 0u;                                                   // (clean for the next type)
            }
        }
        __syncthreads();
    }
}

// a wave per (type, dimension): pick the digit that contains the wanted rank of both queries, fix it in the prefix,
// re-zero the histograms for the next pass; on the last pass write the median = mean of the two middle elements,
// computed in the data's own dtype like numpy/pandas, then widened.
template <typename T>
__global__ void __launch_bounds__(256) select_pick_kernel(const unsigned int *__restrict__ n_k, int K, int D, int shift,
                                                          SelectState<typename OrderedKey<T>::U> *__restrict__ st,
                                                          unsigned int *__restrict__ hist, double *__restrict__ centroids /* K x D, last pass only */) {
    using OK = OrderedKey<T>;
    using U = typename OK::U;
    const int lane = threadIdx.x & 63, idx = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);   // (k, d)
    if (idx >= K * D) return;
    SelectState<U> *s = st + (size_t)idx * 2;
    const bool first = shift + 8 >= OK::BITS;
    const unsigned int n_all = n_k[idx / D];
    const U p0 = first ? U(0) : s[0].prefix, p1 = first ? U(0) : s[1].prefix;
    const bool parted = p0 != p1;
    U pref[2] = {p0, p1};
    unsigned int rank[2] = {first ? (n_all ? (n_all - 1u) / 2u : 0u) : s[0].rank, first ? n_all / 2u : s[1].rank};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        unsigned int *h = hist + ((size_t)idx * 2 + ((q == 1 && parted) ? 1 : 0)) * 256;
        unsigned int c[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = h[lane * 4 + j]; sum += c[j]; }
        unsigned int incl = sum;                                   // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
        const unsigned long long has = __ballot(rank[q] < incl);
        int digit = 255; unsigned int below = 0;
        if (has) {
            const int L = __builtin_ctzll(has);
            unsigned int run = __shfl(incl - sum, L, 64);
            const unsigned int c0 = __shfl(c[0], L, 64), c1 = __shfl(c[1], L, 64), c2 = __shfl(c[2], L, 64);
            int j = 0;
            if (rank[q] >= run + c0) { run += c0; j = 1; if (rank[q] >= run + c1) { run += c1; j = 2; if (rank[q] >= run + c2) { run += c2; j = 3; } } }
            digit = L * 4 + j; below = run;
        } else {
            below = __shfl(incl, 63, 64);                          // (an empty type: nothing matches; the value is never used)
        }
        pref[q] |= U((unsigned int)digit) << shift;
        rank[q] -= has ? below : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    // every lane has read both histograms: zero them for the next pass
    unsigned int *h = hist + (size_t)idx * 2 * 256;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[lane + 64 * j] = 0u;
    if (lane < 2) { s[lane].prefix = pref[lane]; s[lane].rank = rank[lane]; }
    if (shift == 0 && lane == 0) {
        if (n_all == 0u) { centroids[idx] = __longlong_as_double(0x7ff8000000000000ll); return; }   // NaN, like an empty slice
        const T lo = OK::dec(pref[0]), hi = OK::dec(pref[1]);
        const T m = (n_all & 1u) ? lo : T((lo + hi) * T(0.5));
        centroids[idx] = double(m);
    }
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
