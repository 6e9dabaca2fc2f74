// Exact optimal transport for ANY number of cell types up to 2048: the fallback of the one-wave-per-pair kernel
// (emd_kernels.hpp, K <= 256).  The reference's ot.emd2 loop has no such limit (pilotpy/tools/Trajectory.py:507-511).
//
// One 256-thread WORKGROUP per ordered pair; same algorithm (successive shortest augmenting paths with node potentials,
// diagonal warm start, multi-source searches over column labels, rows reached over tight backward arcs and scanned at once,
// the initial labels A_j = min over sources of (M_ij - pu_i) kept between searches and repaired only in the columns whose
// arg-min source ran dry), one augmentation per search.  Vectors live in LDS (7 K doubles + 3 K ints), the flow matrix and A
// in a global slab per resident workgroup, M is read from L2.  A correct fallback, not a fast path: nothing on the PILOT
// path has more than a few dozen cell types.
#pragma once
#include <hip/hip_runtime.h>

#include "emd_kernels.hpp"

namespace pilot {

constexpr int EMDG_WG = 256;
constexpr int EMDG_MAX_K = 2048;
// LDS of one workgroup: pu, pv, ra, rb, dC, fR, fC (doubles), parC, parR, reach list (ints), open / source flags (bytes),
// reduction scratch
__host__ __device__ constexpr size_t emdg_lds_bytes(int K) {
    return sizeof(double) * (7 * (size_t)K + 2 * EMDG_WG / 64 + 8) + sizeof(int) * (3 * (size_t)K + EMDG_WG / 64 + 8) + 2 * (size_t)K + 16;
}
// global slab per resident workgroup: F (K*K), A (K) doubles, then Apar (K ints, padded to 8 bytes)
__host__ __device__ constexpr size_t emdg_slab_doubles(int K) { return (size_t)K * K + K + (K + 1) / 2; }

struct BlockMin { double v; int i; };
// workgroup-wide (value, index) minimum, smallest index among equal values; every thread gets the result
__device__ inline BlockMin block_argmin(double v, int i, double *red_v, int *red_i) {
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    if (lane == 0) { red_v[wave] = v; red_i[wave] = i; }
    __syncthreads();
    BlockMin r{red_v[0], red_i[0]};
#pragma unroll
    for (int w = 1; w < EMDG_WG / 64; ++w)
        if (red_v[w] < r.v || (red_v[w] == r.v && red_i[w] < r.i)) { r.v = red_v[w]; r.i = red_i[w]; }
    __syncthreads();
    return r;
}
__device__ inline double block_sum(double v, double *red_v) {
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red_v[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < EMDG_WG / 64; ++w) s += red_v[w];
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(EMDG_WG) emd_generic_kernel(EmdParams p, const double *__restrict__ rowmin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char emdg_smem[];
    const int K = p.K, N = p.N, tid = threadIdx.x;
    double *pu = reinterpret_cast<double *>(emdg_smem), *pv = pu + K, *ra = pv + K, *rb = ra + K, *dC = rb + K, *fR = dC + K, *fC = fR + K;
    double *red_v = fC + K;                                           // EMDG_WG / 64 entries (+ spare)
    double *sh_d = red_v + EMDG_WG / 64 + 4;                          // a few broadcast doubles
    int *parC = reinterpret_cast<int *>(sh_d + EMDG_WG / 64 + 4), *parR = parC + K, *reach = parR + K;
    int *red_i = reach + K, *sh_i = red_i + EMDG_WG / 64 + 4;         // sh_i: [0] reach count, [1] flag, [2] work item
    unsigned char *openC = reinterpret_cast<unsigned char *>(sh_i + 4), *src = openC + K;
    double *F = p.f_slab + (size_t)blockIdx.x * emdg_slab_doubles(K);
    double *A = F + (size_t)K * K;
    int *Apar = reinterpret_cast<int *>(A + K);
    const double *M = p.M;
    const double INF = __builtin_inf();
    const long total = (long)p.n_rows * N;
    const long n_items = p.upper_only ? (long)p.n_rows * (N - p.row_begin) - (long)p.row_step * p.n_rows * (p.n_rows - 1) / 2 : total;
    auto row_offset = [&](long r) { return r * (N - p.row_begin) - (long)p.row_step * r * (r - 1) / 2; };

    for (bool first = true;; first = false) {
        if (tid == 0) sh_i[2] = first ? (int)blockIdx.x : (int)gridDim.x + atomicAdd(p.queue, 1);
        __syncthreads();
        const long t = sh_i[2];
        __syncthreads();
        if (t >= n_items) break;
        int r, j_s;
        if (p.upper_only) {
            long rr = 0;
            while (rr + 1 < p.n_rows && row_offset(rr + 1) <= t) ++rr;      // (a fallback: the linear walk is fine)
            r = (int)rr;
            j_s = p.row_begin + r * p.row_step + (int)(t - row_offset(rr));
        } else {
            r = (int)(t / N); j_s = (int)(t % N);
        }
        const long q = (long)r * N + j_s;
        const int i_s = p.row_begin + r * p.row_step;

        // POT pre-step b *= sum(a) / sum(b); potentials; zero flow; diagonal warm start
        double sa = 0.0, sb = 0.0;
        for (int k = tid; k < K; k += EMDG_WG) {
            ra[k] = p.P[(size_t)i_s * K + k]; rb[k] = p.P[(size_t)j_s * K + k];
            sa += ra[k]; sb += rb[k];
        }
        sa = block_sum(sa, red_v); sb = block_sum(sb, red_v);
        const double scale = sa / sb, tol = 1e-15 * (sa > 0.0 ? sa : 1.0);
        for (size_t e = tid; e < (size_t)K * K; e += EMDG_WG) F[e] = 0.0;
        for (int k = tid; k < K; k += EMDG_WG) {
            rb[k] *= scale;
            pu[k] = rowmin[k]; pv[k] = 0.0;
            src[k] = 0;
        }
        __syncthreads();
        for (int k = tid; k < K; k += EMDG_WG)
            if (M[(size_t)k * K + k] - pu[k] == 0.0) {
                const double f = ra[k] < rb[k] ? ra[k] : rb[k];
                if (f > 0.0) { F[(size_t)k * K + k] = f; ra[k] -= f; rb[k] -= f; }
            }
        __syncthreads();

        int n_aug = 0, trip = 0;
        bool have_A = false;
        for (int guard = 0;; ++guard) {
            if (guard > 64 * K + 64) { trip = 5; break; }
            // ---- sources; A_j = min over sources of (M_ij - pu_i), repaired where its arg-min source ran dry ----
            if (tid == 0) { sh_i[0] = 0; sh_i[1] = 0; }
            __syncthreads();
            for (int k = tid; k < K; k += EMDG_WG) {
                const unsigned char is = ra[k] > tol;
                if (is) sh_i[1] = 1;                                  // (benign race: all writers store 1)
                if (have_A && src[k] && !is) reach[atomicAdd(&sh_i[0], 1)] = k;     // a source that ran dry since the last search
                src[k] = is;
            }
            __syncthreads();
            if (!sh_i[1]) break;                                      // no supply left: done
            if (!have_A) {
                for (int j = tid; j < K; j += EMDG_WG) {
                    double a = INF; int ap = -1;
                    for (int i = 0; i < K; ++i)
                        if (src[i]) { const double v = M[(size_t)i * K + j] - pu[i]; if (v < a) { a = v; ap = i; } }
                    A[j] = a; Apar[j] = ap;
                }
                have_A = true;
            } else if (sh_i[0] > 0) {
                for (int j = tid; j < K; j += EMDG_WG) {
                    const int ap = Apar[j];
                    if (ap >= 0 && !src[ap]) {                        // this column's arg-min is gone: rebuild the column
                        double a = INF; int np_ = -1;
                        for (int i = 0; i < K; ++i)
                            if (src[i]) { const double v = M[(size_t)i * K + j] - pu[i]; if (v < a) { a = v; np_ = i; } }
                        A[j] = a; Apar[j] = np_;
                    }
                }
            }
            __syncthreads();
            // ---- labels ----
            for (int k = tid; k < K; k += EMDG_WG) {
                double rc = A[k] - pv[k];
                dC[k] = rc > 0.0 ? rc : 0.0; fC[k] = INF; parC[k] = Apar[k]; openC[k] = 1;
                fR[k] = src[k] ? 0.0 : INF; parR[k] = -1;
            }
            __syncthreads();
            int target = -1;
            double dstar = 0.0;
            for (int step = 0; step <= K; ++step) {
                double bv = INF; int bi = 0x7fffffff;
                for (int j = tid; j < K; j += EMDG_WG)
                    if (openC[j] && (dC[j] < bv || (dC[j] == bv && j < bi))) { bv = dC[j]; bi = j; }
                const BlockMin m = block_argmin(bv, bi, red_v, red_i);
                if (!(m.v < INF)) break;                              // nothing (more) reachable: only rounding dust is left
                const int best = m.i;
                const double bd = m.v;
                if (tid == 0) { openC[best] = 0; fC[best] = bd; sh_i[0] = 0; }
                __syncthreads();
                if (rb[best] > 0.0) { target = best; dstar = bd; break; }     // (uniform: LDS value)
                for (int i = tid; i < K; i += EMDG_WG)
                    if (fR[i] == INF && F[(size_t)i * K + best] > 0.0) {       // rows that ship to this column: reduced cost 0
                        fR[i] = bd; parR[i] = best;
                        reach[atomicAdd(&sh_i[0], 1)] = i;
                    }
                __syncthreads();
                const int n_reach = sh_i[0];
                for (int e = 0; e < n_reach; ++e) {
                    const int i = reach[e];
                    const double pu_i = pu[i];
                    for (int j = tid; j < K; j += EMDG_WG)
                        if (openC[j]) {
                            double rc = M[(size_t)i * K + j] - pu_i - pv[j];
                            rc = rc > 0.0 ? rc : 0.0;
                            const double nd = bd + rc;
                            if (nd < dC[j] || (nd == dC[j] && i < parC[j])) { dC[j] = nd; parC[j] = i; }    // (order of `reach` is not fixed)
                        }
                }
                __syncthreads();
            }
            if (target < 0) break;
            for (int k = tid; k < K; k += EMDG_WG) {
                pu[k] -= fR[k] < dstar ? fR[k] : dstar;
                pv[k] += fC[k] < dstar ? fC[k] : dstar;
            }
            __syncthreads();
            if (tid == 0) {            // the path: at most 2 K hops of pointer chasing
                double delta = rb[target];
                int s_row = -1, hops = 0;
                for (int j = target; hops <= 2 * K + 2; ++hops) {
                    const int i = parC[j];
                    if (i < 0) break;
                    const int jb = parR[i];
                    if (jb < 0) { s_row = i; break; }
                    const double f = F[(size_t)i * K + jb];
                    delta = f < delta ? f : delta;
                    j = jb;
                }
                if (s_row < 0) {
                    sh_i[1] = -1;
                } else {
                    delta = ra[s_row] < delta ? ra[s_row] : delta;
                    for (int j = target;;) {
                        const int i = parC[j];
                        F[(size_t)i * K + j] += delta;
                        const int jb = parR[i];
                        if (jb < 0) break;
                        F[(size_t)i * K + jb] -= delta;
                        j = jb;
                    }
                    ra[s_row] -= delta; rb[target] -= delta;
                    sh_i[1] = 1;
                }
                __threadfence_block();
            }
            __syncthreads();
            if (sh_i[1] < 0) { trip = 3; break; }
            ++n_aug;
            __syncthreads();
        }
        // cost = sum F_ij M_ij
        double cost = 0.0;
        for (size_t e = tid; e < (size_t)K * K; e += EMDG_WG) { const double f = F[e]; if (f != 0.0) cost += f * M[e]; }
        cost = block_sum(cost, red_v);
        if (tid == 0) {
            p.emd[q] = trip ? __builtin_nan("") : cost;
            if (p.n_aug) p.n_aug[q] = trip ? -(n_aug * 8 + trip) : n_aug;
        }
        __syncthreads();
    }
}

// row minima of M (the initial row potentials), once per call
__global__ void emd_rowmin_kernel(const double *__restrict__ M, int K, double *__restrict__ rowmin) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        double m = __builtin_inf();
        for (int j = 0; j < K; ++j) { const double v = M[(size_t)i * K + j]; m = v < m ? v : m; }
        rowmin[i] = m;
    }
}

}  // namespace pilot
