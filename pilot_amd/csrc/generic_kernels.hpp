// Reference-semantics Sinkhorn kernel for the shapes the MFMA kernels do not take: K > 128 cell types, or a reg so small
// that exp(-max(M)/reg) leaves the f64 range (max(M)/reg > 600).  The reference has neither limit
// (pilotpy/tools/Trajectory.py:512-515 forwards any K x K cost and any reg to POT).
//
// One 256-thread workgroup per ordered pair runs POT 0.9.x sinkhorn_stabilized LITERALLY in fp64 -- including the
// log-absorption that the fast kernels only keep books of: when max(u, v) > tau the scalings are folded into the potentials
// (alpha, beta) and the pair's own kernel matrix K' = exp(-(M - alpha - beta) / reg) is rebuilt, so entries that underflow
// in the fixed Gibbs image come back (what ADVICE r01 asked for).  K' (and its transpose, for coalesced row products) lives
// in a per-workgroup global scratch that stays in L2; the vectors live in LDS.  A matrix-vector product puts the outputs on
// the lanes (coalesced rows of K' / K'^T, four outputs per lane at a time) and the contraction range split over up to sixteen waves
// (as many as LDS has rows of partial sums for); the partial sums of an output are added in wave order, so values agree with the CPU restatement (row index
// ascending) to a few ulps and are the same on every run.  (Round 3: before, one thread per output summed its K terms
// serially -- a load round trip per term, K of the 256 threads working: 72 us per update at K = 130; lanes along the
// contraction with a shuffle tree per output was no better: 33 dependent load + reduce rounds per wave.)  Throughput is
// that of a vector-unit fp64 code with K^2 exponentials per absorption: a correct fallback, not a fast path.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

struct GenericParams {
    const double *P, *M;       // N x K proportions (caller's layout), K x K cost
    int N, K;
    int n_pairs, row_begin, row_step;
    double reg, tau, stop_thr;
    int max_iter, period;
    double *emd; int *iters; double *err; int *flags;      // indexed by q = local_row * N + j (iters / err nullable)
    double *kws;               // per workgroup: K' (K x K, row-major) followed by its transpose
    int *queue;                // dynamic pair queue (zeroed by the host)
    const int *list;           // nullable: explicit work-item list (pairs another kernel handed over), length *list_len
    const int *list_len;
    int nsplit;                // waves that share one output block of a matrix-vector product (power of two <= 16; LDS holds
                               // nsplit rows of K partial sums)
};

#ifndef PILOT_GENERIC_WG
#define PILOT_GENERIC_WG 1024
#endif
constexpr int GENERIC_WG = PILOT_GENERIC_WG, GENERIC_WAVES = GENERIC_WG / 64;     // (sixteen waves: a sixteenth of the contraction range each)

__device__ inline double block_reduce_sum(double x, double *red) {
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < GENERIC_WG / 64; ++w) s += red[w];       // fixed order
    return s;
}
__device__ inline double wave_tree_sum(double x) {      // every lane gets the sum of the 64, always added in the same order
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__device__ inline double block_reduce_max(double x, double *red) {
    for (int off = 32; off >= 1; off >>= 1) { const double y = __shfl_xor(x, off); x = y > x ? y : x; }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < GENERIC_WG / 64; ++w) s = red[w] > s ? red[w] : s;
    return s;
}

static __global__ void __launch_bounds__(GENERIC_WG) sinkhorn_generic_kernel(GenericParams p) {
    extern __shared__ double sm[];
    const int K = p.K, N = p.N;
    double *a = sm, *b = a + K, *u = b + K, *v = u + K, *up = v + K, *vp = up + K, *alpha = vp + K, *beta = alpha + K;
    double *red = beta + K;                          // GENERIC_WAVES doubles
    double *part = red + GENERIC_WAVES;              // GENERIC_WAVES x K partial sums of a matrix-vector product
    int *qs = reinterpret_cast<int *>(part + p.nsplit * K);
    double *Km = p.kws + (size_t)blockIdx.x * 2 * K * K, *Kt = Km + (size_t)K * K;
    const double reg = p.reg;
    auto build_kernel = [&]() {                      // K' = exp(-(M - alpha_i - beta_j) / reg): POT get_K
        for (int t = threadIdx.x; t < K * K; t += GENERIC_WG) {
            const int i = t / K, j = t % K;
            const double val = exp(-(p.M[t] - alpha[i] - beta[j]) / reg);
            Km[t] = val;
            Kt[(size_t)j * K + i] = val;
        }
        __threadfence_block();
        __syncthreads();
    };
    for (;;) {
        if (threadIdx.x == 0) qs[0] = atomicAdd(p.queue, 1);
        __syncthreads();
        const int item = qs[0];
        __syncthreads();
        if (item >= (p.list_len ? *p.list_len : p.n_pairs)) break;
        const int q = p.list ? p.list[item] : item;
        const int i_s = p.row_begin + (q / N) * p.row_step, j_s = q % N;
        for (int k = threadIdx.x; k < K; k += GENERIC_WG) {
            a[k] = p.P[(size_t)i_s * K + k]; b[k] = p.P[(size_t)j_s * K + k];
            alpha[k] = 0.0; beta[k] = 0.0; u[k] = 1.0 / K; v[k] = 1.0 / K;
        }
        __syncthreads();
        build_kernel();
        double err = 1.0;
        int iters = 0, nabs = 0, last_abs = -1, flags = 0;
        for (int ii = 0; ii < p.max_iter; ++ii) {
            for (int k = threadIdx.x; k < K; k += GENERIC_WG) { up[k] = u[k]; vp[k] = v[k]; }
            __syncthreads();
            const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
            const int ns = p.nsplit, sw = wave % ns, og = wave / ns, n_og = GENERIC_WAVES / ns;
            const int c0 = (K * sw) / ns, c1 = (K * (sw + 1)) / ns;       // this wave's part of the contraction range
            // v = b / (K'^T u): outputs j on the lanes (row i of K' is contiguous over j), rows c0 .. c1 on this wave; four
            // outputs per lane at a time: 32 independent loads in flight
            auto product = [&](const double *X, const double *x, const double *num, double *out) {
                for (int j0 = lane + 256 * og; j0 < K; j0 += 256 * n_og) {
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    const bool h1 = j0 + 64 < K, h2 = j0 + 128 < K, h3 = j0 + 192 < K;
#pragma unroll 8
                    for (int i = c0; i < c1; ++i) {
                        const double xi = x[i];
                        const double *row = X + (size_t)i * K + j0;
                        s0 += row[0] * xi;
                        if (h1) s1 += row[64] * xi;
                        if (h2) s2 += row[128] * xi;
                        if (h3) s3 += row[192] * xi;
                    }
                    part[sw * K + j0] = s0;
                    if (h1) part[sw * K + j0 + 64] = s1;
                    if (h2) part[sw * K + j0 + 128] = s2;
                    if (h3) part[sw * K + j0 + 192] = s3;
                }
                __syncthreads();
                for (int j = threadIdx.x; j < K; j += GENERIC_WG) {
                    double t = part[j];
                    for (int w = 1; w < ns; ++w) t += part[w * K + j];          // wave order
                    out[j] = num[j] / t;
                }
                __syncthreads();
            };
            product(Km, u, b, v);
            // u = a / (K' v): outputs i on the lanes (row j of K'^T is contiguous over i)
            product(Kt, v, a, u);
            iters = ii + 1;
            double mu = 0.0, mv = 0.0, nanf = 0.0;
            for (int k = threadIdx.x; k < K; k += GENERIC_WG) {
                if (u[k] != u[k] || v[k] != v[k]) nanf = 1.0;
                mu = fabs(u[k]) > mu ? fabs(u[k]) : mu;
                mv = fabs(v[k]) > mv ? fabs(v[k]) : mv;
            }
            const bool has_nan = block_reduce_max(nanf, red) > 0.0;
            mu = block_reduce_max(mu, red);
            mv = block_reduce_max(mv, red);
            if (!has_nan && (mu > p.tau || mv > p.tau)) {      // POT: absorb the scalings into the potentials
                for (int k = threadIdx.x; k < K; k += GENERIC_WG) {
                    alpha[k] += reg * log(u[k]); u[k] = 1.0 / K;
                    beta[k] += reg * log(v[k]); v[k] = 1.0 / K;
                }
                __syncthreads();
                build_kernel();
                ++nabs; last_abs = ii;
            }
            if (ii % p.period == 0) {                          // || Gamma^T 1 - b ||_2 with Gamma = get_Gamma(alpha, beta, u, v)
                double e2 = 0.0;        // (lane 0 of every wave carries the wave's columns)
                for (int j = wave; j < K; j += GENERIC_WG / 64) {
                    const double lv = log(v[j]);
                    double s = 0.0;
                    for (int i = lane; i < K; i += 64) s += exp(-(p.M[(size_t)i * K + j] - alpha[i] - beta[j]) / reg + log(u[i]) + lv);
                    s = wave_tree_sum(s);
                    if (lane == 0) e2 += (s - b[j]) * (s - b[j]);
                }
                err = sqrt(block_reduce_sum(e2, red));
            }
            if (err <= p.stop_thr) { flags |= 1; break; }
            if (has_nan) {                                      // POT: "Numerical errors": back to the last good iterate
                for (int k = threadIdx.x; k < K; k += GENERIC_WG) { u[k] = up[k]; v[k] = vp[k]; }
                __syncthreads();
                flags |= 2;
                break;
            }
        }
        if (last_abs >= 0 && last_abs == iters - 1) flags |= 4;
        if (nabs > 0) flags |= 8;
        // ot.sinkhorn2: sum(M * Gamma)
        double val = 0.0;
        for (int t = threadIdx.x; t < K * K; t += GENERIC_WG) {
            const int i = t / K, j = t % K;
            val += p.M[t] * exp(-(p.M[t] - alpha[i] - beta[j]) / reg + log(u[i]) + log(v[j]));
        }
        val = block_reduce_sum(val, red);
        if (threadIdx.x == 0) {
            p.emd[q] = val;
            if (p.iters) p.iters[q] = iters;
            if (p.err) p.err[q] = err;
            p.flags[q] = flags | 16;       // FLAG_F64
        }
        __syncthreads();
    }
}

}  // namespace pilot
