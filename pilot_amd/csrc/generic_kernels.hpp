// Reference-semantics Sinkhorn kernel for the shapes the MFMA kernels do not take: K > 128 cell types, or a reg so small
// that exp(-max(M)/reg) leaves the f64 range (max(M)/reg > 600).  The reference has neither limit
// (pilotpy/tools/Trajectory.py:512-515 forwards any K x K cost and any reg to POT).
//
// One 256-thread workgroup per ordered pair runs POT 0.9.x sinkhorn_stabilized LITERALLY in fp64 -- including the
// log-absorption that the fast kernels only keep books of: when max(u, v) > tau the scalings are folded into the potentials
// (alpha, beta) and the pair's own kernel matrix K' = exp(-(M - alpha - beta) / reg) is rebuilt, so entries that underflow
// in the fixed Gibbs image come back (what ADVICE r01 asked for).  K' (and its transpose, for coalesced row products) lives
// in a per-workgroup global scratch that stays in L2; the vectors live in LDS.  Sums run in the oracle's order (row index
// ascending), so values agree with the CPU restatement to the last few ulps of exp / log.  Throughput is that of a
// vector-unit fp64 code with K^2 exponentials per absorption: a correct fallback, not a fast path.
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
};

constexpr int GENERIC_WG = 256;

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
    double *red = beta + K;                          // 8 doubles
    int *qs = reinterpret_cast<int *>(red + 8);
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
            // v = b / (K'^T u): column j summed over rows in ascending order (coalesced over j)
            for (int j = threadIdx.x; j < K; j += GENERIC_WG) {
                double s = 0.0;
                for (int i = 0; i < K; ++i) s += Km[(size_t)i * K + j] * u[i];
                v[j] = b[j] / s;
            }
            __syncthreads();
            // u = a / (K' v): row i summed over columns in ascending order (the transpose keeps the reads coalesced)
            for (int i = threadIdx.x; i < K; i += GENERIC_WG) {
                double s = 0.0;
                for (int j = 0; j < K; ++j) s += Kt[(size_t)j * K + i] * v[j];
                u[i] = a[i] / s;
            }
            __syncthreads();
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
                double e2 = 0.0;
                for (int j = threadIdx.x; j < K; j += GENERIC_WG) {
                    const double lv = log(v[j]);
                    double s = 0.0;
                    for (int i = 0; i < K; ++i) s += exp(-(p.M[(size_t)i * K + j] - alpha[i] - beta[j]) / reg + log(u[i]) + lv);
                    e2 += (s - b[j]) * (s - b[j]);
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
