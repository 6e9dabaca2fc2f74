/*
 * oracle/pilot_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C fp64 restatement of the arithmetic behind the reference hot path
 *   pilotpy/tools/Trajectory.py:479-523 (wasserstein_d), whose inner calls are
 *   ot.sinkhorn2(a, b, M, reg, method="sinkhorn_stabilized")   Trajectory.py:515
 *   ot.emd2(a, b, M)                                           Trajectory.py:511
 * The arithmetic itself lives in the third-party package POT
 * (pot>=0.9.1,<0.10.0, /root/reference/setup.py:19), which is NOT vendored in
 * the reference tree and NOT installed in the build container, so:
 *
 *   PARITY UNPINNED: the Sinkhorn control flow below is a restatement of the
 *   published POT 0.9.x algorithm (ot/bregman/_sinkhorn.py::sinkhorn_stabilized
 *   and ::sinkhorn2), not a diff against POT output.  What IS pinned:
 *     - converged Sinkhorn values against an independent log-domain solver
 *       (unique entropic optimum)            tests/test_oracle_sinkhorn.py
 *     - exact EMD values against scipy.optimize.linprog(HiGHS) (unique LP
 *       optimum value)                        tests/test_oracle_emd.py
 *     - the reference's own loop / layout, run here through a stub import
 *                                             tests/golden/gen_golden.py
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pilot_amd/) never does.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* info[] slots filled by the Sinkhorn oracle (all optional) */
enum {
    ORACLE_INFO_ITERS = 0,        /* number of (v,u) updates executed            */
    ORACLE_INFO_NABSORB = 1,      /* number of tau-absorptions                   */
    ORACLE_INFO_LAST_ABSORB = 2,  /* ii of the last absorption, -1 if none       */
    ORACLE_INFO_FLAGS = 3,        /* bit0: converged, bit1: NaN-revert,          */
                                  /* bit2: absorption fell on the final update,  */
                                  /* bit3: at least one absorption happened      */
    ORACLE_INFO_N = 4
};

ORACLE_API int pilot_oracle_version(void) { return 1; }

/* exp(-(M - alpha_i - beta_j)/reg): POT sinkhorn_stabilized.get_K */
static void get_K(const double *M, const double *alpha, const double *beta,
                  int na, int nb, double reg, double *K)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j)
            K[(size_t)i * nb + j] = exp(-(M[(size_t)i * nb + j] - alpha[i] - beta[j]) / reg);
}

/* exp(-(M - alpha_i - beta_j)/reg + log u_i + log v_j): POT get_Gamma */
static void get_Gamma(const double *M, const double *alpha, const double *beta,
                      const double *u, const double *v, int na, int nb,
                      double reg, double *G)
{
    for (int i = 0; i < na; ++i) {
        const double lu = log(u[i]);
        for (int j = 0; j < nb; ++j)
            G[(size_t)i * nb + j] =
                exp(-(M[(size_t)i * nb + j] - alpha[i] - beta[j]) / reg + lu + log(v[j]));
    }
}

/*
 * One ot.sinkhorn2(a, b, M, reg, method="sinkhorn_stabilized") call.
 *
 * POT 0.9.x sinkhorn_stabilized (single-histogram branch), defaults
 * numItermax=1000, tau=1e3, stopThr=1e-9, print_period=20:
 *
 *   alpha=beta=0; u=1/na; v=1/nb; K=get_K(alpha,beta); err=1
 *   for ii in range(numItermax):
 *       uprev,vprev=u,v
 *       v = b/(K.T@u); u = a/(K@v)
 *       if max|u|>tau or max|v|>tau:
 *           alpha+=reg*log(u); beta+=reg*log(v); u=1/na; v=1/nb; K=get_K(alpha,beta)
 *       if ii % print_period == 0:
 *           err = ||get_Gamma(alpha,beta,u,v).sum(0) - b||_2
 *       if err <= stopThr: break
 *       if any(isnan(u)) or any(isnan(v)): u,v=uprev,vprev; break
 *   return sum(M * get_Gamma(alpha,beta,u,v))          # ot.sinkhorn2
 *
 * legacy_loop != 0 selects the POT <= 0.7 `while loop:` form in which the
 * update at ii == numItermax still runs (numItermax+1 updates); kept only so
 * the iteration-cap sensitivity can be measured ("parity unpinned").
 *
 * Note (restated faithfully, not "fixed"): right after an absorption u,v are
 * reset to 1/na,1/nb rather than 1, so the plan evaluated in that same
 * iteration is scaled by 1/(na*nb) until the next update restores the scale.
 */
/* doubles of scratch one call needs (K, Gamma, alpha, u, uprev, beta, v, vprev) */
ORACLE_API size_t pilot_oracle_sinkhorn_ws_doubles(int na, int nb)
{
    return (size_t)2 * na * nb + 3 * (size_t)na + 3 * (size_t)nb;
}

/* The call itself on caller-provided scratch `buf` (pilot_oracle_sinkhorn_ws_doubles(na, nb) doubles): the pair loop keeps
 * one block per thread instead of a malloc/free per pair (the allocator serialised the OpenMP leg). */
static double sinkhorn2_stabilized_ws(
    const double *a, const double *b, const double *M, int na, int nb,
    double reg, int numItermax, double tau, double stopThr, int print_period,
    int legacy_loop, int *info, double *err_out, double *buf)
{
    double *K = buf, *G = K + (size_t)na * nb;
    double *alpha = G + (size_t)na * nb, *u = alpha + na, *uprev = u + na;
    double *beta = uprev + na, *v = beta + nb, *vprev = v + nb;
    for (int i = 0; i < na; ++i) { alpha[i] = 0.0; u[i] = 1.0 / na; }
    for (int j = 0; j < nb; ++j) { beta[j] = 0.0; v[j] = 1.0 / nb; }
    get_K(M, alpha, beta, na, nb, reg, K);

    double err = 1.0;
    int iters = 0, nabs = 0, last_abs = -1, flags = 0;
    const int last_ii = legacy_loop ? numItermax : numItermax - 1;
    for (int ii = 0; ii <= last_ii; ++ii) {
        memcpy(uprev, u, sizeof(double) * na);
        memcpy(vprev, v, sizeof(double) * nb);
        /* v = b / (K.T @ u) */
        for (int j = 0; j < nb; ++j) v[j] = 0.0;
        for (int i = 0; i < na; ++i) {
            const double ui = u[i];
            const double *Ki = K + (size_t)i * nb;
            for (int j = 0; j < nb; ++j) v[j] += Ki[j] * ui;
        }
        for (int j = 0; j < nb; ++j) v[j] = b[j] / v[j];
        /* u = a / (K @ v) */
        for (int i = 0; i < na; ++i) {
            const double *Ki = K + (size_t)i * nb;
            double s = 0.0;
            for (int j = 0; j < nb; ++j) s += Ki[j] * v[j];
            u[i] = a[i] / s;
        }
        iters = ii + 1;

        int has_nan = 0;
        double mu = 0.0, mv = 0.0;
        for (int i = 0; i < na; ++i) { if (isnan(u[i])) has_nan = 1; if (fabs(u[i]) > mu) mu = fabs(u[i]); }
        for (int j = 0; j < nb; ++j) { if (isnan(v[j])) has_nan = 1; if (fabs(v[j]) > mv) mv = fabs(v[j]); }

        /* np.max of an array holding a NaN is NaN and `NaN > tau` is False */
        if (!has_nan && (mu > tau || mv > tau)) {
            for (int i = 0; i < na; ++i) { alpha[i] += reg * log(u[i]); u[i] = 1.0 / na; }
            for (int j = 0; j < nb; ++j) { beta[j] += reg * log(v[j]); v[j] = 1.0 / nb; }
            get_K(M, alpha, beta, na, nb, reg, K);
            ++nabs; last_abs = ii;
        }
        if (ii % print_period == 0) {
            get_Gamma(M, alpha, beta, u, v, na, nb, reg, G);
            double e2 = 0.0;
            for (int j = 0; j < nb; ++j) {
                double s = 0.0;
                for (int i = 0; i < na; ++i) s += G[(size_t)i * nb + j];
                e2 += (s - b[j]) * (s - b[j]);
            }
            err = sqrt(e2);
        }
        if (err <= stopThr) { flags |= 1; break; }
        if (has_nan) {
            memcpy(u, uprev, sizeof(double) * na);
            memcpy(v, vprev, sizeof(double) * nb);
            flags |= 2;
            break;
        }
    }
    if (last_abs >= 0 && last_abs == iters - 1) flags |= 4;
    if (nabs > 0) flags |= 8;

    get_Gamma(M, alpha, beta, u, v, na, nb, reg, G);
    double val = 0.0;
    for (size_t t = 0; t < (size_t)na * nb; ++t) val += M[t] * G[t];
    if (info) {
        info[ORACLE_INFO_ITERS] = iters;
        info[ORACLE_INFO_NABSORB] = nabs;
        info[ORACLE_INFO_LAST_ABSORB] = last_abs;
        info[ORACLE_INFO_FLAGS] = flags;
    }
    if (err_out) *err_out = err;
    return val;
}

ORACLE_API double pilot_oracle_sinkhorn2_stabilized(
    const double *a, const double *b, const double *M, int na, int nb,
    double reg, int numItermax, double tau, double stopThr, int print_period,
    int legacy_loop, int *info, double *err_out)
{
    double *buf = (double *)malloc(sizeof(double) * pilot_oracle_sinkhorn_ws_doubles(na, nb));
    const double val = sinkhorn2_stabilized_ws(a, b, M, na, nb, reg, numItermax, tau, stopThr, print_period, legacy_loop,
                                               info, err_out, buf);
    free(buf);
    return val;
}

/*
 * The reference's pair loop, Trajectory.py:512-515, over rows
 * row_begin, row_begin+row_step, ... < row_end and ALL N columns (diagonal
 * included, no symmetry shortcut).  P is N x K row-major (one proportion
 * vector per sample, Trajectory.py:428-430), M is K x K (already divided by
 * its max, Trajectory.py:101).  emd receives one row of N values per selected
 * row; iters/err (nullable) have the same shape.  n_threads<=1: the
 * reference's single-threaded order; >1: OpenMP over pairs (bench baseline).
 */
/* stop_floor_ulps > 0 (NOT POT: the bench's equal-work leg only): stopThr is floored per pair at
 * stop_floor_ulps * FLT_EPSILON * ||b||_2, the f32 kernels' rule (sinkhorn_kernels.hpp, setup_body), so that the CPU runs
 * the update counts the f32 GPU kernels run.  0 = POT's rule. */
ORACLE_API int pilot_oracle_sinkhorn_grid_ex(
    const double *P, int N, int K, const double *M, double reg,
    int numItermax, double tau, double stopThr, int print_period, int legacy_loop, double stop_floor_ulps,
    int row_begin, int row_end, int row_step, int n_threads,
    double *emd, int *iters, double *err, int *flags)
{
    if (N <= 0 || K <= 0 || row_step <= 0 || row_begin < 0 || row_end > N) return -1;
    const int nrows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    const long total = (long)nrows * N;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        double *buf = (double *)malloc(sizeof(double) * pilot_oracle_sinkhorn_ws_doubles(K, K));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (long t = 0; t < total; ++t) {
            const int r = (int)(t / N), j = (int)(t % N);
            const int i = row_begin + r * row_step;
            int info[ORACLE_INFO_N];
            double e, thr = stopThr;
            if (stop_floor_ulps > 0.0) {
                double n2 = 0.0;
                for (int k = 0; k < K; ++k) n2 += P[(size_t)j * K + k] * P[(size_t)j * K + k];
                const double fl = stop_floor_ulps * 1.1920928955078125e-07 * sqrt(n2);
                if (fl > thr) thr = fl;
            }
            emd[t] = sinkhorn2_stabilized_ws(P + (size_t)i * K, P + (size_t)j * K, M, K, K,
                                             reg, numItermax, tau, thr, print_period,
                                             legacy_loop, info, &e, buf);
            if (iters) iters[t] = info[ORACLE_INFO_ITERS];
            if (err) err[t] = e;
            if (flags) flags[t] = info[ORACLE_INFO_FLAGS];
        }
        free(buf);
    }
    return 0;
}

ORACLE_API int pilot_oracle_sinkhorn_grid(
    const double *P, int N, int K, const double *M, double reg,
    int numItermax, double tau, double stopThr, int print_period, int legacy_loop,
    int row_begin, int row_end, int row_step, int n_threads,
    double *emd, int *iters, double *err, int *flags)
{
    return pilot_oracle_sinkhorn_grid_ex(P, N, K, M, reg, numItermax, tau, stopThr, print_period, legacy_loop, 0.0,
                                         row_begin, row_end, row_step, n_threads, emd, iters, err, flags);
}

/*
 * Exact optimal-transport cost, the value ot.emd2(a, b, M) returns
 * (Trajectory.py:511).  POT solves the transportation LP with a LEMON-derived
 * network simplex (ot/lp/EMD_wrapper.cpp, network_simplex_simple.h; not in the
 * reference tree).  The LP optimum VALUE is unique, so any exact method gives
 * the same number up to rounding; this oracle uses successive shortest paths
 * with node potentials on the dense bipartite residual graph (every
 * augmentation keeps complementary slackness, so the final flow is optimal),
 * after POT's own pre-step b *= sum(a)/sum(b) (ot/lp/__init__.py::emd2).
 * Validated against scipy.optimize.linprog(HiGHS) in tests/test_oracle_emd.py.
 * G (nullable, na x nb) receives the optimal plan.
 */
ORACLE_API double pilot_oracle_emd2(const double *a, const double *b_in, const double *M,
                                    int na, int nb, double *G_out)
{
    const int nn = na + nb;
    double *F = (double *)calloc((size_t)na * nb, sizeof(double));
    double *w = (double *)malloc(sizeof(double) * ((size_t)3 * nn + nb));
    int *iw = (int *)malloc(sizeof(int) * (size_t)2 * nn);
    double *pu = w, *pv = pu + na, *ra = pv + nb, *rb = ra + na;
    double *dist = rb + nb;            /* [0,na): rows, [na,nn): cols */
    double *b = dist + nn;
    int *parent = iw, *done = iw + nn; /* parent of a col = row index; of a row = col index */

    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < na; ++i) sa += a[i];
    for (int j = 0; j < nb; ++j) sb += b_in[j];
    for (int j = 0; j < nb; ++j) b[j] = b_in[j] * (sa / sb);
    for (int i = 0; i < na; ++i) {
        double m = M[(size_t)i * nb];
        for (int j = 1; j < nb; ++j) if (M[(size_t)i * nb + j] < m) m = M[(size_t)i * nb + j];
        pu[i] = m; ra[i] = a[i];
    }
    for (int j = 0; j < nb; ++j) { pv[j] = 0.0; rb[j] = b[j]; }
    const double tol = 1e-15 * (sa > 0 ? sa : 1.0);

    for (int s = 0; s < na; ++s) {
        while (ra[s] > tol) {
            for (int n = 0; n < nn; ++n) { dist[n] = INFINITY; parent[n] = -1; done[n] = 0; }
            dist[s] = 0.0;
            int target = -1;
            double dstar = 0.0;
            for (;;) {
                int best = -1; double bd = INFINITY;
                for (int n = 0; n < nn; ++n) if (!done[n] && dist[n] < bd) { bd = dist[n]; best = n; }
                if (best < 0) break;
                done[best] = 1;
                if (best >= na) {                     /* a column */
                    const int j = best - na;
                    if (rb[j] > 0.0) { target = j; dstar = bd; break; }
                    for (int i = 0; i < na; ++i) {    /* backward arcs j -> i where F_ij > 0 */
                        if (done[i] || F[(size_t)i * nb + j] <= 0.0) continue;
                        double rc = M[(size_t)i * nb + j] - pu[i] - pv[j];   /* == 0 up to rounding */
                        double nd = bd - rc; if (nd < bd) nd = bd;
                        if (nd < dist[i]) { dist[i] = nd; parent[i] = j; }
                    }
                } else {                              /* a row: forward arcs i -> j */
                    const int i = best;
                    for (int j = 0; j < nb; ++j) {
                        if (done[na + j]) continue;
                        double rc = M[(size_t)i * nb + j] - pu[i] - pv[j];
                        if (rc < 0.0) rc = 0.0;
                        const double nd = bd + rc;
                        if (nd < dist[na + j]) { dist[na + j] = nd; parent[na + j] = i; }
                    }
                }
            }
            if (target < 0) { ra[s] = 0.0; break; }   /* only rounding dust left */
            /* potentials: rc'(i,j) = rc + min(d_i,d*) - min(d_j,d*) >= 0 */
            for (int i = 0; i < na; ++i) pu[i] -= (dist[i] < dstar ? dist[i] : dstar);
            for (int j = 0; j < nb; ++j) pv[j] += (dist[na + j] < dstar ? dist[na + j] : dstar);
            /* bottleneck */
            double delta = ra[s] < rb[target] ? ra[s] : rb[target];
            for (int j = target;;) {
                const int i = parent[na + j];
                if (i == s) break;
                const int jb = parent[i];
                if (F[(size_t)i * nb + jb] < delta) delta = F[(size_t)i * nb + jb];
                j = jb;
            }
            for (int j = target;;) {
                const int i = parent[na + j];
                F[(size_t)i * nb + j] += delta;
                if (i == s) break;
                const int jb = parent[i];
                F[(size_t)i * nb + jb] -= delta;
                j = jb;
            }
            ra[s] -= delta; rb[target] -= delta;
        }
    }
    double cost = 0.0;
    for (size_t t = 0; t < (size_t)na * nb; ++t) cost += F[t] * M[t];
    if (G_out) memcpy(G_out, F, sizeof(double) * (size_t)na * nb);
    free(F); free(w); free(iw);
    return cost;
}

/*
 * The same LP value by a FASTER successive-shortest-path solver -- the algorithm of the HIP kernel (emd_kernels.hpp) restated
 * for one CPU thread: diagonal warm start on the zero-reduced-cost arcs (i, i), multi-source searches (from every row with
 * supply left) over column labels only (rows are reached over tight backward arcs and scanned at once), initial labels
 * A_j = min over sources of (M_ij - pu_i) kept while the source set does not change.  One augmentation per search.
 * Used as the CPU baseline of `bench.py --mode emd` (kind "port": POT's LEMON network simplex is not available here and
 * would be the reference's own solver); checked against pilot_oracle_emd2 and HiGHS in tests/test_oracle_emd.py.
 * Square problems only (na == nb == K), which is all the reference path produces.
 */
ORACLE_API double pilot_oracle_emd2_fast(const double *a, const double *b_in, const double *M, int K)
{
    double *F = (double *)calloc((size_t)K * K, sizeof(double));
    double *w = (double *)malloc(sizeof(double) * (size_t)8 * K);
    int *iw = (int *)malloc(sizeof(int) * (size_t)5 * K);
    double *pu = w, *pv = w + K, *ra = w + 2 * K, *rb = w + 3 * K, *A = w + 4 * K, *dC = w + 5 * K, *fR = w + 6 * K, *fC = w + 7 * K;
    int *Apar = iw, *parC = iw + K, *parR = iw + 2 * K, *openC = iw + 3 * K, *src = iw + 4 * K;
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < K; ++i) { sa += a[i]; sb += b_in[i]; }
    const double tol = 1e-15 * (sa > 0 ? sa : 1.0);
    for (int i = 0; i < K; ++i) {
        double m = M[(size_t)i * K];
        for (int j = 1; j < K; ++j) if (M[(size_t)i * K + j] < m) m = M[(size_t)i * K + j];
        pu[i] = m; pv[i] = 0.0; ra[i] = a[i]; rb[i] = b_in[i] * (sa / sb);
    }
    for (int i = 0; i < K; ++i)
        if (M[(size_t)i * K + i] - pu[i] == 0.0) {
            const double f = ra[i] < rb[i] ? ra[i] : rb[i];
            if (f > 0.0) { F[(size_t)i * K + i] = f; ra[i] -= f; rb[i] -= f; }
        }
    int n_src_prev = -1;
    for (int guard = 0; guard < 64 * K + 64; ++guard) {
        int n_src = 0, changed = 0;
        for (int i = 0; i < K; ++i) {
            const int is = ra[i] > tol;
            if (is != src[i] || n_src_prev < 0) changed = 1;
            src[i] = is; n_src += is;
        }
        if (!n_src) break;
        if (changed) {
            for (int j = 0; j < K; ++j) { A[j] = INFINITY; Apar[j] = -1; }
            for (int i = 0; i < K; ++i) {
                if (!src[i]) continue;
                const double *Mi = M + (size_t)i * K;
                for (int j = 0; j < K; ++j) { const double v = Mi[j] - pu[i]; if (v < A[j]) { A[j] = v; Apar[j] = i; } }
            }
        }
        n_src_prev = n_src;
        for (int j = 0; j < K; ++j) {
            double rc = A[j] - pv[j]; if (rc < 0.0) rc = 0.0;
            dC[j] = rc; fC[j] = INFINITY; parC[j] = Apar[j]; openC[j] = 1;
            fR[j] = src[j] ? 0.0 : INFINITY; parR[j] = -1;
        }
        int target = -1;
        double dstar = 0.0;
        for (;;) {
            int best = -1; double bd = INFINITY;
            for (int j = 0; j < K; ++j) if (openC[j] && dC[j] < bd) { bd = dC[j]; best = j; }
            if (best < 0) break;
            openC[best] = 0; fC[best] = bd;
            if (rb[best] > 0.0) { target = best; dstar = bd; break; }
            for (int i = 0; i < K; ++i) {          /* rows that ship to this column: reduced cost 0, scanned at once */
                if (fR[i] != INFINITY || F[(size_t)i * K + best] <= 0.0) continue;
                fR[i] = bd; parR[i] = best;
                const double *Mi = M + (size_t)i * K;
                for (int j = 0; j < K; ++j) {
                    if (!openC[j]) continue;
                    double rc = Mi[j] - pu[i] - pv[j]; if (rc < 0.0) rc = 0.0;
                    const double nd = bd + rc;
                    if (nd < dC[j]) { dC[j] = nd; parC[j] = i; }
                }
            }
        }
        if (target < 0) break;                      /* only rounding dust left */
        for (int i = 0; i < K; ++i) {
            pu[i] -= (fR[i] < dstar ? fR[i] : dstar);
            pv[i] += (fC[i] < dstar ? fC[i] : dstar);
        }
        int s_row = -1;
        double delta = rb[target];
        for (int j = target;;) {
            const int i = parC[j];
            const int jb = parR[i];
            if (jb < 0) { s_row = i; break; }
            if (F[(size_t)i * K + jb] < delta) delta = F[(size_t)i * K + jb];
            j = jb;
        }
        if (ra[s_row] < delta) delta = ra[s_row];
        for (int j = target;;) {
            const int i = parC[j];
            F[(size_t)i * K + j] += delta;
            const int jb = parR[i];
            if (jb < 0) break;
            F[(size_t)i * K + jb] -= delta;
            j = jb;
        }
        ra[s_row] -= delta; rb[target] -= delta;
    }
    double cost = 0.0;
    for (size_t t = 0; t < (size_t)K * K; ++t) cost += F[t] * M[t];
    free(F); free(w); free(iw);
    return cost;
}

/*
 * The same LP value by a NETWORK SIMPLEX on the bipartite transportation graph -- the algorithm family POT's ot.emd2 runs
 * (ot/lp/network_simplex_simple.h, LEMON-derived; not in the reference tree, not available here): na + nb nodes plus an
 * artificial root, a spanning-tree basis kept as parent pointers with the tree arc of every node stored AT the node
 * (orientation + flow; arcs outside the tree carry no flow because the problem has no capacities), block-search pricing over the
 * na * nb arcs, ratio test along the cycle with the strongly-feasible tie rule (first side strict, second side non-strict),
 * the cut-off subtree re-hung by reversing its parent pointers; depths and node potentials are then recomputed from the
 * parent pointers (O(na + nb) per pivot, dwarfed by pricing at these sizes).  Written from the published algorithm, for the
 * bench's CPU baseline of `--mode emd` (kind "port"): a network simplex is what a PILOT user's CPU runs in this mode, and it
 * is several times faster than the successive-shortest-path solvers above.  Checked against pilot_oracle_emd2 and HiGHS in
 * tests/test_oracle_emd.py.  ws: pilot_oracle_emd2_ns_ws_bytes(na, nb) bytes of scratch (NULL: allocated per call).
 */
ORACLE_API size_t pilot_oracle_emd2_ns_ws_bytes(int na, int nb)
{
    const size_t n = (size_t)na + nb + 1;
    return n * (3 * sizeof(double) + 5 * sizeof(int)) + (size_t)nb * sizeof(double) + 64;
}

ORACLE_API double pilot_oracle_emd2_ns(const double *a, const double *b_in, const double *M, int na, int nb, void *ws_in,
                                       int *n_pivots_out)
{
    const int n = na + nb, root = n;
    void *ws = ws_in ? ws_in : malloc(pilot_oracle_emd2_ns_ws_bytes(na, nb));
    double *flow = (double *)ws, *pi = flow + (n + 1), *acost = pi + (n + 1), *b = acost + (n + 1);
    int *parent = (int *)(b + nb), *up = parent + (n + 1), *depth = up + (n + 1), *stamp = depth + (n + 1), *stack = stamp + (n + 1);
    double sa = 0.0, sb = 0.0, maxc = 0.0;
    for (int i = 0; i < na; ++i) sa += a[i];
    for (int j = 0; j < nb; ++j) sb += b_in[j];
    for (int j = 0; j < nb; ++j) b[j] = b_in[j] * (sa / sb);          /* ot.emd2: b *= sum(a) / sum(b) */
    for (size_t t = 0; t < (size_t)na * nb; ++t) if (M[t] > maxc) maxc = M[t];
    const double ART = 2.0 * (n + 1) * maxc + 1.0;                     /* artificial arcs: dearer than any path of real ones */
    /* initial basis: every node hangs off the root on an artificial arc (supplies flow up, demands flow down) */
    for (int i = 0; i < na; ++i) { parent[i] = root; up[i] = 1; flow[i] = a[i]; acost[i] = ART; pi[i] = -ART; depth[i] = 1; }
    for (int j = 0; j < nb; ++j) { parent[na + j] = root; up[na + j] = 0; flow[na + j] = b[j]; acost[na + j] = ART; pi[na + j] = ART; depth[na + j] = 1; }
    parent[root] = -1; pi[root] = 0.0; depth[root] = 0; up[root] = 0; flow[root] = 0.0; acost[root] = 0.0;
    const long n_arcs = (long)na * nb;
    long block = (long)sqrt((double)n_arcs);
    if (block < 10) block = 10;
    const double eps = 1e-13 * (maxc > 0.0 ? maxc : 1.0);
    long next_arc = 0;
    int n_pivots = 0;
    const int max_pivots = 200 * (n + 1) + 1000;
    for (; n_pivots < max_pivots; ++n_pivots) {
        /* pricing: the most negative reduced cost of the first block (cyclic scan) that holds a negative one */
        long in_arc = -1;
        double best = -eps;
        {
            long scanned = 0, in_block = 0, e = next_arc;
            int i = (int)(e / nb), j = (int)(e % nb);
            while (scanned < n_arcs) {
                /* one row segment at a time: pi[i] and the row of M stay in registers, no division per arc */
                long seg = nb - j;
                if (seg > n_arcs - scanned) seg = n_arcs - scanned;
                if (seg > block - in_block) seg = block - in_block;
                const double pii = pi[i];
                const double *Mi = M + (size_t)i * nb, *pj = pi + na;
                for (long k = 0; k < seg; ++k) {
                    const double rc = Mi[j + k] + pii - pj[j + k];
                    if (rc < best) { best = rc; in_arc = e + k; }
                }
                scanned += seg; in_block += seg; e += seg; j += (int)seg;
                if (j == nb) { j = 0; if (++i == na) { i = 0; e = 0; } }
                if (in_block == block) {
                    if (in_arc >= 0) { next_arc = e; break; }
                    in_block = 0;
                }
            }
        }
        if (in_arc < 0) break;                                         /* optimal */
        const int u = (int)(in_arc / nb), v = na + (int)(in_arc % nb); /* entering arc u -> v */
        /* join node of the cycle */
        int x = u, y = v;
        while (x != y) { if (depth[x] >= depth[y]) x = parent[x]; else y = parent[y]; }
        const int join = x;
        /* ratio test: delta flows u -> v, v up to the join, the join down to u.  On u's side it runs parent -> child, so arcs
         * oriented child -> parent lose flow; on v's side it runs child -> parent, so arcs oriented parent -> child lose flow */
        double delta = INFINITY;
        int u_out = -1, side = 0;
        for (x = u; x != join; x = parent[x]) if (up[x] && flow[x] < delta) { delta = flow[x]; u_out = x; side = 1; }
        for (x = v; x != join; x = parent[x]) if (!up[x] && flow[x] <= delta) { delta = flow[x]; u_out = x; side = 2; }
        if (u_out < 0) break;                                          /* unbounded: cannot happen with costs >= 0 */
        for (x = u; x != join; x = parent[x]) flow[x] += up[x] ? -delta : delta;
        for (x = v; x != join; x = parent[x]) flow[x] += up[x] ? delta : -delta;
        /* the subtree below the leaving arc is re-hung on the entering arc: parent pointers reversed from the entering arc's
         * end node in that subtree up to u_out; the tree arc of every node on that path moves to its old parent */
        {
            int q = side == 1 ? u : v;                                 /* new root of the cut-off subtree */
            int prev = side == 1 ? v : u, prev_up = side == 1 ? 1 : 0;
            double prev_flow = delta, prev_cost = M[in_arc];
            for (;;) {
                const int old_p = parent[q], old_up = up[q];
                const double old_flow = flow[q], old_cost = acost[q];
                parent[q] = prev; up[q] = prev_up; flow[q] = prev_flow; acost[q] = prev_cost;
                if (q == u_out) break;
                prev = q; prev_up = !old_up; prev_flow = old_flow; prev_cost = old_cost;
                q = old_p;
            }
        }
        /* depths and potentials from the parent pointers (reduced cost 0 on every tree arc: cost + pi[src] - pi[dst] = 0) */
        for (x = 0; x < n; ++x) stamp[x] = 0;
        for (int s0 = 0; s0 < n; ++s0) {
            int len = 0;
            for (x = s0; x != root && !stamp[x]; x = parent[x]) stack[len++] = x;   /* up to the first resolved ancestor */
            while (len > 0) {                                                      /* ... and back down */
                const int z = stack[--len], p = parent[z];
                depth[z] = depth[p] + 1;
                pi[z] = up[z] ? pi[p] - acost[z] : pi[p] + acost[z];
                stamp[z] = 1;
            }
        }
    }
    double cost = 0.0;
    for (int x = 0; x < n; ++x)
        if (parent[x] != root && parent[x] >= 0) cost += flow[x] * acost[x];   /* real tree arcs (artificial ones end at ~0 flow) */
    if (n_pivots_out) *n_pivots_out = n_pivots;
    if (!ws_in) free(ws);
    return cost;
}

ORACLE_API int pilot_oracle_emd_grid_ns(const double *P, int N, int K, const double *M,
                                        int row_begin, int row_end, int row_step, int n_threads, double *emd)
{
    if (N <= 0 || K <= 0 || row_step <= 0 || row_begin < 0 || row_end > N) return -1;
    const int nrows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    const long total = (long)nrows * N;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        void *ws = malloc(pilot_oracle_emd2_ns_ws_bytes(K, K));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (long t = 0; t < total; ++t) {
            const int i = row_begin + (int)(t / N) * row_step, j = (int)(t % N);
            emd[t] = pilot_oracle_emd2_ns(P + (size_t)i * K, P + (size_t)j * K, M, K, K, ws, NULL);
        }
        free(ws);
    }
    return 0;
}

/* The reference's exact pair loop, Trajectory.py:507-511. */
ORACLE_API int pilot_oracle_emd_grid_fast(const double *P, int N, int K, const double *M,
                                          int row_begin, int row_end, int row_step, int n_threads, double *emd)
{
    if (N <= 0 || K <= 0 || row_step <= 0 || row_begin < 0 || row_end > N) return -1;
    const int nrows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    const long total = (long)nrows * N;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
#endif
    for (long t = 0; t < total; ++t) {
        const int i = row_begin + (int)(t / N) * row_step, j = (int)(t % N);
        emd[t] = pilot_oracle_emd2_fast(P + (size_t)i * K, P + (size_t)j * K, M, K);
    }
    (void)n_threads;
    return 0;
}

ORACLE_API int pilot_oracle_emd_grid(const double *P, int N, int K, const double *M,
                                     int row_begin, int row_end, int row_step, int n_threads,
                                     double *emd)
{
    if (N <= 0 || K <= 0 || row_step <= 0 || row_begin < 0 || row_end > N) return -1;
    const int nrows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    const long total = (long)nrows * N;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
#endif
    for (long t = 0; t < total; ++t) {
        const int i = row_begin + (int)(t / N) * row_step, j = (int)(t % N);
        emd[t] = pilot_oracle_emd2(P + (size_t)i * K, P + (size_t)j * K, M, K, K, NULL);
    }
    (void)n_threads;
    return 0;
}

/*
 * Cell-level W2 EXTENSION (not in the reference; BASELINE config 5): entropic OT between two point clouds with uniform weights
 * and cost |x - y|^2 / scale, POT 0.9.x ot.bregman.sinkhorn_log control flow (log-domain updates v then u, marginal error
 * every check_period updates, stop on err < stopThr), value <Gamma, C>.  The C twin of oracle.py::cell_w2 (numpy), so that
 * converged pairs of thousands of cells -- hundreds of updates over an n x m matrix -- are affordable in a test.
 * OpenMP over the rows / columns of a pass; every log-sum-exp is computed by ONE thread in index order (max-shifted like
 * scipy.special.logsumexp), the error norm is summed in index order: the result does not depend on the thread count.
 */
ORACLE_API double pilot_oracle_cell_w2(const double *X, int n, const double *Y, int m, int D, double scale, double reg,
                                       int numItermax, double stopThr, int check_period, int n_threads, int *iters_out,
                                       double *err_out)
{
    double *Mr = (double *)malloc(sizeof(double) * (size_t)n * m);      /* -C / reg */
    double *u = (double *)calloc((size_t)n, sizeof(double)), *v = (double *)calloc((size_t)m, sizeof(double));
    double *col = (double *)malloc(sizeof(double) * (size_t)m);
    if (n_threads < 1) n_threads = 1;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) { const double t = X[(size_t)i * D + d] - Y[(size_t)j * D + d]; s += t * t; }
            Mr[(size_t)i * m + j] = -(s / scale) / reg;
        }
    const double loga = -log((double)n), logb = -log((double)m), b = 1.0 / m;
    double err = 1.0;
    int iters = 0;
    for (int ii = 0; ii < numItermax; ++ii) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
        for (int j = 0; j < m; ++j) {                 /* v = logb - logsumexp(Mr + u[:, None], axis=0) */
            double mx = -INFINITY;
            for (int i = 0; i < n; ++i) { const double t = Mr[(size_t)i * m + j] + u[i]; if (t > mx) mx = t; }
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += exp(Mr[(size_t)i * m + j] + u[i] - mx);
            v[j] = logb - (log(s) + mx);
        }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
        for (int i = 0; i < n; ++i) {                 /* u = loga - logsumexp(Mr + v[None, :], axis=1) */
            const double *row = Mr + (size_t)i * m;
            double mx = -INFINITY;
            for (int j = 0; j < m; ++j) { const double t = row[j] + v[j]; if (t > mx) mx = t; }
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += exp(row[j] + v[j] - mx);
            u[i] = loga - (log(s) + mx);
        }
        iters = ii + 1;
        if (ii % check_period == 0) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
            for (int j = 0; j < m; ++j) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += exp(Mr[(size_t)i * m + j] + u[i] + v[j]);
                col[j] = s - b;
            }
            double e2 = 0.0;
            for (int j = 0; j < m; ++j) e2 += col[j] * col[j];
            err = sqrt(e2);
            if (err < stopThr) break;
        }
    }
    double val = 0.0;
    for (int i = 0; i < n; ++i) {                     /* sum(exp(Mr + u + v) * C), C = -reg * Mr */
        const double *row = Mr + (size_t)i * m;
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += exp(row[j] + u[i] + v[j]) * (-reg * row[j]);
        val += s;
    }
    if (iters_out) *iters_out = iters;
    if (err_out) *err_out = err;
    free(Mr); free(u); free(v); free(col);
    return val;
}
