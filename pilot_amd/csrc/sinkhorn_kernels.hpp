// Batched Sinkhorn pair-grid kernels for gfx950 (MI355X).  Device code only; the C ABI is in
// pilot_ot.hip.  Replaces the per-pair POT loop of pilotpy/tools/Trajectory.py:512-515.
//
// Mapping (see DESIGN.md "Kernel K2"):
//   * one wavefront solves a tile of TILE ordered pairs at once (TILE = 32 in f32, 16 in f64);
//     the scalings u, v of all pairs of the tile form K x TILE panels and one Sinkhorn update is
//     two panel products  G^T U  and  G V  with the SHARED K x K Gibbs kernel G = exp(-M/reg);
//   * G is the stationary MFMA "A" operand, pre-arranged once per launch in LDS in exactly the
//     lane order the instruction wants (conflict-free, one ds_read per MFMA);
//   * the result tile of v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 has its column (= pair) on
//     the lane and its rows (= cell types) in the accumulator registers, so it is fed back as the
//     "B" operand of the next product with NO lane movement and NO LDS round trip: k-step (t', r)
//     consumes accumulator register r of row-tile t' and the LDS image stores G with its k index
//     permuted to match (Mfma<T>::row_of);
//   * element-wise work (v = b / G^T u, u = a / G v, marginal error, tau tracking) happens in that
//     same register layout; per-pair reductions over cell types are in-register sums plus one or
//     two cross-lane xor-shuffles (the lane groups holding the same column);
//   * every pair keeps POT's control flow: v first, then u; error checked when ii % period == 0;
//     a converged pair is frozen (predicated updates) while its tile mates continue.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

constexpr int WAVE = 64;
constexpr int WAVES_PER_WG = 4;

// flag bits, identical to include/pilot_ot.h
constexpr int FLAG_CONVERGED = 1, FLAG_NAN = 2, FLAG_ABSORB_LAST = 4, FLAG_ABSORBED = 8, FLAG_F64 = 16;
constexpr int FLAG_NEEDS_TRACK = 1 << 30;  // internal: fast kernel hands the pair to the tracking kernel

template <typename T> struct Mfma;

template <> struct Mfma<float> {
    static constexpr int TILE = 32;  // rows per accumulator tile == pairs per wave
    static constexpr int NREG = 16;  // accumulator registers per tile
    static constexpr int NGRP = 2;   // lane groups (64 / TILE) == k per MFMA
    using acc_t = float __attribute__((ext_vector_type(16)));
    // row of accumulator register r held by lane group g (cdna guide: row=(reg&3)+8*(reg>>2)+4*(lane>>5))
    __host__ __device__ static constexpr int row_of(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float eps() { return 1.1920929e-07f; }
};

template <> struct Mfma<double> {
    static constexpr int TILE = 16;
    static constexpr int NREG = 4;
    static constexpr int NGRP = 4;
    using acc_t = double __attribute__((ext_vector_type(4)));
    // f64 MFMA uses its own C/D map: row = (lane>>4) + 4*reg
    __host__ __device__ static constexpr int row_of(int r, int g) { return g + 4 * r; }
    __device__ static inline acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __device__ static inline double rcp(double x) { return 1.0 / x; }
    __device__ static inline double eps() { return 2.220446049250313e-16; }
};

// smallest row any lane group holds in register r
template <typename T> __host__ __device__ constexpr int row_min(int r) { return Mfma<T>::row_of(r, 0); }

// index of the LDS/global "A image" element read by `lane` for k-step (tp, r) and output row-tile t
template <typename T>
__host__ __device__ constexpr int img_index(int RT, int tp, int r, int t, int lane) {
    return (((tp * Mfma<T>::NREG + r) * RT + t) * WAVE) + lane;
}

template <typename T> __device__ inline T group_sum(T x) {
    // sum over the lane groups that hold the same column (lane % TILE)
    if constexpr (Mfma<T>::NGRP == 4) x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}
template <typename T> __device__ inline T group_max(T x) {
    if constexpr (Mfma<T>::NGRP == 4) { T y = __shfl_xor(x, 16); x = x > y ? x : y; }
    T y = __shfl_xor(x, 32);
    return x > y ? x : y;
}
template <typename T> __device__ inline T abs_t(T x) { return x < T(0) ? -x : x; }

// Y[t] = sum over k of X_img[out row][k] * IN[k]   for the whole K x TILE panel
template <typename T, int RT>
__device__ inline void panel_product(const T *__restrict__ img, const typename Mfma<T>::acc_t (&IN)[RT],
                                     typename Mfma<T>::acc_t (&OUT)[RT], int K, int lane) {
    using M = Mfma<T>;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) OUT[t][r] = T(0);
    // k-steps of row-tiles tp < RT-1 are always inside K (RT = ceil(K / TILE)); only the last tile
    // has padding, skipped per chunk of CH registers with a wave-uniform branch.
    constexpr int CH = (M::NREG >= 16) ? 4 : 1;
#pragma unroll
    for (int tp = 0; tp < RT; ++tp) {
#pragma unroll
        for (int c = 0; c < M::NREG / CH; ++c) {
            if (tp < RT - 1 || tp * M::TILE + row_min<T>(c * CH) < K) {
#pragma unroll
                for (int rr = 0; rr < CH; ++rr) {
                    const int r = c * CH + rr;
                    const T x = IN[tp][r];
#pragma unroll
                    for (int t = 0; t < RT; ++t)
                        OUT[t] = M::mfma(img[img_index<T>(RT, tp, r, t, lane)], x, OUT[t]);
                }
            }
        }
    }
}

struct GridParams {
    const void *P;       // N x K, element type T
    const void *img;     // 3 images of KP*KP elements of T: G^T-form, G-form, (G o M)-form
    int N, K;
    int n_pairs;         // number of work items
    const int *list;     // nullable: explicit work-item list (indices into the output arrays)
    const int *list_len; // nullable: device-side length of `list` (overrides n_pairs)
    int row_begin, row_step;
    int max_iter, period;
    double stop_thr, tau, floor_ulps;
    double *emd;         // outputs indexed by work item q = local_row * N + j
    int *iters;
    double *err;
    int *flags;
    int *track_list;     // fast kernel appends pairs that need POT absorption tracking
    int *track_count;
};

// TRACK = false: plain scaling iterations; a pair whose POT residual scaling would exceed tau
//                (i.e. POT would absorb) is handed to the TRACK = true kernel via track_list.
// TRACK = true : additionally carries the reciprocal reference scalings so the iteration at which
//                every POT absorption happens is known (needed for POT's err-after-absorption and
//                plan/(K*K)-on-the-final-update behaviour; see oracle/pilot_oracle.c).
// live panel registers per lane: A, B, U, V, ACC (+ RU, RV when tracking)
template <typename T, int RT, bool TRACK> constexpr int min_waves_per_simd() {
    return ((TRACK ? 7 : 5) * RT * Mfma<T>::NREG * int(sizeof(T) / 4) + 64 <= 256) ? 2 : 1;
}

template <typename T, int RT, bool SYM, bool TRACK>
__global__ void __launch_bounds__(WAVE * WAVES_PER_WG, (min_waves_per_simd<T, RT, TRACK>()))
sinkhorn_grid_kernel(GridParams p) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    constexpr int TILE = M::TILE, NREG = M::NREG;
    constexpr int KP = RT * TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw);

    const int n_items = p.list_len ? *p.list_len : p.n_pairs;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const int first_tile = blockIdx.x * WAVES_PER_WG;
    if (first_tile * TILE >= n_items) return;  // whole workgroup idle (tracking launch sized for the worst case)

    // stage the stationary operand: image 0 (and image 1 unless G is symmetric) into LDS
    {
        const T *g = static_cast<const T *>(p.img);
        constexpr int n = (SYM ? 1 : 2) * KP * KP;
        for (int i = threadIdx.x; i < n; i += WAVE * WAVES_PER_WG) lds[i] = g[i];
    }
    __syncthreads();
    const T *img_gt = lds;                         // out = G^T in
    const T *img_g = SYM ? lds : lds + KP * KP;    // out = G in
    const T *img_gm = static_cast<const T *>(p.img) + 2 * KP * KP;  // out = (G o M) in, read once from L2

    const int tile = first_tile + wave;
    if (tile * TILE >= n_items) return;
    const int col = lane % TILE, grp = lane / TILE;
    const int K = p.K, N = p.N;
    int item = tile * TILE + col;
    const bool live = item < n_items;
    if (!live) item = n_items - 1;                 // tail lanes shadow a valid pair, never store
    const int q = p.list ? p.list[item] : item;
    const int i = p.row_begin + (q / N) * p.row_step, j = q % N;

    const T *Pt = static_cast<const T *>(p.P);
    acc_t A[RT], B[RT], U[RT], V[RT], ACC[RT];
    acc_t RU[TRACK ? RT : 1], RV[TRACK ? RT : 1];
    const T uinit = T(1) / T(K);
    T bnorm2 = T(0);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const int row = t * TILE + M::row_of(r, grp);
            const bool ok = row < K;
            A[t][r] = ok ? Pt[(size_t)i * K + row] : T(0);
            B[t][r] = ok ? Pt[(size_t)j * K + row] : T(0);
            U[t][r] = ok ? uinit : T(0);
            V[t][r] = ok ? uinit : T(0);
            bnorm2 += B[t][r] * B[t][r];
            if constexpr (TRACK) { RU[t][r] = ok ? T(1) : T(0); RV[t][r] = ok ? T(1) : T(0); }
        }
    bnorm2 = group_sum<T>(bnorm2);
    T thr = T(p.stop_thr);
    if constexpr (sizeof(T) == 4) {
        const T fl = T(p.floor_ulps) * M::eps() * sqrtf(bnorm2);
        thr = thr > fl ? thr : fl;
    }
    const T tau = T(p.tau);
    const T kk = T(K) * T(K);
    const int krem = K - (RT - 1) * TILE;  // valid rows in the last row-tile

    bool done = false;
    int iters = 0, flags = 0, abs_at = -1;
    T errv = T(1);

    for (int ii = 0; ii < p.max_iter; ++ii) {
        panel_product<T, RT>(img_gt, U, ACC, K, lane);  // ACC = G^T u
        if (ii > 0 && (ii - 1) % p.period == 0) {
            // POT evaluates the marginal error of iteration ii-1 from (u_new, v): Gamma^T 1 = v o (G^T u_new)
            T e2 = T(0);
            T sc = T(1);
            if constexpr (TRACK) sc = (abs_at == ii - 1) ? T(1) / kk : T(1);  // u,v were reset to 1/K each
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const T d = V[t][r] * ACC[t][r] * sc - B[t][r];
                    e2 += d * d;
                }
            e2 = group_sum<T>(e2);
            const T e = sqrt(e2);
            if (!done) {
                errv = e;
                if (e <= thr) { done = true; iters = ii; flags |= FLAG_CONVERGED; }
                else if (e != e) { done = true; iters = ii; flags |= FLAG_NAN; }  // POT: "Numerical errors"
            }
        }
        if (__ballot(!done) == 0ull) break;

        // v = b / (G^T u)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                T vn = B[t][r] * M::rcp(ACC[t][r]);
                if (t == RT - 1 && M::row_of(r, grp) >= krem) vn = T(0);  // padded rows stay 0
                V[t][r] = done ? V[t][r] : vn;
            }
        panel_product<T, RT>(img_g, V, ACC, K, lane);   // ACC = G v
        T mx = T(0);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                T un = A[t][r] * M::rcp(ACC[t][r]);
                if (t == RT - 1 && M::row_of(r, grp) >= krem) un = T(0);
                U[t][r] = done ? U[t][r] : un;
                T mu, mv;
                if constexpr (TRACK) { mu = abs_t(U[t][r] * RU[t][r]); mv = abs_t(V[t][r] * RV[t][r]); }
                else { mu = abs_t(U[t][r]); mv = abs_t(V[t][r]); }
                mx = mx > mu ? mx : mu;
                mx = mx > mv ? mx : mv;
            }
        mx = group_max<T>(mx);
        if (!done) {
            if (mx > tau) {
                if constexpr (TRACK) {
                    // POT: alpha += reg log u, beta += reg log v, u = 1/K, v = 1/K.  In total-scaling
                    // terms: new references u_ref = u*K, v_ref = v/K; stored u, v unchanged.
#pragma unroll
                    for (int t = 0; t < RT; ++t)
#pragma unroll
                        for (int r = 0; r < NREG; ++r) {
                            const bool pad = (t == RT - 1 && M::row_of(r, grp) >= krem);
                            RU[t][r] = pad ? T(0) : M::rcp(U[t][r] * T(K));
                            RV[t][r] = pad ? T(0) : T(K) * M::rcp(V[t][r]);
                        }
                    abs_at = ii;
                    flags |= FLAG_ABSORBED;
                } else {
                    done = true; iters = ii + 1; flags |= FLAG_NEEDS_TRACK;
                }
            }
        }
    }
    if (!done) iters = p.max_iter;

    // value <Gamma, M> = u^T (G o M) v   (ot.sinkhorn2 returns sum(M * Gamma))
    panel_product<T, RT>(img_gm, V, ACC, K, lane);
    T val = T(0);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < NREG; ++r) val += U[t][r] * ACC[t][r];
    val = group_sum<T>(val);
    if constexpr (TRACK) {
        if (abs_at >= 0 && abs_at == iters - 1) { val = val / kk; flags |= FLAG_ABSORB_LAST; }
    }
    if (val != val) flags |= FLAG_NAN;
    if constexpr (sizeof(T) == 8) flags |= FLAG_F64;

    if (live && grp == 0) {
        if (!TRACK && (flags & FLAG_NEEDS_TRACK)) {
            const int slot = atomicAdd(p.track_count, 1);
            p.track_list[slot] = q;
        } else {
            p.emd[q] = double(val);
            if (p.iters) p.iters[q] = iters;
            if (p.err) p.err[q] = double(errv);
            if (p.flags) p.flags[q] = flags;
        }
    }
}

// One-block setup: Gibbs kernel images in MFMA operand order + P converted to T.
template <typename T>
__global__ void sinkhorn_setup_kernel(const double *__restrict__ Msrc, int K, int RT, double reg,
                                      T *__restrict__ img, const double *__restrict__ Psrc, T *__restrict__ Pdst,
                                      long n_p) {
    using M = Mfma<T>;
    const int KP = RT * M::TILE;
    const int nimg = KP * KP;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    for (int idx = tid; idx < nimg; idx += nthr) {
        const int lane = idx % WAVE;
        int rest = idx / WAVE;
        const int t = rest % RT; rest /= RT;
        const int r = rest % M::NREG;
        const int tp = rest / M::NREG;
        const int orow = t * M::TILE + lane % M::TILE;                  // output row of the product
        const int k = tp * M::TILE + M::row_of(r, lane / M::TILE);      // contraction index
        double gt = 0.0, g = 0.0, gm = 0.0;
        if (orow < K && k < K) {
            const double m_ko = Msrc[(size_t)k * K + orow], m_ok = Msrc[(size_t)orow * K + k];
            gt = exp(-m_ko / reg);        // (G^T)[orow][k] = G[k][orow]
            g = exp(-m_ok / reg);         // G[orow][k]
            gm = g * m_ok;                // (G o M)[orow][k]
        }
        img[idx] = T(gt);
        img[nimg + idx] = T(g);
        img[2 * nimg + idx] = T(gm);
    }
    if (Pdst)
        for (long idx = tid; idx < n_p; idx += nthr) Pdst[idx] = T(Psrc[idx]);
}

}  // namespace pilot
