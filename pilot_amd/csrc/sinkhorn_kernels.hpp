// Batched Sinkhorn pair-grid kernels for gfx950 (MI355X).  Device code only; the C ABI is in
// pilot_ot.hip.  Replaces the per-pair POT loop of pilotpy/tools/Trajectory.py:512-515.
//
// Mapping (see DESIGN.md "Kernel K2"):
//   * one wavefront iterates TILE ordered pairs at once (TILE = 32 in f32, 16 in f64): the scalings
//     u, v of its pairs form K x TILE panels and one Sinkhorn update is two panel products  G^T U  and
//     G V  with the SHARED K x K Gibbs kernel G = exp(-M/reg);
//   * G is the stationary MFMA "A" operand, pre-arranged once per launch in LDS in exactly the lane
//     order the instruction wants (conflict-free, one ds_read per MFMA) -- and, in the fp16-split configuration with a
//     symmetric cost and K <= 64, held in 64 VGPRs per lane instead, so that the update loop reads no LDS at all;
//   * the result tile of v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 has its column (= pair) on
//     the lane and its rows (= cell types) in the accumulator registers, so it is fed back as the "B"
//     operand of the next product with NO lane movement and NO LDS round trip: k-step (t', r) consumes
//     accumulator register r of row-tile t'.  Cell types are assigned to accumulator slots so that the
//     first ceil(K/2) (f32) / ceil(K/4) (f64) k-steps hold all of them (Cfg*::lidx) and the rest
//     are skipped; the LDS image stores G permuted to match;
//   * element-wise work (v = b / G^T u, u = a / G v, marginal error, tau tracking) happens in that same
//     register layout; per-pair reductions over cell types are in-register sums plus one or two
//     cross-lane xor-shuffles (the lane groups holding the same column);
//   * pairs converge after very different numbers of updates (41 ... 1000), so waves are PERSISTENT and
//     every column is a slot: the moment a pair stops (POT's rule: error checked when ii % period == 0,
//     cap at num_iter_max) its scalings are parked in a wave-private LDS ring and the slot is refilled
//     with the next pair of the wave's queue -- columns of an MFMA are independent, so a pair's
//     arithmetic does not depend on its slot mates;
//   * whenever the ring holds a tile's worth of finished pairs the wave turns them into costs
//     <Gamma, M> = u^T (G o M) v with ONE more panel product (1/16 of a product per pair) and writes the
//     scalars: no scratch in HBM, no second kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

constexpr int WAVE = 64;
constexpr int WAVES_PER_WG = 4;

// flag bits, identical to include/pilot_ot.h
constexpr int FLAG_CONVERGED = 1, FLAG_NAN = 2, FLAG_ABSORB_LAST = 4, FLAG_ABSORBED = 8, FLAG_F64 = 16;

// ---- MFMA configurations: element type + instruction shape -----------------------------------------------
// TILE  rows per accumulator tile == pairs (columns) per wave;  NREG accumulator registers per tile;
// NGRP  lane groups (64 / TILE) == k per MFMA.  lidx(t, r, g): cell type held by accumulator register r of
// row-tile t in lane group g; cell types are dealt to slots in (tile, register, group) order, so k-step
// (t, r) covers NGRP consecutive cell types and the first ceil(K / NGRP) k-steps hold all of them.
// lidx_of_row(t, p): cell type of HARDWARE row p of row-tile t (the A operand is addressed by hardware row).

struct CfgF32x32 {   // v_mfma_f32_32x32x2_f32: hardware row of register r in group g = (r&3) + 8*(r>>2) + 4*g
    using T = float;
    static constexpr bool SPLIT = false, HALF = false;
    static constexpr int NP = 0;
    static constexpr int TILE = 32, NREG = 16, NGRP = 2, VEC = 4;
    using acc_t = float __attribute__((ext_vector_type(16)));
    using vec4_t = float __attribute__((ext_vector_type(4)));
    __host__ __device__ static constexpr int lidx(int t, int r, int g) { return 2 * (16 * t + r) + g; }
    __host__ __device__ static constexpr int lidx_of_row(int t, int p) {
        return lidx(t, (p & 3) + 4 * (p >> 3), (p >> 2) & 1);
    }
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float eps() { return 1.1920929e-07f; }
};

struct CfgF32x16 {   // v_mfma_f32_16x16x4_f32: hardware row of register r in group g = 4*g + r
    using T = float;
    static constexpr bool SPLIT = false, HALF = false;
    static constexpr int NP = 0;
    static constexpr int TILE = 16, NREG = 4, NGRP = 4, VEC = 4;
    using acc_t = float __attribute__((ext_vector_type(4)));
    using vec4_t = float __attribute__((ext_vector_type(4)));
    __host__ __device__ static constexpr int lidx(int t, int r, int g) { return 4 * (4 * t + r) + g; }
    __host__ __device__ static constexpr int lidx_of_row(int t, int p) { return lidx(t, p & 3, p >> 2); }
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float eps() { return 1.1920929e-07f; }
};

// f32 values with every product on v_mfma_f32_16x16x32_bf16 through exact 3-way bf16 operand splits (see
// panel_product_split): same accumulator layout and slot dealing as CfgF32x16, so all element-wise code is shared.
struct CfgS32x16 {
    using T = float;
    static constexpr bool SPLIT = true, HALF = false;
    static constexpr int NP = 3;        // operand pieces
    static constexpr int TILE = 16, NREG = 4, NGRP = 4, VEC = 4;
    using acc_t = float __attribute__((ext_vector_type(4)));
    using vec4_t = float __attribute__((ext_vector_type(4)));
    __host__ __device__ static constexpr int lidx(int t, int r, int g) { return 4 * (4 * t + r) + g; }
    __host__ __device__ static constexpr int lidx_of_row(int t, int p) { return lidx(t, p & 3, p >> 2); }
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {     // (the f32 MFMA: only the selector of tail_rows, unused here)
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float eps() { return 1.1920929e-07f; }
};

// f32 values with every product on v_mfma_f32_16x16x32_f16 through 2-way fp16 operand splits (11 + 11 significant bits,
// 3 piece products instead of 6, 2 instead of 5.5 vector instructions per panel element): the fast path while
// max(M)/reg <= H_MAX_COST_OVER_REG.  fp16 has 5 exponent bits, so the kernel works in a SCALED domain chosen once: the
// Gibbs images hold 2^15 G (entries in [2^-2, 2^15]: both pieces normal), the panels hold 32 u and 32 v (the fast kernel
// hands a pair over at u, v > tau <= 2000, so 32 u < 65504), and a = 2^25 a, b = 2^25 b make the update  v~ = b~ / (G~^T u~)
// come out in the same scaled domain with no extra instruction.  Same accumulator layout as the other 16x16 configurations.
constexpr float H_PANEL_SCALE = 32.f, H_GIBBS_SCALE = 32768.f, H_IN_SCALE = 33554432.f;     // 2^5, 2^15, 2^25 = 2^15 * 2^5 * 2^5
// Range.  Down to exp(-11.78) 2^15 = 2^-2 both pieces of a Gibbs entry are normal fp16 numbers (22 significant bits).  Below,
// the low piece is a subnormal with ABSOLUTE spacing 2^-24, i.e. the entry x is off by <= 2^-24 / x relative: a FIXED
// perturbation of the cost of that arc by reg 2^-24 / x.  At max(M)/reg = 16 the smallest entry is x = 2^15 e^-16 = 2^-8.1,
// so even a plan with ALL its mass on the smallest entries moves by <= reg 2^-15.9 = 1e-6, a tenth of the f32 tolerance;
// measured on c2 / c3 / c4 samples: <= 3.2e-7 against the fp64 oracle up to 16 (2.7e-7 for the bf16 split), 4.8e-7 at 20,
// 1.5e-6 at 22 (tools/f16x2_range_probe.py).  Beyond ~15 the tracking pass takes over the run time and the gain is gone.
constexpr double H_MAX_COST_OVER_REG = 16.0;
constexpr double H_MAX_TAU = 2000.0;
struct CfgH32x16 {
    using T = float;
    static constexpr bool SPLIT = true, HALF = true;
    static constexpr int NP = 2;
    static constexpr int TILE = 16, NREG = 4, NGRP = 4, VEC = 4;
    using acc_t = float __attribute__((ext_vector_type(4)));
    using vec4_t = float __attribute__((ext_vector_type(4)));
    __host__ __device__ static constexpr int lidx(int t, int r, int g) { return 4 * (4 * t + r) + g; }
    __host__ __device__ static constexpr int lidx_of_row(int t, int p) { return lidx(t, p & 3, p >> 2); }
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {     // (unused: selector of tail_rows only)
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float eps() { return 1.1920929e-07f; }
};

struct CfgF64x16 {   // v_mfma_f64_16x16x4_f64 has its own C/D map: hardware row = (lane>>4) + 4*reg (k-step order)
    using T = double;
    static constexpr bool SPLIT = false, HALF = false;
    static constexpr int NP = 0;
    static constexpr int TILE = 16, NREG = 4, NGRP = 4, VEC = 2;
    using acc_t = double __attribute__((ext_vector_type(4)));
    using vec4_t = double __attribute__((ext_vector_type(2)));
    __host__ __device__ static constexpr int lidx(int t, int r, int g) { return 16 * t + 4 * r + g; }
    __host__ __device__ static constexpr int lidx_of_row(int t, int p) { return 16 * t + p; }
    __device__ static inline acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __device__ static inline double rcp(double x) { return 1.0 / x; }
    __device__ static inline double eps() { return 2.220446049250313e-16; }
};

// index of the LDS/global "A image" element read by `lane` for k-step (tp, r) and output row-tile t
template <class C>
__host__ __device__ constexpr int img_index(int RT, int tp, int r, int t, int lane) {
    return (((tp * C::NREG + r) * RT + t) * WAVE) + lane;
}
// finished pairs wait in a wave-private LDS ring for their cost product: per slot the u panel and the v panel (KP values
// each, [tile][group][reg] order) + 4 elements of padding (a lane's 16-byte reads of consecutive slots then fall on
// different banks); ring_meta: per slot the output index, the flags and POT's plan scale (1, or 1/K^2)
// (fp16-split configuration: the panels are parked as packed pieces, [part][k-block][lane group] x 16 bytes per column --
// an odd row-tile count rounds up to whole k-blocks)
template <class C> __host__ __device__ constexpr int ring_panel_elems(int RT) { return C::HALF ? 2 * ((RT + 1) / 2) * C::NGRP * 4 : RT * C::TILE; }
template <class C> __host__ __device__ constexpr int ring_slot_stride(int RT) { return 2 * ring_panel_elems<C>(RT) + 4; }
// (parked flush) 4-byte words a lane parks: its U registers, or its packed U pieces
template <class C> __host__ __device__ constexpr int park_lane_elems(int RT) { return C::HALF ? 2 * ((RT + 1) / 2) * 4 : RT * C::NREG; }
constexpr int RING_MAX = 16;

// Cross-lane exchanges between the lane groups of a column (lanes l, l^16, l^32, l^48) with gfx950's
// v_permlane32_swap / v_permlane16_swap: VALU-rate, no LDS crossbar round trip (ds_bpermute costs ~100+ cycles of
// latency per hop, twice per reduction, in the serial part of every update).
// permlane32_swap(v, v) returns {lower half kept, upper half := v's lower half} and {lower := v's upper half, upper
// kept}; adding the two gives v[l] + v[l^32] in every lane.  permlane16_swap does the same for odd/even 16-lane rows.
__device__ inline float sum_xor32(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float sum_xor16(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline double sum_xor32(double x) {
    union { double d; unsigned int u[2]; } v, a, b;
    v.d = x;
    auto lo = __builtin_amdgcn_permlane32_swap(v.u[0], v.u[0], false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(v.u[1], v.u[1], false, false);
    a.u[0] = lo[0]; a.u[1] = hi[0]; b.u[0] = lo[1]; b.u[1] = hi[1];
    return a.d + b.d;
}
__device__ inline double sum_xor16(double x) {
    union { double d; unsigned int u[2]; } v, a, b;
    v.d = x;
    auto lo = __builtin_amdgcn_permlane16_swap(v.u[0], v.u[0], false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(v.u[1], v.u[1], false, false);
    a.u[0] = lo[0]; a.u[1] = hi[0]; b.u[0] = lo[1]; b.u[1] = hi[1];
    return a.d + b.d;
}
// the lanes where `b` holds (HIP's __ballot(int) materialises the bool as 0 / 1 in a register and compares it again: two
// vector instructions per call; PILOT_BALLOT_INT=1 is that form, for A/B)
#ifndef PILOT_BALLOT_INT
#define PILOT_BALLOT_INT 0
#endif
#ifndef PILOT_BALLOT_B_DEFINED
#define PILOT_BALLOT_B_DEFINED
__device__ inline unsigned long long ballot_b(bool b) {
#if PILOT_BALLOT_INT
    return __ballot(b);
#else
    return __builtin_amdgcn_ballot_w64(b);
#endif
}
#endif
template <class C> __device__ inline typename C::T group_sum(typename C::T x) {
    // sum over the lane groups that hold the same column (lane % TILE); every lane of the column gets the total
    x = sum_xor32(x);
    if constexpr (C::NGRP == 4) x = sum_xor16(x);
    return x;
}
// "does any lane of my column satisfy pred": one ballot + scalar folds, no cross-lane data movement
template <class C> __device__ inline unsigned long long column_any_mask(bool pred) {
    const unsigned long long m = ballot_b(pred);
    unsigned long long f = m | (m >> 32);
    if constexpr (C::NGRP == 4) f |= f >> 16;
    return f & ((1ull << C::TILE) - 1ull);
}
template <typename T> __device__ inline T abs_t(T x) { return x < T(0) ? -x : x; }

// Where the stationary A operand of a product comes from: the lane-ordered image in LDS / global memory
// (one ds_read / global load per MFMA), or -- when the whole image fits in <= 64 registers per lane (small K,
// symmetric cost) -- registers loaded once per wave, which removes every LDS access from the update loop.
template <class C> struct AFromImage {
    const typename C::T *img; int lane;
    __device__ inline typename C::T operator()(int step_tile) const { return img[step_tile * WAVE + lane]; }
};
template <class C, int NA> struct AFromRegs {
    typename C::T a[NA];
    __device__ inline typename C::T operator()(int step_tile) const { return a[step_tile]; }
};

// OUT[t] = sum over k of X_img[out row][k] * IN[k]   for the whole K x TILE panel
// `last_init` seeds the accumulator of the last row-tile (1 in padded slots keeps 0 / OUT finite there).
//
// k-steps of row-tiles tp < RT-1 are always inside K (RT = ceil(K / TILE)); only the last tile has padding.
// The number of k-steps it needs is wave-uniform but only known at run time, so the tail is a jump table
// over straight-line variants (NLAST = CH, 2*CH, ... registers): no branch sits between an operand's
// ds_read and its MFMA, and the scheduler sees one block per variant.
template <class C, int RT, int NLAST, class AOp>
__device__ inline void panel_product_fixed(const AOp &aop, const typename C::acc_t (&IN)[RT],
                                           typename C::acc_t (&OUT)[RT],
                                           const typename C::acc_t &last_init) {
    using M = C;
    using T = typename C::T;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) OUT[t][r] = (t == RT - 1) ? last_init[r] : T(0);
    // k-step outer, output tile inner: RT independent accumulator chains are interleaved, so a wave alone on
    // its SIMD never waits on the dependent-accumulator latency
#pragma unroll
    for (int tp = 0; tp < RT; ++tp)
#pragma unroll
        for (int r = 0; r < (tp < RT - 1 ? M::NREG : NLAST); ++r)
#pragma unroll
            for (int t = 0; t < RT; ++t)
                OUT[t] = M::mfma(aop((tp * M::NREG + r) * RT + t), IN[tp][r], OUT[t]);
}

template <class C, int RT, class AOp>
__device__ inline void panel_product(const AOp &aop, const typename C::acc_t (&IN)[RT],
                                     typename C::acc_t (&OUT)[RT], int K,
                                     const typename C::acc_t &last_init) {
    using M = C;
    // registers of the last row-tile that hold a cell type (group 0 holds the smallest index of a register)
    const int n_last = (K - M::lidx(RT - 1, 0, 0) + M::NGRP - 1) / M::NGRP;
    if constexpr (M::NREG == 16) {
        switch ((n_last + 1) / 2) {
        case 1: panel_product_fixed<C, RT, 2>(aop, IN, OUT, last_init); break;
        case 2: panel_product_fixed<C, RT, 4>(aop, IN, OUT, last_init); break;
        case 3: panel_product_fixed<C, RT, 6>(aop, IN, OUT, last_init); break;
        case 4: panel_product_fixed<C, RT, 8>(aop, IN, OUT, last_init); break;
        case 5: panel_product_fixed<C, RT, 10>(aop, IN, OUT, last_init); break;
        case 6: panel_product_fixed<C, RT, 12>(aop, IN, OUT, last_init); break;
        case 7: panel_product_fixed<C, RT, 14>(aop, IN, OUT, last_init); break;
        default: panel_product_fixed<C, RT, 16>(aop, IN, OUT, last_init); break;
        }
    } else {
        switch (n_last) {
        case 1: panel_product_fixed<C, RT, 1>(aop, IN, OUT, last_init); break;
        case 2: panel_product_fixed<C, RT, 2>(aop, IN, OUT, last_init); break;
        case 3: panel_product_fixed<C, RT, 3>(aop, IN, OUT, last_init); break;
        default: panel_product_fixed<C, RT, 4>(aop, IN, OUT, last_init); break;
        }
    }
}

// ---- VALU tail rows (16x16x4 configurations) --------------------------------------------------------------------------------
// When K mod 16 is 1..4 the last row-tile holds at most four cell types (accumulator register 0 of the four lane groups)
// and a product spends ceil(K/4) MFMAs (32 matrix-pipe cycles each) on 16 rows of which 12+ are padding.  With
// TV = 1 (<= 2 live rows) or 2 (<= 4) those rows are computed on the VALU instead: every lane multiplies the
// ceil(K/4) panel values it already holds (its lane group's share of the contraction index) with the matching
// weights of TWO rows at once (f32: v_pk_fma_f32, ~8 pipe cycles; f64: two v_fma_f64), and the four lane-group partials of a column are
// added with the permlane swaps.  13 pk-FMAs + one reduction replace 13 MFMAs at K = 50.  The element-wise loops skip
// the three all-padding registers of that tile as well.  Every kernel variant (stream, tracking) uses
// this one function, so a pair's bits still do not depend on which kernel solves it.
template <typename T> using pair_of = T __attribute__((ext_vector_type(2)));
template <int RT> __host__ __device__ constexpr int tail_steps() { return (RT - 1) * 4 + 1; }   // k-steps when only register 0 of the last tile is live
// global tail image (f32, written by sinkhorn_setup_kernel behind the first-product table): [form 0: G^T-form, 1: G-form]
// [chain 0..1][k-step][lane] pairs (X[row 2c][k], X[row 2c+1][k]), row h = cell type 16 (RT-1) + h, k = lidx(step, lane / 16)
template <int RT> __host__ __device__ constexpr int tail_form_stride() { return 2 * tail_steps<RT>() * WAVE; }  // in pairs

template <typename T, int RT> struct TailFromImage {
    const pair_of<T> *w; int lane;
    __device__ inline pair_of<T> operator()(int c, int st) const { return w[(c * tail_steps<RT>() + st) * WAVE + lane]; }
};
template <typename T, int RT, int TV> struct TailFromRegs {
    pair_of<T> a[(TV > 0 ? TV : 1) * tail_steps<RT>()];
    __device__ inline pair_of<T> operator()(int c, int st) const { return a[c * tail_steps<RT>() + st]; }
};
struct TailNone {};

template <class C, int RT, int TV, class WOp>
__device__ inline typename C::acc_t tail_rows(const WOp &w, const typename C::acc_t (&IN)[RT], const typename C::acc_t &last_init, int grp) {
    static_assert(C::NGRP == 4 && C::NREG == 4 && C::lidx(1, 0, 1) == 17, "VALU tail rows: 16x16x4 layouts with lidx = 16t + 4r + g");
    using T = typename C::T;
    using f2_t = pair_of<T>;
    f2_t acc[TV];
#pragma unroll
    for (int c = 0; c < TV; ++c) acc[c] = f2_t{T(0), T(0)};
#pragma unroll
    for (int tp = 0; tp < RT; ++tp)
#pragma unroll
        for (int r = 0; r < (tp < RT - 1 ? 4 : 1); ++r) {
            // (forcing v_pk_fma_f32 with inline asm where the compiler splits it into two v_fma_f32 was measured: no gain)
            const f2_t x = {IN[tp][r], IN[tp][r]};
#pragma unroll
            for (int c = 0; c < TV; ++c) acc[c] = __builtin_elementwise_fma(w(c, tp * 4 + r), x, acc[c]);
        }
    if constexpr (sizeof(T) == 8) {     // f64: the permlane reduction is cheaper than two more 64-cycle MFMAs (measured)
#pragma unroll
    for (int c = 0; c < TV; ++c) {
        acc[c][0] = sum_xor16(sum_xor32(acc[c][0]));
        acc[c][1] = sum_xor16(sum_xor32(acc[c][1]));
    }
    T val = (grp & 1) ? acc[0][1] : acc[0][0];
    if constexpr (TV == 2) {
        const T hi = (grp & 1) ? acc[1][1] : acc[1][0];
        val = grp < 2 ? val : hi;
    }
    typename C::acc_t out = last_init;
    out[0] = last_init[0] + val;    // 1 in padded slots (their weights are 0), 0 in live ones
    return out;
    } else {
    // the four lane-group partials of a column are the four k-slots of ONE more MFMA whose A operand is 1 in the hardware
    // row of that cell type and 0 elsewhere: the matrix pipe does the cross-lane sum and drops it into the right
    // accumulator slot, on top of the padding seed (2 MFMAs = 64 pipe cycles instead of ~14 VALU/permlane instructions)
    typename C::acc_t out = last_init;
    const int lane = threadIdx.x % WAVE;
#pragma unroll
    for (int c = 0; c < TV; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const T sel = C::lidx_of_row(0, lane % C::TILE) == 2 * c + h ? T(1) : T(0);
            out = C::mfma(sel, acc[c][h], out);
        }
    return out;
    }
}

// panel product with the last row-tile on the VALU (TV > 0): MFMA chains for row-tiles 0 .. RT-2 only
template <class C, int RT, int TV, class AOp, class WOp>
__device__ inline void panel_product_tail(const AOp &aop, const WOp &wop, const typename C::acc_t (&IN)[RT],
                                          typename C::acc_t (&OUT)[RT], const typename C::acc_t &last_init, int grp) {
    using M = C;
    using T = typename C::T;
#pragma unroll
    for (int t = 0; t < RT - 1; ++t)
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) OUT[t][r] = T(0);
#pragma unroll
    for (int tp = 0; tp < RT; ++tp)
#pragma unroll
        for (int r = 0; r < (tp < RT - 1 ? M::NREG : 1); ++r)
#pragma unroll
            for (int t = 0; t < RT - 1; ++t)
                OUT[t] = M::mfma(aop((tp * M::NREG + r) * RT + t), IN[tp][r], OUT[t]);
    OUT[RT - 1] = tail_rows<C, RT, TV>(wop, IN, last_init, grp);
}

template <class C> __device__ inline void store_regs(typename C::T *dst, const typename C::acc_t &x) {
    using V = typename C::vec4_t;
#pragma unroll
    for (int c = 0; c < C::NREG / C::VEC; ++c) {
        V v;
#pragma unroll
        for (int e = 0; e < C::VEC; ++e) v[e] = x[c * C::VEC + e];
        *reinterpret_cast<V *>(dst + c * C::VEC) = v;
    }
}
template <class C> __device__ inline void load_regs(const typename C::T *src, typename C::acc_t &x) {
    using V = typename C::vec4_t;
#pragma unroll
    for (int c = 0; c < C::NREG / C::VEC; ++c) {
        const V v = *reinterpret_cast<const V *>(src + c * C::VEC);
#pragma unroll
        for (int e = 0; e < C::VEC; ++e) x[c * C::VEC + e] = v[e];
    }
}

// ---- bf16-split products (configuration CfgS32x16) ---------------------------------------------------------------------------
// gfx950's f32-input MFMA runs at the f32 VECTOR rate and shares the SIMD's f32 lanes with every other VALU instruction
// (measured: SQ_VALU_MFMA_COEXEC_CYCLES = 0), while v_mfma_f32_16x16x32_bf16 does 8x the contraction depth in half the
// cycles on the matrix pipe proper and lets VALU work issue beside it.  A product of f32 operands is therefore taken
// apart into bf16 pieces: x = x1 + x2 + x3 exactly (8 + 8 + 8 significant bits, round-to-nearest residuals), and
//     sum_k a_k b_k  ~=  sum_k (a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1)_k
// with every piece product exact and the sums accumulated in f32 by the MFMA (terms of order 2^-24 |a b| dropped: the
// rounding level of an f32 FMA chain).  The stationary operand G is split once per call by the prep kernel; the scaling
// panel (the B operand) is split in registers before each product, 5.5 VALU instructions per element.
// Layout: the 16x16 f32 result tile of 16x16x32 has the same lane/register map as 16x16x4's, and a lane's 8 bf16 B values
// of k-block kb are exactly its 2 x 4 accumulator registers of row-tiles 2 kb and 2 kb + 1 -- so the result still feeds the
// next product with no lane movement: k-slot (group g, element e) of block kb <-> cell type lidx(2 kb + e / 4, e % 4, g).
constexpr float SPLIT_SAFE_MIN = 7.70371978e-34f;     // 2^-110: below it the low pieces of a 3-way bf16 split are subnormal
constexpr float BAND1_DOWN = 2.93873588e-39f;          // 2^-128: scale of the second exponent band (see band1_offset)
constexpr double BAND1_UP_LN = 88.722839111672999;     // 128 ln 2
using bf16x8_t = __bf16 __attribute__((ext_vector_type(8)));
using bf16x2_t = __bf16 __attribute__((ext_vector_type(2)));
using f16x8_t = _Float16 __attribute__((ext_vector_type(8)));
using f16x2_t = _Float16 __attribute__((ext_vector_type(2)));
using u32x4_t = unsigned int __attribute__((ext_vector_type(4)));
using f32x2_t = float __attribute__((ext_vector_type(2)));

__device__ inline unsigned int cvt_pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32, round to nearest even
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ inline unsigned int cvt_pk_f16(float lo, float hi) {       // v_cvt_pk_f16_f32, round to nearest even
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}
// x - (the low / high fp16 of the packed register h) in ONE instruction each: v_fma_mix_f32 reads an fp16 operand straight
// out of a packed register (the compiler's own choice is two v_cvt_f32_f16 and a double-rate v_pk_add_f32)
__device__ inline float resid_f16_lo(float x, unsigned int h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
__device__ inline float resid_f16_hi(float x, unsigned int h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
// fp16-split configuration, round 3: the scalings LIVE as packed fp16 pieces (the B operands of the products as they are), the
// f32 quotient x = r b (r = 1 / acc) is a temporary: hi = fp16(x), lo = fp16(x - hi) -- 2 multiplies, v_cvt_pk_f16_f32, two
// v_fma_mix_f32 (residual in one instruction), v_cvt_pk_f16_f32 per element pair.  (One fused multiply-add per half,
// v_fma_mixlo_f16 / v_fma_mixhi_f16, would be 4 instructions instead of 6, but those two issue at 7.4 cycles against 4.3 for
// the others -- tools/ubench/valu_rates.hip -- and measured slower: 0.745 vs 0.702 ms at c3.)
__device__ inline void quot_pieces(float x0, float x1, unsigned int &hi, unsigned int &lo) {
    hi = cvt_pk_f16(x0, x1);
    lo = cvt_pk_f16(resid_f16_lo(x0, hi), resid_f16_hi(x1, hi));
}
// hi + lo of the low / high halves of two packed registers as f32 (one v_fma_mix_f32 with two fp16 sources)
__device__ inline float pieces_sum_lo(unsigned int h, unsigned int l) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ inline float pieces_sum_hi(unsigned int h, unsigned int l) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
// packed three-way maximum of fp16 pairs (gfx950: v_pk_maximum3_f16; a NaN operand makes the result NaN)
__device__ inline unsigned int pk_max3_f16(unsigned int a, unsigned int b, unsigned int c) {
    unsigned int r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (x0, x1) -> NP packed pairs of pieces, largest first.  bf16 (NP = 3): hi + mid + lo == x exactly (each residual is exactly
// representable).  fp16 (NP = 2): hi + lo == x to 2^-22 |x| while the low piece is a normal number (scaled domain of CfgH32x16).
template <int NP> struct Pieces { unsigned int p[NP]; };
template <class C> __device__ inline Pieces<C::NP> split_pair(float x0, float x1) {
    Pieces<C::NP> o;
    if constexpr (C::HALF) {
        o.p[0] = cvt_pk_f16(x0, x1);
        o.p[1] = cvt_pk_f16(resid_f16_lo(x0, o.p[0]), resid_f16_hi(x1, o.p[0]));
    } else {
        o.p[0] = cvt_pk_bf16(x0, x1);
        float r0 = x0 - __uint_as_float(o.p[0] << 16), r1 = x1 - __uint_as_float(o.p[0] & 0xffff0000u);
        o.p[1] = cvt_pk_bf16(r0, r1);
        r0 -= __uint_as_float(o.p[1] << 16);
        r1 -= __uint_as_float(o.p[1] & 0xffff0000u);
        o.p[2] = cvt_pk_bf16(r0, r1);
    }
    return o;
}
__host__ __device__ constexpr int split_kblocks(int RT) { return (RT + 1) / 2; }
// piece products of one term block, smallest first: (A piece, B piece).  3 pieces: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1;
// 2 pieces: a2 b1, a1 b2, a1 b1
template <int NP> __host__ __device__ constexpr int n_terms() { return NP == 3 ? 6 : 3; }
template <int NP> __host__ __device__ constexpr int term_a(int i) {
    if (NP == 3) return i == 0 ? 2 : (i == 2 || i == 3 ? 1 : 0);
    return i == 0 ? 1 : 0;
}
template <int NP> __host__ __device__ constexpr int term_b(int i) {
    if (NP == 3) return i == 1 ? 2 : (i == 2 || i == 4 ? 1 : 0);
    return i == 1 ? 1 : 0;
}
// TAIL16: the last k-block of an ODD row-tile count holds one row-tile, i.e. 16 k-slots: the lane's first four of eight (k-slot e of
// group g <-> cell type lidx(2 kb + e / 4, e % 4, g), so e < 4 is row-tile 2 kb and the 16x16x16 instruction's own slot 4 g + e) -- the
// same operand registers, their low 8 bytes, on v_mfma_f32_16x16x16_{f16,bf16}: half the matrix-pipe time of a k-block that was half
// padding (K = 100: 7 row-tiles, 4 k-blocks -> 3.5: c4 25.06 -> 23.2 ms, K = 100 at N = 600 1.69 -> 1.60 ms; tools/ubench/mfma_tail16.hip
// checks the instruction pair against each other).
// HAZARD (found the hard way, profiles/r06/ab_experiments.md section 4): a 16x16x16 MFMA that reads as SrcC the result of a 16x16x32 MFMA
// issued a few instructions earlier gets a stale accumulator -- the hardware forwards accumulators only between MFMAs of one shape,
// and hipcc (ROCm 7.2) inserts no wait states for the mixed pair (it does for a bare back-to-back pair, which is why a two-instruction
// test passes).  So: either the tail MFMAs of a tile are issued many MFMAs after its full ones (tails last: the register-image order 2,
// PILOT_TAIL16_ORDER 1), or tail16_gap() stands between them.
#ifndef PILOT_TAIL16
#define PILOT_TAIL16 7      // bit 0: register image, bit 1: LDS image, bit 2: cost flush / band-1 products
#endif
template <int RT, int BIT = 7> __host__ __device__ constexpr bool tail16_kb(int kb) { return (PILOT_TAIL16 & BIT) && (RT & 1) && kb == (RT + 1) / 2 - 1; }
using u32x2_t = unsigned int __attribute__((ext_vector_type(2)));
using f16x4_t = _Float16 __attribute__((ext_vector_type(4)));
using s16x4_t = short __attribute__((ext_vector_type(4)));
#ifndef PILOT_TAIL16_GAP
#define PILOT_TAIL16_GAP 12     // (measured on c4-shaped and K = 33 .. 112 grids: 4 wait states still give wrong sums, 8 / 12 / 16 are right)
#endif
#ifndef PILOT_TAIL16_ORDER
#define PILOT_TAIL16_ORDER 1    // LDS-image products: 1 = tails last (no gap needed), 0 = tile after tile with the gap (c4 23.21 vs 23.13 ms, K = 80 0.968 vs 0.997)
#endif
// wait states between a 16x16x32 MFMA and a 16x16x16 MFMA that reads its result as SrcC (ROCm 7.2 inserts none, the
// hardware forwards an accumulator only between MFMAs of the same shape)
__device__ inline void tail16_gap() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(PILOT_TAIL16_GAP - 1) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <class C, bool TAIL16 = false> __device__ inline typename C::acc_t mfma_pieces(u32x4_t a, u32x4_t b, typename C::acc_t c) {
    if constexpr (TAIL16) {
        const u32x2_t a2 = {a[0], a[1]}, b2 = {b[0], b[1]};
        if constexpr (C::HALF) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a2), __builtin_bit_cast(f16x4_t, b2), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a2), __builtin_bit_cast(s16x4_t, b2), c, 0, 0, 0);
    } else {
    if constexpr (C::HALF) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
}

// the scaling panel as packed B operands: p[part][k-block] = the lane's 8 k-slots (accumulator registers of row-tiles 2 kb, 2 kb + 1)
template <int RT, int NP> struct SplitPanel { u32x4_t p[NP][split_kblocks(RT)]; };
// LIVE1: only accumulator register 0 of the last row-tile holds cell types (K mod 16 in 1..4); the others are zero
template <class C, int RT, bool LIVE1 = false>
__device__ inline void split_panel(const typename C::acc_t (&IN)[RT], SplitPanel<RT, C::NP> &B) {
    constexpr int KB = split_kblocks(RT);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int h = 0; h < 4; ++h) {       // element pairs (0,1), (2,3) of row-tile 2 kb, then of row-tile 2 kb + 1
            const int t = 2 * kb + h / 2;
            Pieces<C::NP> sp;
#pragma unroll
            for (int part = 0; part < C::NP; ++part) sp.p[part] = 0u;
            if (t < RT && !(LIVE1 && t == RT - 1 && (h & 1)))
                sp = split_pair<C>(IN[t < RT ? t : 0][2 * (h & 1)], (LIVE1 && t == RT - 1) ? 0.f : IN[t < RT ? t : 0][2 * (h & 1) + 1]);
#pragma unroll
            for (int part = 0; part < C::NP; ++part) B.p[part][kb][h] = sp.p[part];
        }
}
// one 16-row output tile of X * panel; `form` = the [part][k-block][out tile][lane] x 16-byte image of X (LDS or global)
// TP: pieces that take part (default: all).  TP = 2 of a 3-piece bf16 image = the leading 16 bits of both operands, three
// piece products -- what the band-1 products use (below).
template <class C, int RT, int TP = C::NP>
__device__ inline typename C::acc_t split_tile_product(const typename C::T *form, int lane, int t, const SplitPanel<RT, C::NP> &B,
                                                       typename C::acc_t acc) {
    constexpr int KB = split_kblocks(RT);
    const u32x4_t *img = reinterpret_cast<const u32x4_t *>(form);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        u32x4_t a[TP];
#pragma unroll
        for (int part = 0; part < TP; ++part) a[part] = img[((part * KB + kb) * RT + t) * WAVE + lane];
        if (tail16_kb<RT, 4>(kb)) {         // (a compile-time condition after unrolling)
            if (kb > 0) tail16_gap();
#pragma unroll
            for (int i = 0; i < n_terms<TP>(); ++i) acc = mfma_pieces<C, true>(a[term_a<TP>(i)], B.p[term_b<TP>(i)][kb], acc);
        } else {
#pragma unroll
        for (int i = 0; i < n_terms<TP>(); ++i) acc = mfma_pieces<C>(a[term_a<TP>(i)], B.p[term_b<TP>(i)][kb], acc);
        }
    }
    return acc;
}
// OUT = X_form * IN for the whole panel.  The operand pieces of a later step are requested from LDS before the MFMAs of step
// (t, kb) are issued (lds_ahead), so a wave does not sit on an LDS round trip between MFMA groups.
// form1 (nullable, wave-uniform): the band-1 image of the same operand; OUT = X0 * IN + 2^-128 (X1 * IN).
// Band 1 holds the Gibbs entries below 2^-110: what they contribute to a product is a small part of it (the plans of the
// 600 x 50 benchmark at reg 0.01 put 1e-8 .. 1e-2 of their mass there), so the band-1 partial product is taken from the
// leading TWO bf16 pieces of both operands (16 bits, three piece products instead of six): the dropped
// piece of the stationary operand is a fixed relative perturbation <= 2^-16 of those entries -- a cost change of
// reg * 2^-16 on entries that carry <= 1e-2 of the mass -- and the panel's is rounding noise of the same size on that part
// (c3 at reg 0.01: 37.8 -> 32.0 ms per matrix, max distance to the fp64 oracle on 12 000 pairs 1.5e-7 -> 4.4e-7).
// How many steps ahead of its MFMAs a step's operand pieces are requested from LDS (PILOT_LDS_AHEAD = n forces it for A/B builds).
// One step ahead (rounds 3 - 5) leaves a wave waiting on the LDS round trip at every other step; further ahead costs 8 registers
// per step.  fp16-split configuration, same box (tools/ahead_variants.sh, profiles/r06/ahead_variants.txt): 1 / 2 / 3 steps ahead
// K = 72 0.963 / 0.916 / 0.939 ms, K = 80 0.936 / 0.895 / 0.901, K = 96 0.958 / 0.951 / 0.934, K = 100 1.477 / 1.441 / 1.460,
// c4 21.75 / 21.67 / 21.54 ms per launch; the three-piece bf16 images (12 registers per step) stay at one.
template <class C, int RT> __host__ __device__ constexpr int lds_ahead() {
#ifdef PILOT_LDS_AHEAD
    return PILOT_LDS_AHEAD;
#else
    return !C::HALF ? 1 : (RT <= 5 ? 2 : 3);
#endif
}
template <class C, int RT>
__device__ inline void panel_product_pieces(const typename C::T *form, const typename C::T *form1, int lane,
                                            const SplitPanel<RT, C::NP> &B, typename C::acc_t (&OUT)[RT],
                                            const typename C::acc_t &last_init) {
    constexpr int KB = split_kblocks(RT), NP = C::NP;
    const u32x4_t *img = reinterpret_cast<const u32x4_t *>(form) + lane;
    // the steps (tile, k-block) in issue order.  TAILS_LAST (odd row-tile counts with the 16-wide tail k-block, >= 3 row-tiles): the full
    // k-blocks of every tile first, then the tails tile after tile -- a tile's 16x16x16 MFMAs then read an accumulator that was
    // written many MFMAs earlier (the hardware forwards an accumulator only between MFMAs of one shape, and ROCm 7.2 puts no wait
    // states between a 16x16x32 and a 16x16x16 that reads its result: see tail16_gap)
    constexpr bool TAILS_LAST = PILOT_TAIL16_ORDER == 1 && tail16_kb<RT, 2>(KB - 1) && KB > 1;
    constexpr int STEPS = RT * KB, FULL = RT * (KB - 1);
    auto step_t = [](int s) { return TAILS_LAST ? (s < FULL ? s / (KB - 1) : s - FULL) : s / KB; };
    auto step_kb = [](int s) { return TAILS_LAST ? (s < FULL ? s % (KB - 1) : KB - 1) : s % KB; };
    // the operand pieces of step s + AHEAD are requested from LDS before the MFMAs of step s are issued (a rotating set of AHEAD + 1
    // register groups; the indices are compile-time after unrolling)
    constexpr int AHEAD = lds_ahead<C, RT>();
    u32x4_t a[AHEAD + 1][NP];
#pragma unroll
    for (int d = 0; d < AHEAD; ++d)
        if (d < STEPS) {
#pragma unroll
            for (int part = 0; part < NP; ++part) a[d][part] = img[((part * KB + step_kb(d)) * RT + step_t(d)) * WAVE];
        }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[t][r] = (t == RT - 1) ? last_init[r] : 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int t = step_t(s), kb = step_kb(s);
        if (s + AHEAD < STEPS) {
            const int nt = step_t(s + AHEAD), nkb = step_kb(s + AHEAD);
#pragma unroll
            for (int part = 0; part < NP; ++part) a[(s + AHEAD) % (AHEAD + 1)][part] = img[((part * KB + nkb) * RT + nt) * WAVE];
        }
        __builtin_amdgcn_sched_barrier(0x6);        // (VALU / SALU may move across; the LDS reads stay ahead of the MFMAs)
        const u32x4_t (&ac)[NP] = a[s % (AHEAD + 1)];
        if (tail16_kb<RT, 2>(kb)) {
            if (kb > 0 && !TAILS_LAST) tail16_gap();
#pragma unroll
            for (int i = 0; i < n_terms<NP>(); ++i) OUT[t] = mfma_pieces<C, true>(ac[term_a<NP>(i)], B.p[term_b<NP>(i)][kb], OUT[t]);
        } else {
#pragma unroll
            for (int i = 0; i < n_terms<NP>(); ++i) OUT[t] = mfma_pieces<C>(ac[term_a<NP>(i)], B.p[term_b<NP>(i)][kb], OUT[t]);
        }
        __builtin_amdgcn_sched_barrier(0x6);
    }
    if (form1) {        // wave-uniform; a compile-time nullptr in the single-band kernels
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            typename C::acc_t zero;
#pragma unroll
            for (int r = 0; r < 4; ++r) zero[r] = 0.f;
            const typename C::acc_t lo = split_tile_product<C, RT, 2>(form1, lane, t, B, zero);
#pragma unroll
            for (int r = 0; r < 4; ++r) OUT[t][r] = fmaf(lo[r], BAND1_DOWN, OUT[t][r]);
        }
    }
}
// The stationary operand in REGISTERS (fp16-split configuration, symmetric cost, <= 4 row-tiles: 2 parts x k-blocks x tiles x
// 16 bytes <= 64 VGPRs per lane, and G^T = G serves both products): no LDS read in the update loop.  From LDS every MFMA
// pair waits for a 1 KB operand read, and with all four SIMDs of a CU inside products those reads alone take two thirds of
// the LDS bandwidth (one ds_read_b128 = 4 LDS cycles per 16-cycle MFMA and SIMD, MI355X_MICROARCH.md "LDS").
template <int RT, int NP> struct SplitImage { u32x4_t a[NP][split_kblocks(RT)][RT]; };
#ifndef PILOT_AREG_ORDER
#define PILOT_AREG_ORDER 2      // (round 4: 1 -> 2, c3 kernel 0.634 -> 0.617 ms on one box; orders 0, 3, 4: 0.634, 0.626, 0.619)
#endif
#ifndef PILOT_E2_EARLY
#define PILOT_E2_EARLY 0      // (tried: neutral, c3 kernel 0.611 - 0.614 ms either way)
#endif
#ifndef PILOT_DEFER_HANDOVER
#define PILOT_DEFER_HANDOVER 1
#endif
#ifndef PILOT_E2_PACKED
#define PILOT_E2_PACKED 1     // (round 6: the lane's squared error two slots per v_pk_fma_f32; instructions -12 per error test, time unchanged)
#endif
template <class C, int RT>
__device__ inline void panel_product_pieces_regs(const SplitImage<RT, C::NP> &A, const SplitPanel<RT, C::NP> &B,
                                                 typename C::acc_t (&OUT)[RT], const typename C::acc_t &last_init) {
    constexpr int KB = split_kblocks(RT), NP = C::NP;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[t][r] = (t == RT - 1) ? last_init[r] : 0.f;
#if PILOT_AREG_ORDER == 0
    // one output tile after the other: the element-wise work on tile t can start while the MFMAs of tile t + 1 run
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < n_terms<NP>(); ++i) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][kb][t], B.p[term_b<NP>(i)][kb], OUT[t]);
#elif PILOT_AREG_ORDER == 2
    // chains interleaved up to the last k-block, the last one tile after tile: tile t is complete 3 (RT - 1 - t) MFMAs before
    // the end, so its element-wise work can run under the MFMAs of the later tiles (same per-tile term order: same bits)
#pragma unroll
    for (int kb = 0; kb < KB - 1; ++kb)
#pragma unroll
        for (int i = 0; i < n_terms<NP>(); ++i)
#pragma unroll
            for (int t = 0; t < RT; ++t) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][kb][t], B.p[term_b<NP>(i)][kb], OUT[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < n_terms<NP>(); ++i) OUT[t] = mfma_pieces<C, tail16_kb<RT, 1>(KB - 1)>(A.a[term_a<NP>(i)][KB - 1][t], B.p[term_b<NP>(i)][KB - 1], OUT[t]);
#elif PILOT_AREG_ORDER == 3
    // every k-block tile after tile
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < n_terms<NP>(); ++i) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][kb][t], B.p[term_b<NP>(i)][kb], OUT[t]);
#elif PILOT_AREG_ORDER == 4
    // like 2, the last k-block in tile PAIRS (two chains interleaved)
#pragma unroll
    for (int kb = 0; kb < KB - 1; ++kb)
#pragma unroll
        for (int i = 0; i < n_terms<NP>(); ++i)
#pragma unroll
            for (int t = 0; t < RT; ++t) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][kb][t], B.p[term_b<NP>(i)][kb], OUT[t]);
#pragma unroll
    for (int t0 = 0; t0 < RT; t0 += 2)
#pragma unroll
        for (int i = 0; i < n_terms<NP>(); ++i)
#pragma unroll
            for (int t = t0; t < (t0 + 2 < RT ? t0 + 2 : RT); ++t) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][KB - 1][t], B.p[term_b<NP>(i)][KB - 1], OUT[t]);
#else
    // the RT accumulator chains interleaved
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < n_terms<NP>(); ++i)
#pragma unroll
            for (int t = 0; t < RT; ++t) OUT[t] = mfma_pieces<C>(A.a[term_a<NP>(i)][kb][t], B.p[term_b<NP>(i)][kb], OUT[t]);
#endif
}
template <class C, int RT, bool LIVE1 = false>
__device__ inline void panel_product_split(const typename C::T *form, const typename C::T *form1, int lane,
                                           const typename C::acc_t (&IN)[RT], typename C::acc_t (&OUT)[RT],
                                           const typename C::acc_t &last_init) {
    SplitPanel<RT, C::NP> B;
    split_panel<C, RT, LIVE1>(IN, B);
    panel_product_pieces<C, RT>(form, form1, lane, B, OUT, last_init);
}

// Work queue of the fast launch at one / two row-tiles (K <= 32): 16-item batches are handed out by QUEUE_SHARDS ticket counters on
// cache lines of their own (counter c hands out the batches c, c + QUEUE_SHARDS, ...).  One counter for all serialises: 11.4 ns of one
// L2 atomic unit per draw whatever the batch costs -- the reference test's cohort (634 x 14, 25 000 draws: 0.29 of the main kernel's
// 0.35 ms) and every grid of a few cell types ran at 1.4e9 pairs/s at most, and every wave's last, empty draw queued on the same
// address (6 144 waves: 70 us).  From three row-tiles on a batch costs more than its draw and the single counter stays: with the
// shards c3 / c4 measured +1 % / +4 % (a wave gives up after four empty counters: the long batches of the last counters are left to
// fewer waves), and a variant that flags empty counters and steals from any live one cost the small kernels a wave per SIMD in
// registers (profiles/r06/ab_experiments.md section 9).
constexpr int QUEUE_SHARD_MAX_RT = 2;
constexpr int QUEUE_SHARDS = 32, QUEUE_SHARD_STRIDE = 32;
struct GridParams {
    const void *P;        // N x KP, element type T, every row in accumulator-slot order [tile][group][reg], 0 in padding;
                          // then N stop thresholds (one per column patient)
    const void *img;      // operand block of the call: see img_layout below
    int N, K;
    int n_pairs;          // number of work items
    const int *list;      // nullable: explicit work-item list (indices into the output arrays)
    const int *list_len;  // nullable: device-side length of `list` (overrides n_pairs)
    int row_begin, row_step;
    int max_iter, period;
    double stop_thr, tau, floor_ulps;
    double *emd;          // outputs indexed by work item q = local_row * N + j
    int *iters;
    double *err;
    int *flags;
    int *track_list;      // fast kernel appends pairs that need POT absorption tracking
    int *track_count;
    int *queue_head;      // dynamic work queue of this launch (zeroed by the host before the launch)
    int *queue_shards;    // nullable (the fast launch): QUEUE_SHARDS ticket counters, QUEUE_SHARD_STRIDE ints apart, zeroed by the host --
                          //   used instead of queue_head, see the draw in sinkhorn_stream_kernel
    const int *solo_len;  // fast launch: device-side number of leading list items (exact duplicates a == b) for solo_pairs
    int *solo_head;       //   their queue head
    int solo_blocks;      //   leading workgroups of the launch that run solo_pairs (they start first); 0: none
    int ring;             // slots of the per-wave LDS ring of finished pairs (1 .. RING_MAX; sized by the host to fit LDS)
    int bands;            // bf16-split tracking kernel: 2 = the Gibbs kernel in two exponent bands (small reg), else 1
    int *fb_list;         // nullable (bf16-split configuration, small reg): pairs that went NaN / inf in f32 are appended here
    int *fb_count;        //   instead of being written out, and the f64 kernel solves them again (see ring_flush)
    int *nan_list;        // nullable: pairs that end in NaN are appended here and re-solved by the POT-literal kernel, which
    int *nan_count;       //   reverts to the last good iterate like POT does (generic_kernels.hpp)
    int debug;            // experiment switches (PILOT_OT_DEBUG): bit0 no priority, bit1 no longest-first order
    const int *unequal;   // fp16-split configuration: control slot CTRL_UNEQUAL (see the refill of the stream kernel)
};

// Operand block of one call in global memory, in elements of T (4-byte units for the bf16-split configuration):
//   [form 0: G^T][form 1: G][form 2: G o M]   form_elems<C>(RT) each -- the stationary MFMA A operands in lane order
//   [first-product table: KP]                  (G^T u0)[slot], u0 = 1/K
//   [tail-row weights]                         2 forms x 2 chains x tail_steps x WAVE pairs (VALU tail rows)
//   [plain tables: 3 x 64 x WAVE]              G[lane][k], G[k][lane], (G o M)[lane][k] for solo_pairs
template <class C> __host__ __device__ constexpr int form_elems(int RT) {
    return C::SPLIT ? C::NP * ((RT + 1) / 2) * RT * WAVE * 4 : RT * C::TILE * RT * C::TILE;
}
template <class C> __host__ __device__ constexpr int acc0_offset(int RT) { return 3 * form_elems<C>(RT); }
template <class C> __host__ __device__ constexpr int tail_offset(int RT) { return acc0_offset<C>(RT) + RT * C::TILE; }
template <class C> __host__ __device__ constexpr int plain_offset(int RT) { return tail_offset<C>(RT) + 2 * 2 * ((RT - 1) * 4 + 1) * WAVE * 2; }
// [band-1 forms 0..2] (bf16-split configuration, small reg): the Gibbs kernel in TWO EXPONENT BANDS.  An f32 (or a 3-way
// bf16 split of it) represents exp(-M/reg) faithfully only above SPLIT_SAFE_MIN; at reg = 0.01 a fifth of the entries of
// the 600 x 50 benchmark lie below and most optimal plans put 1e-8 .. 1e-2 of their mass on them.  Band 0 keeps the
// entries >= SPLIT_SAFE_MIN (others 0), band 1 holds the others multiplied by 2^128, and a product is
// X0 * in + 2^-128 (X1 * in): both partial products are f32-accurate, so entries down to 2^-238 = exp(-165) take part
// with full precision -- the range POT's log-absorption covers, without a per-pair kernel matrix.
template <class C> __host__ __device__ constexpr int band1_offset(int RT) { return plain_offset<C>(RT) + 3 * 64 * WAVE; }
template <class C> __host__ __device__ constexpr int img_total_plain(int RT) { return band1_offset<C>(RT) + ((C::SPLIT && !C::HALF) ? 3 * form_elems<C>(RT) : 0); }
// [tracking block] (fp16-split configuration): a complete bf16-split operand block behind the fp16 one -- pairs in which POT
// would tau-absorb leave the scaled fp16 domain and are redone by the bf16-split tracking kernel
template <class C> __host__ __device__ constexpr int track_img_offset(int RT) { return C::HALF ? band1_offset<C>(RT) : 0; }
template <class C> __host__ __device__ constexpr int img_total(int RT) { return img_total_plain<C>(RT) + (C::HALF ? img_total_plain<CfgS32x16>(RT) : 0); }


// ---- one wave per pair: the exact duplicates ---------------------------------------------------------------------------
// Pairs with a == b (the diagonal of the grid, duplicate patients) need the most updates by far (161 .. 381 at c3 against
// a mean of 39) and a pair's updates are a serial chain, so on a small or sharded grid they ARE the run time.  A 16-pair
// MFMA tile spends 2 x (ceil(K/4) dependent MFMAs + an LDS panel exchange) per update on them; here one wavefront owns
// one pair instead: lane i holds row i of G (and of G^T) in registers, the scaling vector lives one element per lane and
// is broadcast k by k through v_readlane into an SGPR operand of v_fmac -- no LDS, no barrier, ~2.5x shorter update.
// Selection is by value (a == b bit for bit: top bucket of the order key), never by load, so which kernel solves a pair
// does not depend on the row subset or on the launch size; K <= 64.
template <typename T> __device__ inline T readlane_t(T x, int k);
template <> __device__ inline float readlane_t<float>(float x, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
}
template <> __device__ inline double readlane_t<double>(double x, int k) {
    union { double d; int u[2]; } v;
    v.d = x;
    v.u[0] = __builtin_amdgcn_readlane(v.u[0], k);
    v.u[1] = __builtin_amdgcn_readlane(v.u[1], k);
    return v.d;
}
template <typename T> __device__ inline T wave_sum(T x) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) x += __shfl_xor(x, off);
    return sum_xor32(sum_xor16(x));
}

// TRACK: POT's tau-absorptions are kept books of exactly as in the tracking stream kernel (reference scalings ru, rv per
// element, the 1/K resets folded into the error test and the final cost) instead of handing the pair over; used for the
// short f64 hand-over list of the small-reg path, where a 16-pair f64 MFMA tile would run 1000 updates nearly empty.
// NCH = ceil(K / 16) (== the row-tile count) is a template parameter: the matrix-vector product is straight-line code with
// ALL its broadcast reads issued up front (a chunk loop with a wave-uniform `k0 < K` test puts one LDS round trip in front of
// every 16 multiply-adds).
template <class C, bool SYM, int NCH, bool TRACK = false>
__device__ inline void solo_pairs(const GridParams &p, unsigned char *solo_smem, const int *n_items_ptr, int *head_ptr) {
    constexpr int RT = NCH, KP = NCH * C::TILE;
    using T = typename C::T;
    const int lane = threadIdx.x % WAVE;
    const int K = p.K, N = p.N;
    const T *img = static_cast<const T *>(p.img);
    const T *plain = img + plain_offset<C>(RT);
    const T *Pt = static_cast<const T *>(p.P);
    constexpr int KC = NCH * 16;          // contraction length, padded (the tables hold 0 beyond K)
    // (rows as explicit pairs: every multiply-add below is one v_pk_fma_f32 on adjacent registers)
    using P2 = pair_of<T>;
    P2 g1[KC / 2], g2[SYM ? 1 : KC / 2];
#pragma unroll
    for (int k = 0; k < KC / 2; ++k) {
        g1[k] = P2{plain[(2 * k) * WAVE + lane], plain[(2 * k + 1) * WAVE + lane]};
        if constexpr (!SYM) g2[k] = P2{plain[64 * WAVE + (2 * k) * WAVE + lane], plain[64 * WAVE + (2 * k + 1) * WAVE + lane]};
    }
    // y_lane = sum_k g[k] * x_k: x goes through a wave-private LDS line and comes back as broadcast 16-byte reads (half the
    // instructions of a v_readlane per element; the wave is issue-bound)
    T *xline = reinterpret_cast<T *>(solo_smem) + (threadIdx.x / WAVE) * WAVE;
    using V4 = T __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr int VE = 16 / sizeof(T);
    auto matvec = [&](const P2 (&g)[KC / 2], T x) {
        xline[lane] = x;
        constexpr int NV = NCH * 16 / VE;
        V4 xv[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) xv[c] = *reinterpret_cast<const V4 *>(xline + c * VE);
        P2 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = P2{T(0), T(0)};
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int h = 0; h < VE / 2; ++h) {
                const int k2 = c * (VE / 2) + h;
                const P2 xx = {xv[c][2 * h], xv[c][2 * h + 1]};
                acc[k2 & 3] = __builtin_elementwise_fma(g[k2], xx, acc[k2 & 3]);
            }
        const P2 s2 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        return s2[0] + s2[1];
    };
    const bool live = lane < K;
    const int pos = ((lane / 16) * 4 + (lane % 4)) * 4 + (lane % 16) / 4;     // accumulator slot of cell type `lane`
    const bool inrec = lane < KP;
    // (fp16-split configuration: the first-product table is stored in the scaled domain of the tile kernel, 2^20 G^T u0)
    const T acc0 = inrec ? img[acc0_offset<C>(RT) + pos] * (C::HALF ? T(1) / (T(H_GIBBS_SCALE) * T(H_PANEL_SCALE)) : T(1)) : T(1);
    const T tau = T(p.tau);
    const int n_items = *n_items_ptr;
    const T kk = T(K) * T(K);
    for (;;) {
        int item = 0;
        if (lane == 0) item = __hip_atomic_fetch_add(head_ptr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        const int q = p.list[item];
        const int i = p.row_begin + (q / N) * p.row_step, j = q % N;
        const T a = inrec ? Pt[(size_t)i * KP + pos] : T(0), b = inrec ? Pt[(size_t)j * KP + pos] : T(0);
        const T thr = Pt[(size_t)N * KP + j];
        T u = live ? T(1) / T(K) : T(0), v = u, ACC = acc0, errv = T(1);
        T ru = live ? T(1) : T(0), rv = ru;                      // TRACK: u * ru, v * rv are POT's residual scalings
        int ii = 0, chk = 1, flags = sizeof(T) == 8 ? FLAG_F64 : 0, abs_at = -1;
        for (;;) {
            v = b * C::rcp(ACC);
            T r1 = SYM ? matvec(g1, v) : matvec(g1, v);
            r1 = live ? r1 : T(1);
            u = a * C::rcp(r1);
            if constexpr (TRACK) {
                if (ballot_b(live && (u * ru > tau || v * rv > tau))) {     // POT: absorb, u = v = 1/K (see the stream kernel)
                    ru = live ? C::rcp(u * T(K)) : T(0);
                    rv = live ? T(K) * C::rcp(v) : T(0);
                    abs_at = ii;
                    flags |= FLAG_ABSORBED;
                    {   // empty bins / scalings out of the exact range: see the stream kernel
                        constexpr T BIG = sizeof(T) == 4 ? T(1.2676506e30) : T(8.452712498170644e270);
                        constexpr T SMALL = T(1) / BIG;
                        if (ballot_b(live && !(u >= SMALL && u < BIG && v >= SMALL && v < BIG))) u = T(__builtin_nanf(""));
                    }
                }
            } else {
                if (ballot_b(live && (u > tau || v > tau))) {      // POT would absorb: the tracking kernel redoes the pair
                    if (lane == 0) p.track_list[__hip_atomic_fetch_add(p.track_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = q;
                    break;
                }
            }
            ++ii;
            if constexpr (SYM) ACC = matvec(g1, u); else ACC = matvec(g2, u);
            ACC = live ? ACC : T(1);
            const bool pending = ii == chk, capped = ii >= p.max_iter;
            if (pending) chk += p.period;
            if (pending || capped) {                              // wave-uniform
                const T sc = (TRACK && abs_at == ii - 1) ? T(1) / kk : T(1);      // u, v were just reset to 1/K each
                const T d = v * ACC * sc - b;
                const T e = sqrt(wave_sum(d * d));
                bool fin = capped;
                if (pending) {
                    errv = e;
                    if (e <= thr) { fin = true; flags |= FLAG_CONVERGED; }
                    else if (e != e) { fin = true; flags |= FLAG_NAN; }
                }
                if (fin) {
                    // cost <Gamma, M> = u^T (G o M) v: the rows of G o M take the place of G's for one product
                    xline[lane] = v;
                    T val = T(0);
                    {
                        const T *gm = plain + 2 * 64 * WAVE + lane;
                        for (int k0 = 0; k0 < K; k0 += 8) {       // 8 coalesced row loads in flight
                            T gk[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) gk[e] = gm[(k0 + e) * WAVE];
#pragma unroll
                            for (int e = 0; e < 8; ++e) val = fma(gk[e], xline[k0 + e], val);
                        }
                    }
                    val = wave_sum(live ? u * val : T(0));
                    if (TRACK && abs_at >= 0 && abs_at == ii - 1) { val *= T(1) / kk; flags |= FLAG_ABSORB_LAST; }
                    if (lane == 0) {
                        if (!(val - val == T(0))) flags |= FLAG_NAN;     // NaN or inf
                        if (p.fb_list && (flags & FLAG_NAN) && sizeof(T) == 4) {          // out of the f32 range: the f64 pass
                            p.fb_list[__hip_atomic_fetch_add(p.fb_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = q;
                        } else if (p.nan_list && (flags & FLAG_NAN)) {
                            p.nan_list[__hip_atomic_fetch_add(p.nan_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = q;
                        } else {
                            p.emd[q] = double(val);
                            if (p.iters) p.iters[q] = ii;
                            if (p.err) p.err[q] = double(errv);
                            p.flags[q] = flags;
                        }
                    }
                    break;
                }
            }
        }
    }
}

// A (short) list of pairs, one wavefront per pair with POT's absorptions tracked: the f64 hand-over list of the small-reg
// path (symmetric cost, K <= 64).  p.list / p.list_len / p.queue_head name the list.
template <class C>
__global__ void __launch_bounds__(WAVE * WAVES_PER_WG) sinkhorn_solo_track_kernel(GridParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char solo_track_smem[];
    switch ((p.K + C::TILE - 1) / C::TILE) {       // wave-uniform
    case 1: solo_pairs<C, true, 1, true>(p, solo_track_smem, p.list_len, p.queue_head); break;
    case 2: solo_pairs<C, true, 2, true>(p, solo_track_smem, p.list_len, p.queue_head); break;
    case 3: solo_pairs<C, true, 3, true>(p, solo_track_smem, p.list_len, p.queue_head); break;
    default: solo_pairs<C, true, 4, true>(p, solo_track_smem, p.list_len, p.queue_head); break;
    }
}

// Finished pairs in a wave's ring -> costs <Gamma, M> = u^T (G o M) v: ONE panel product for up to TILE pairs, one output
// row-tile at a time.  Deliberately NOT inlined: it runs once per 16 finished pairs, and as a call its register needs
// (the v panel, an accumulator tile, operands in flight) are paid at the call site instead of raising the pressure of
// the update loop around it.
//
// Small reg in f32 (bf16-split configuration): with p.bands == 2 the cost product uses both exponent bands of G o M; with
// p.fb_list set, a pair whose cost is not finite is not written out but appended to fb_list for the f64 kernel.
// V_PIECES (fp16-split configuration): the v half of a slot holds packed pieces (the records of the eight-wave kernel, whose waves
// never hold a whole f32 panel) instead of f32 values (the ring of the one-wave kernel)
template <class C, int RT, bool V_PIECES = false>
__device__ inline __attribute__((always_inline)) void ring_flush_body(const typename C::T *ring, const GridParams &p, int cnt) {
    using M = C;
    using T = typename C::T;
    using acc_t = typename C::acc_t;
    constexpr int TILE = C::TILE, NREG = C::NREG, NGRP = C::NGRP;
    constexpr int KP = RT * TILE;
    constexpr int RSTRIDE = ring_slot_stride<C>(RT);
    const int lane = threadIdx.x % WAVE;
    const int col = lane % TILE, grp = lane / TILE;
    const T *img_gm = static_cast<const T *>(p.img) + 2 * form_elems<C>(RT);     // read from L2, once per 16 finished pairs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int s = col < cnt ? col : cnt - 1;        // columns beyond the fill level redo the last slot, unused
    const T *rec = ring + s * RSTRIDE;
    constexpr int PE = ring_panel_elems<C>(RT);
    const T scale = rec[2 * PE];
    // one output row-tile at a time: only the v panel (or its bf16 pieces) is live
    T val = T(0), val1 = T(0);
    if constexpr (C::HALF) {
        // the ring holds u as packed pieces (u = hi + lo) and v as f32 values: v's pieces -- the B operand of the cost product -- are
        // formed again here, once per 16 finished pairs, with the update loop's own instruction sequence (same bits).  In the
        // update loop v's pieces are then dead once the product G v is issued and share their registers with u's (7 row-tiles:
        // 32 registers, which the loop used to spill; profiles/r06/ab_experiments.md)
        constexpr int KB = split_kblocks(RT);
        SplitPanel<RT, 2> Bv;
        if constexpr (V_PIECES) {
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    Bv.p[part][kb] = *reinterpret_cast<const u32x4_t *>(rec + PE + ((part * KB + kb) * NGRP + grp) * 4);
        } else {
            acc_t Vr[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) load_regs<C>(rec + PE + (t * NGRP + grp) * NREG, Vr[t]);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int t = 2 * kb + h / 2, e = 2 * (h & 1);
                    unsigned int hi = 0u, lo = 0u;
                    if (t < RT) quot_pieces(float(Vr[t][e]), float(Vr[t][e + 1]), hi, lo);      // (padded and dead slots hold 0: pieces 0)
                    Bv.p[0][kb][h] = hi; Bv.p[1][kb][h] = lo;
                }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc_t zero;
#pragma unroll
            for (int r = 0; r < NREG; ++r) zero[r] = T(0);
            const acc_t w = split_tile_product<C, RT>(img_gm, lane, t, Bv, zero);
            using u32x2_t = unsigned int __attribute__((ext_vector_type(2)));
            const u32x2_t uh = *reinterpret_cast<const u32x2_t *>(rec + ((0 * KB + t / 2) * NGRP + grp) * 4 + 2 * (t & 1));
            const u32x2_t ul = *reinterpret_cast<const u32x2_t *>(rec + ((1 * KB + t / 2) * NGRP + grp) * 4 + 2 * (t & 1));
            val += pieces_sum_lo(uh[0], ul[0]) * w[0];
            val += pieces_sum_hi(uh[0], ul[0]) * w[1];
            val += pieces_sum_lo(uh[1], ul[1]) * w[2];
            val += pieces_sum_hi(uh[1], ul[1]) * w[3];
        }
    } else if constexpr (C::SPLIT) {
        SplitPanel<RT, C::NP> Bv;
        {
            acc_t Vr[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) load_regs<C>(rec + KP + (t * NGRP + grp) * NREG, Vr[t]);
            split_panel<C, RT>(Vr, Bv);
        }
        const T *img_gm1 = static_cast<const T *>(p.img) + band1_offset<C>(RT) + 2 * form_elems<C>(RT);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc_t zero;
#pragma unroll
            for (int r = 0; r < NREG; ++r) zero[r] = T(0);
            acc_t w = split_tile_product<C, RT>(img_gm, lane, t, Bv, zero);
            acc_t ur;
            load_regs<C>(rec + (t * NGRP + grp) * NREG, ur);
            if (p.bands == 2) {                       // wave-uniform
                const acc_t w1 = split_tile_product<C, RT>(img_gm1, lane, t, Bv, zero);
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    w[r] = fmaf(w1[r], BAND1_DOWN, w[r]);
                    val1 += ur[r] * (w1[r] * BAND1_DOWN);     // the part of the cost that sits on band-1 arcs
                }
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r) val += ur[r] * w[r];
        }
    } else {
        acc_t Vr[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) load_regs<C>(rec + KP + (t * NGRP + grp) * NREG, Vr[t]);
        const int n_last = (p.K - M::lidx(RT - 1, 0, 0) + NGRP - 1) / NGRP;     // live k-steps of the last row-tile
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc_t w;
#pragma unroll
            for (int r = 0; r < NREG; ++r) w[r] = T(0);
#pragma unroll
            for (int tp = 0; tp < RT; ++tp)
#pragma unroll
                for (int r = 0; r < NREG; ++r)
                    if (tp < RT - 1 || r < n_last)        // wave-uniform
                        w = M::mfma(img_gm[((tp * NREG + r) * RT + t) * WAVE + lane], Vr[tp][r], w);
            acc_t ur;
            load_regs<C>(rec + (t * NGRP + grp) * NREG, ur);
#pragma unroll
            for (int r = 0; r < NREG; ++r) val += ur[r] * w[r];
        }
    }
    val = group_sum<C>(val) * scale;
    if constexpr (C::HALF) val *= T(1) / T(H_IN_SCALE);     // u~^T (2^15 G o M) v~ = 2^25 u^T (G o M) v
    bool redo = p.fb_list && !(val - val == T(0));             // NaN or inf: out of the f32 range somewhere along the way
    if constexpr (C::SPLIT && !C::HALF) {
        // Two exponent bands: the update loop takes the band-1 partial products from two bf16 pieces (16 bits), which is
        // rounding noise of 2^-16 on THAT part of a product.  Where band-1 arcs carry more than a quarter of a pair's cost
        // (tiny K: every off-diagonal arc can be a band-1 arc; the 600 x 50 benchmark: 1e-8 .. 1e-2) that is no longer
        // below the f32 tolerance (fuzz: 1.2e-5 at K = 2, 3) and the pair is solved again by the f64 pass.
        if (p.bands == 2 && p.fb_list) redo = redo || group_sum<C>(val1) * scale > T(0.25) * val;
    }
    if (grp == 0 && col < cnt) {
        const int *meta = reinterpret_cast<const int *>(rec + 2 * PE + 1);
        const int qq = meta[0];
        int fl = meta[1];
        if (redo || (p.fb_list && (fl & FLAG_NAN))) {
            p.fb_list[__hip_atomic_fetch_add(p.fb_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = qq;
        } else if (p.nan_list && (!(val - val == T(0)) || (fl & FLAG_NAN))) {       // NaN or inf
            p.nan_list[__hip_atomic_fetch_add(p.nan_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = qq;
        } else {
            if (val != val) fl |= FLAG_NAN;
            p.emd[qq] = double(val);
            p.flags[qq] = fl;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the slots are reused only after every lane has read them
    __builtin_amdgcn_wave_barrier();
}
template <class C, int RT>
__device__ __attribute__((noinline)) void ring_flush(const typename C::T *ring, const GridParams &p, int cnt) {
    ring_flush_body<C, RT>(ring, p, cnt);
}
// PARKED flush (split configurations, fast kernel, RT <= 4): the call above costs the kernel a stack (the callee saves the
// registers it borrows: 224 B per lane, written back to HBM once per wave -- 37 MB per c3 launch).  Here the flush is inlined
// instead and the wave makes room for it itself: U goes to a wave-private LDS line, A and B are read again from the slot-
// ordered proportions (L2) and ACC = G^T u is recomputed with one more product (bit-identical: same inputs, same
// instruction sequence), V is dead at the flush point.  No scratch memory at all.
template <class C, int RT, bool TRACK> constexpr bool parked_flush() { return C::SPLIT && !TRACK && RT <= 4; }

constexpr int HANDOVER_BUF = 32, HANDOVER_FLUSH = 16;      // per-wave hand-over buffer of the fast kernels (ints), flush level
constexpr int GREG_MAX = 64;
#ifndef PILOT_AREG_MAX_RT
#define PILOT_AREG_MAX_RT 4
#endif
#ifndef PILOT_SPLIT_OCC2_MAX_RT
#define PILOT_SPLIT_OCC2_MAX_RT 4
#endif
#ifndef PILOT_SPLIT_OCC2_MAX_RT_TRACK
#define PILOT_SPLIT_OCC2_MAX_RT_TRACK 4      // (K = 80 / 96 at reg 0.01: 160 -> 134 ms, 189 -> 146 ms with one wave and no spills; RT = 4: 32.5 -> 47.5 ms)
#endif
#ifndef PILOT_HALF_OCC2_MAX_RT
#define PILOT_HALF_OCC2_MAX_RT 7             // (fp16-split, piece state: c4 at K = 100 26.85 -> 26.07 ms with two waves and 144 B of spills)
#endif
#ifndef PILOT_HALF_SOLO_MIN_RT
#define PILOT_HALF_SOLO_MIN_RT 1         // (3: duplicates of the fp16-split configuration stay in tiles up to K = 32)
#endif
constexpr int HALF_SOLO_MIN_RT = PILOT_HALF_SOLO_MIN_RT;
constexpr int HALF_OCC4_MAX_RT = 2;
constexpr int SPLIT_OCC2_MAX_RT = PILOT_SPLIT_OCC2_MAX_RT, SPLIT_OCC2_MAX_RT_TRACK = PILOT_SPLIT_OCC2_MAX_RT_TRACK, HALF_OCC2_MAX_RT = PILOT_HALF_OCC2_MAX_RT;
// live panel registers per lane: A, B, U, V, ACC (+ RU, RV when tracking)
// the operand image is kept in registers when it needs <= 64 VGPRs per lane and the cost is symmetric
template <class C, int RT, bool SYM> constexpr bool operands_in_regs() {
    return !C::SPLIT && SYM && RT * C::NREG * RT * int(sizeof(typename C::T) / 4) <= GREG_MAX;
}
template <class C, int RT, bool SYM, bool TRACK, int TV = 0> constexpr int panel_regs() {
    return (TRACK ? 7 : 5) * RT * C::NREG * int(sizeof(typename C::T) / 4) + C::NREG * int(sizeof(typename C::T) / 4) + 56 +
           (TV > 0 ? 24 * int(sizeof(typename C::T) / 4) : 0) +     // tail accumulators, broadcast pairs, weights in flight
           (C::SPLIT ? 3 * ((RT + 1) / 2) * 4 + 24 : 0) +           // split panel parts + operand parts in flight
           (operands_in_regs<C, RT, SYM>() ? (RT * C::NREG * RT + 2 * TV * tail_steps<RT>()) * int(sizeof(typename C::T) / 4) : 0);
}
template <class C, int RT, bool SYM, bool TRACK, int TV = 0> constexpr int min_waves_per_simd() {
    // split variants: two waves per SIMD up to SPLIT_OCC2_MAX_RT row tiles (tracking variants: SPLIT_OCC2_MAX_RT_TRACK), one wave
    // with the whole register file beyond (3 waves per SIMD at RT <= 4: slower)
    // (fp16-split fast kernel at one / two row-tiles: four -- 82 / 116 registers; with the sharded work queue the 634 x 14 cohort runs
    // 0.242 / 0.195 / 0.185 / 0.193 ms at 2 / 3 / 4 / 6 workgroups per CU, K = 16 .. 32 -12 .. -22 %: tools/small_k_occupancy_probe.py)
    if (C::HALF && !TRACK && RT <= HALF_OCC4_MAX_RT) return 4;
    if (C::SPLIT) return RT <= (TRACK ? SPLIT_OCC2_MAX_RT_TRACK : (C::HALF ? HALF_OCC2_MAX_RT : SPLIT_OCC2_MAX_RT)) ? 2 : 1;
    return panel_regs<C, RT, SYM, TRACK, TV>() <= 128 ? 4 : (panel_regs<C, RT, SYM, TRACK, TV>() <= 168 ? 3 : (panel_regs<C, RT, SYM, TRACK, TV>() <= 256 ? 2 : 1));
}

// solo_pairs (64 + ~45 registers of T per lane) rides in the fast launch when the cost is symmetric (PILOT's always is),
// K <= 64, and the launch's register budget holds it without spilling
template <class C, int RT, bool SYM, bool TRACK, int TV> constexpr bool solo_in_stream() {
    constexpr int budget = min_waves_per_simd<C, RT, SYM, TRACK, TV>() >= 4 ? 128 : (min_waves_per_simd<C, RT, SYM, TRACK, TV>() == 3 ? 168 : 256);
    // (round 3 had the fp16-split configuration keep its duplicates in tiles up to K = 32, when a tile's update was shorter
    // than the one-wave-per-pair update; with the straight-line matrix-vector product it is the other way round again:
    // c2 kernel 0.154 -> 0.136 ms, the 1/8 shard of c3 0.253 -> 0.186 ms.  A rule by SHAPE, never by load: the same pair takes
    // the same path in every shard.)
    return !TRACK && SYM && RT <= 4 && !(C::HALF && RT < HALF_SOLO_MIN_RT) && (64 + 45) * int(sizeof(typename C::T) / 4) <= budget;
}

// TRACK = false: plain scaling iterations; a pair whose POT residual scaling would exceed tau (i.e. POT
//                would absorb) is handed to the TRACK = true kernel via track_list.
// TRACK = true : additionally carries the reciprocal reference scalings so the iteration at which every
//                POT absorption happens is known (needed for POT's err-after-absorption and
//                plan/(K*K)-on-the-final-update behaviour; see oracle/pilot_oracle.c).
template <class C, int RT, bool SYM, bool TRACK, int TV = 0>
__global__ void __launch_bounds__(WAVE * WAVES_PER_WG, (min_waves_per_simd<C, RT, SYM, TRACK, TV>()))
sinkhorn_stream_kernel(GridParams p) {
    using M = C;
    using T = typename C::T;
    using acc_t = typename M::acc_t;
    static_assert(!(C::HALF && TRACK), "the fp16-split configuration has no tracking variant (scalings beyond tau leave its scaled domain)");
    constexpr int TILE = M::TILE, NREG = M::NREG, NGRP = M::NGRP;
    constexpr int KP = RT * TILE;
    constexpr int FORM = form_elems<C>(RT);
    constexpr int RSTRIDE = ring_slot_stride<C>(RT);
    // fp16-split configuration: panels are 32 u, 32 v; a, b, the threshold and the error carry 2^25 (see CfgH32x16)
    constexpr T PANEL_SCALE = C::HALF ? T(H_PANEL_SCALE) : T(1), IN_SCALE = C::HALF ? T(H_IN_SCALE) : T(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw);

    int block = blockIdx.x;
    if constexpr (solo_in_stream<C, RT, SYM, TRACK, TV>()) {
        // the leading workgroups of the fast launch run the exact-duplicate pairs, one per wave (they start first)
        if (block < p.solo_blocks) { solo_pairs<C, SYM, RT>(p, smem_raw, p.solo_len, p.solo_head); return; }
        block -= p.solo_blocks;
    }
    const int n_items = p.list_len ? *p.list_len : p.n_pairs;
    const int lane = threadIdx.x % WAVE;
    if (block * WAVES_PER_WG * TILE >= n_items) return;  // more workgroups than work (tracking launch)

    // stage the stationary operand: image 0 (and image 1 unless G is symmetric) + first-product table
    // (bf16-split tracking kernel with two exponent bands: the band-1 images follow the band-0 ones)
    const bool two_bands = C::SPLIT && TRACK && p.bands == 2;
    const int n_img = (SYM ? 1 : 2) * FORM * (two_bands ? 2 : 1);
    constexpr int n_tail = (TV > 0 && !C::SPLIT) ? (SYM ? 1 : 2) * TV * tail_steps<RT>() * WAVE * 2 : 0;
    {
        const T *g = static_cast<const T *>(p.img);
        constexpr int n_b0 = (SYM ? 1 : 2) * FORM;
        for (int i = threadIdx.x; i < n_b0; i += WAVE * WAVES_PER_WG) lds[i] = g[i];
        if (two_bands)
            for (int i = threadIdx.x; i < n_b0; i += WAVE * WAVES_PER_WG) lds[n_b0 + i] = g[band1_offset<C>(RT) + i];
        for (int i = threadIdx.x; i < KP; i += WAVE * WAVES_PER_WG) lds[n_img + i] = g[acc0_offset<C>(RT) + i];
        if constexpr (TV > 0 && !C::SPLIT) {   // tail-row weights (chains 0 .. TV-1 of form 0, and of form 1 unless symmetric)
            constexpr int n_form = TV * tail_steps<RT>() * WAVE * 2;           // floats per form actually used
            const T *tg = g + tail_offset<C>(RT);
            for (int i = threadIdx.x; i < n_form; i += WAVE * WAVES_PER_WG) {
                lds[n_img + KP + i] = tg[i];
                if constexpr (!SYM) lds[n_img + KP + n_form + i] = tg[2 * tail_form_stride<RT>() + i];
            }
        }
    }
    __syncthreads();
    const T *img_gt = lds;                                         // out = G^T in
    const T *img_g = SYM ? lds : lds + FORM;                       // out = G in
    // wave-private ring of finished pairs (u, v panels + scale / output index / flags in the slot's padding)
    T *ring = lds + n_img + KP + n_tail + (threadIdx.x / WAVE) * p.ring * RSTRIDE;
    constexpr bool PARK = parked_flush<C, RT, TRACK>();
    // (PARK) one 16-byte line per lane and row-tile behind the rings: U while the inlined flush runs
    // (fp16-split configuration: the packed pieces of U, [part][k-block] x 16 bytes per lane)
    constexpr int PARK_LANE = park_lane_elems<C>(RT);
    T *park = lds + n_img + KP + n_tail + WAVES_PER_WG * p.ring * RSTRIDE + (threadIdx.x / WAVE) * (PARK_LANE * WAVE) + (threadIdx.x % WAVE) * 4;
    // hand-over buffer (fast kernels): pairs in which POT would tau-absorb wait here, HANDOVER_BUF to a wave, and go to the
    // tracking list HANDOVER_FLUSH or more at a time.  One atomic on track_count per hand-over was 11 ns of one L2 atomic
    // unit per pair -- at K = 2, where a third of the 360 000 pairs absorb, 1.2 of the fast launch's 1.27 ms
    // (profiles/r04/small_k_scaling.txt).
    static_assert(HANDOVER_FLUSH - 1 + C::TILE <= HANDOVER_BUF, "a wave's hand-over buffer must hold one flush level plus one tile of pairs");
    int *hb = reinterpret_cast<int *>(lds + n_img + KP + n_tail + WAVES_PER_WG * p.ring * RSTRIDE + (PARK ? WAVES_PER_WG * PARK_LANE * WAVE : 0)) +
              (threadIdx.x / WAVE) * HANDOVER_BUF;
    int hb_cnt = 0;
    auto hb_flush = [&]() {
        int base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(p.track_count, hb_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < hb_cnt) p.track_list[base + lane] = hb[lane];
        hb_cnt = 0;
    };
    // small symmetric problems keep the whole operand image in registers (no LDS access in the loop)
    constexpr int NA = RT * NREG * RT;
    constexpr bool GREG = operands_in_regs<C, RT, SYM>();
    AFromRegs<C, GREG ? NA : 1> areg;
    if constexpr (GREG) {
#pragma unroll
        for (int i = 0; i < NA; ++i) areg.a[i] = lds[i * WAVE + lane];
    }
    const AFromImage<C> a_gt{img_gt, lane}, a_g{img_g, lane};
    const T *acc0 = lds + n_img;                                   // G^T u0, u0 = 1/K (a new pair's first product)
    const int col = lane % TILE, grp = lane / TILE;
    // tail-row weights: LDS images, or registers next to the register-resident operand image
    constexpr int NTF = (TV > 0 ? TV : 1) * tail_steps<RT>() * WAVE;            // pairs per form in LDS
    const pair_of<T> *tl_base = reinterpret_cast<const pair_of<T> *>(acc0 + KP);
    const TailFromImage<T, RT> w_gt{tl_base, lane}, w_g{SYM ? tl_base : tl_base + NTF, lane};
    TailFromRegs<T, RT, (GREG && TV > 0) ? TV : 0> wreg;
    if constexpr (GREG && TV > 0 && !C::SPLIT) {
#pragma unroll
        for (int i = 0; i < TV * tail_steps<RT>(); ++i) wreg.a[i] = tl_base[i * WAVE + lane];
    }
    // all-padding registers of the last row-tile (TV > 0: only register 0 is live) are skipped by the element-wise loops
    auto dead = [](int t, int r) { return TV > 0 && t == RT - 1 && r > 0; };
    auto product = [&](const AFromImage<C> &a_img, const TailFromImage<T, RT> &w_img, const acc_t (&IN)[RT], acc_t (&OUT)[RT],
                       const acc_t &init) {
        if constexpr (C::SPLIT) {
            if constexpr (TRACK) panel_product_split<C, RT, (TV > 0)>(a_img.img, two_bands ? a_img.img + (SYM ? 1 : 2) * FORM : nullptr, lane, IN, OUT, init);
            else panel_product_split<C, RT, (TV > 0)>(a_img.img, nullptr, lane, IN, OUT, init);
        } else if constexpr (TV > 0) {
            if constexpr (GREG) panel_product_tail<C, RT, TV>(areg, wreg, IN, OUT, init, grp);
            else panel_product_tail<C, RT, TV>(a_img, w_img, IN, OUT, init, grp);
        } else {
            if constexpr (GREG) panel_product<C, RT>(areg, IN, OUT, p.K, init);
            else panel_product<C, RT>(a_img, IN, OUT, p.K, init);
        }
    };

    const int K = p.K, N = p.N;
    // (fp16-split configuration: the copy of the proportions in its scaled domain, written by the prep kernel behind the plain one)
    const T *Pt = static_cast<const T *>(p.P) + (C::HALF ? (size_t)p.N * (RT * M::TILE) + p.N : 0);
    const T uinit = PANEL_SCALE / T(K);
    // (fp16-split configuration: the test reads the leading fp16 piece, which is within 2^-11 of the scaling -- the
    // threshold is lowered by 2^-10 so that no scaling beyond tau is missed; the few pairs this sends over early are
    // iterated by the tracking kernel, which decides in f32)
    const T tau = T(p.tau) * PANEL_SCALE * (C::HALF ? T(1) - T(0.0009765625) : T(1));
    const T kk = T(K) * T(K);
    const unsigned long long colmask = (1ull << TILE) - 1ull;  // lanes of group 0

    acc_t A[RT], B[RT], U[RT], V[RT], ACC[RT];
    acc_t RU[TRACK ? RT : 1], RV[TRACK ? RT : 1];
    // fp16-split configuration: the scalings live as packed fp16 pieces, the B operands of the products as they are
    // (quot_pieces); U and V above are unused there
    constexpr int KB = split_kblocks(RT);
    constexpr bool LIVE1 = TV > 0;
    SplitPanel<RT, 2> PU, PV;
    // the operand image in registers (see panel_product_pieces_regs)
    constexpr bool AREG = C::HALF && SYM && RT <= PILOT_AREG_MAX_RT;
    SplitImage<AREG ? RT : 1, 2> AR;
    if constexpr (AREG) {
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int t = 0; t < RT; ++t) AR.a[part][kb][t] = reinterpret_cast<const u32x4_t *>(lds)[((part * KB + kb) * RT + t) * WAVE + lane];
    }
    // piece pairs of a new pair's u = 1/K (0 in padded slots)
    unsigned int pu0_hi = 0u, pu0_lo = 0u;
    if constexpr (C::HALF) { const Pieces<2> s0 = split_pair<C>(PANEL_SCALE / T(p.K), PANEL_SCALE / T(p.K)); pu0_hi = s0.p[0]; pu0_lo = s0.p[1]; }
    auto pieces_live = [](int kb, int h) { const int t = 2 * kb + h / 2; return t < RT && !(LIVE1 && t == RT - 1 && (h & 1)); };
    // mask of the live halves of piece pair (kb, h): padded slots of the last row-tile hold 0
    auto pad_mask = [&](int kb, int h, const acc_t &padc) -> unsigned int {
        const int t = 2 * kb + h / 2;
        if (t != RT - 1) return 0xffffffffu;
        const int e = 2 * (h & 1);
        return (padc[e] != T(0) ? 0u : 0x0000ffffu) | ((padc[e + 1] != T(0) || LIVE1) ? 0u : 0xffff0000u);
    };
    // P = the pieces of the panel X
    auto quot_panel = [&](const acc_t (&X)[RT], SplitPanel<RT, 2> &P) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int t = 2 * kb + h / 2, e = 2 * (h & 1);
                unsigned int hi = 0u, lo = 0u;
                if (pieces_live(kb, h)) {
                    const int tt = t < RT ? t : 0;
                    quot_pieces(float(X[tt][e]), (LIVE1 && t == RT - 1) ? 0.f : float(X[tt][e + 1]), hi, lo);
                }
                P.p[0][kb][h] = hi; P.p[1][kb][h] = lo;
            }
    };
    // 1 in the padded accumulator slots of the last row-tile, 0 elsewhere: seeds every product so that
    // b/acc and a/acc are 0/1 = 0 there (a, b are 0 in padded slots) without per-element selects
    acc_t PADC;
#pragma unroll
    for (int r = 0; r < NREG; ++r) PADC[r] = M::lidx(RT - 1, r, grp) >= K ? T(1) : T(0);
    auto product_h = [&](const T *form, const SplitPanel<RT, 2> &P, acc_t (&OUT)[RT]) {
        if constexpr (AREG) panel_product_pieces_regs<C, RT>(AR, P, OUT, PADC);
        else if constexpr (C::HALF) panel_product_pieces<C, RT>(form, nullptr, lane, P, OUT, PADC);
    };
    // per-column state (replicated in the lane groups of the column)
    bool active = false;
    int q = 0, ii = 0, chk = 1, flags = 0, abs_at = -1;
    T errv = T(1), thr = T(0);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            A[t][r] = B[t][r] = U[t][r] = V[t][r] = T(0);
            ACC[t][r] = T(1);
            if constexpr (TRACK) { RU[t][r] = RV[t][r] = T(0); }
        }
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int h = 0; h < 4; ++h) { PU.p[part][kb][h] = 0u; PV.p[part][kb][h] = 0u; }

    int ring_cnt = 0;
    auto flush = [&](int cnt) { ring_flush<C, RT>(ring, p, cnt); };
    // (the lambdas used below are defined after the per-column state they touch)

    // work queue: waves draw batches of TILE items from one device-wide counter, so a wave that got
    // long-running pairs simply draws fewer batches
    int res_next = 0, res_end = 0, res_base = 0, qbatch = 0, ibatch = 0, jbatch = 0;
    bool exhausted = false, first_draw = true;
    // (see the first batch of a wave below) position of the tile queue's first item: behind the duplicates of the solo waves
    const int wave_id = block * WAVES_PER_WG + (int)(threadIdx.x / WAVE);
    const int n_tile_waves = ((int)gridDim.x - (solo_in_stream<C, RT, SYM, TRACK, TV>() ? p.solo_blocks : 0)) * WAVES_PER_WG;
    const int queue_start = (solo_in_stream<C, RT, SYM, TRACK, TV>() && p.solo_len && p.solo_blocks > 0) ? *p.solo_len : 0;
    bool want = true;  // column asks for a (new) pair
    constexpr bool SHARDED_QUEUE = RT <= QUEUE_SHARD_MAX_RT;
    int qc = wave_id % QUEUE_SHARDS, q_tries = 0;                       // (sharded queue: the counter this wave draws from)
    const int q_max_tries = n_tile_waves >= QUEUE_SHARDS ? 4 : QUEUE_SHARDS;
    const bool hand_all_over = C::HALF && p.unequal && *p.unequal != 0;
    for (;;) {
        // ---- (re)fill columns: a new pair starts with u = v = 1/K and ACC = G^T u0 (table) ----------
        // (round 6, tried and dropped: refills -- and with them the error checks, which follow a refill by 1 + 20 n updates -- only on
        // every 2nd / 4th update of the wave, so that the refill and error blocks run in a quarter of the updates instead of 40 % /
        // 56 %: c3 kernel 0.609 -> 0.626 / 0.651 ms, the idle column-updates cost more than the skipped blocks; at one row-tile, where the
        // update itself is a few dozen instructions, the gate changes nothing either (the 634 x 14 cohort 0.350 / 0.351 / 0.347 ms at gates
        // 1 / 2 / 4, K = 20 .. 32 +2 .. +7 %): profiles/r06/ab_experiments.md sections 3 and 8)
        const unsigned long long wmask = ballot_b(want) & colmask;
        if (wmask) {
            if (res_next >= res_end && !exhausted) {
                // A wave's FIRST batch is the one of its own number (behind the exact duplicates the solo waves take), the later
                // ones come from the device-wide counter, which therefore hands out positions behind the statically dealt
                // part: the 2048 waves of a launch no longer start with 2048 atomics on one address (11.4 ns each, serial:
                // 23 us before the last wave had its first pair).
                int base = 0;
                if (first_draw) {
                    base = queue_start + wave_id * TILE;
                    first_draw = false;
                } else if (SHARDED_QUEUE && p.queue_shards) {
                    // ticket t of counter c is batch c + QUEUE_SHARDS t behind the statically dealt ones.  A wave starts on the counter
                    // of its own number and moves on when a counter has run out; every counter has waves that start on it and drain it
                    // (a launch with fewer waves than counters makes the full round), the batches of all counters interleave, so they run
                    // out within a round of each other and a wave gives up after a few empty ones
                    base = n_items;
                    while (q_tries < q_max_tries) {
                        int t = 0;
                        if (lane == 0) t = __hip_atomic_fetch_add(p.queue_shards + qc * QUEUE_SHARD_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        t = __builtin_amdgcn_readfirstlane(t);
                        const long b = (long)queue_start + ((long)n_tile_waves + qc + (long)QUEUE_SHARDS * t) * TILE;
                        if (b < n_items) { base = (int)b; q_tries = 0; break; }
                        ++q_tries;
                        qc = qc + 1 == QUEUE_SHARDS ? 0 : qc + 1;
                    }
                } else {
                    if (lane == 0) base = __hip_atomic_fetch_add(p.queue_head, TILE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    base = __builtin_amdgcn_readfirstlane(base) + n_tile_waves * TILE;
                }
                exhausted = base >= n_items;
                res_next = exhausted ? n_items : base;
                res_end = (base + TILE < n_items) ? base + TILE : n_items;
                if (exhausted) res_end = n_items;
                // the work-item numbers of the whole batch, one per lane (lane % TILE): the refills that take items of this
                // batch later read them from here instead of chaining a list load in front of their row loads
                res_base = base;
                const int bi = base + col;
                qbatch = (p.list && bi < n_items) ? p.list[bi] : bi;
                // patient indices of the batch, one division per lane and batch (a refill used to divide per item)
                const int qv = bi < n_items ? qbatch : 0;
                ibatch = p.row_begin + (qv / N) * p.row_step;
                jbatch = qv % N;
            }
            const int avail = res_end - res_next;
            const int n_want = (int)__popcll(wmask);
            const int rank = (int)__popcll(wmask & ((1ull << col) - 1ull));
            const int item = res_next + rank;
            const bool take = want && rank < avail;
            const int bsel = 4 * ((item - res_base) & (TILE - 1));
            const int qsel = __builtin_amdgcn_ds_bpermute(bsel, qbatch);     // (every lane takes part)
            const int isel = __builtin_amdgcn_ds_bpermute(bsel, ibatch), jsel = __builtin_amdgcn_ds_bpermute(bsel, jbatch);
            res_next = __builtin_amdgcn_readfirstlane(res_next + (n_want < avail ? n_want : avail));
            if (want && !take && exhausted) {   // no work left: the slot goes dark
                want = false;
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) { A[t][r] = B[t][r] = U[t][r] = V[t][r] = T(0); ACC[t][r] = T(1); }
                if constexpr (C::HALF) {
#pragma unroll
                    for (int part = 0; part < 2; ++part)
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int h = 0; h < 4; ++h) { PU.p[part][kb][h] = 0u; PV.p[part][kb][h] = 0u; }
                }
            }
            if (take) {
                want = false;
                active = true;
                q = qsel;
                const int i = isel, j = jsel;
                const T *pa = Pt + (size_t)i * KP + grp * NREG, *pb = Pt + (size_t)j * KP + grp * NREG;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    load_regs<C>(pa + t * NGRP * NREG, A[t]);          // 16-byte loads: a lane's slots are contiguous
                    load_regs<C>(pb + t * NGRP * NREG, B[t]);
                }
                thr = Pt[(size_t)N * KP + j];                 // stop threshold of column patient j (prep: f32 floor folded in; HALF: scaled copy)
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    load_regs<C>(acc0 + (t * NGRP + grp) * NREG, ACC[t]);
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        U[t][r] = (t == RT - 1) ? uinit - uinit * PADC[r] : uinit;   // 0 in padded slots; v follows from ACC
                        if constexpr (TRACK) { RU[t][r] = T(1); RV[t][r] = T(1); }
                    }
                }
                if constexpr (C::HALF) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            if (pieces_live(kb, h)) {
                                const unsigned int m = pad_mask(kb, h, PADC);
                                PU.p[0][kb][h] = pu0_hi & m; PU.p[1][kb][h] = pu0_lo & m;
                            }
                }
                chk = 1;
                ii = 0; flags = 0; abs_at = -1; errv = T(1);
            }
            if constexpr (C::HALF) {
                // Histograms of unequal mass (never PILOT's proportions; the ABI takes any P): u grows and v shrinks by the
                // mass ratio at every update, and long before tau is reached the shrinking panel has left the range in which
                // two fp16 pieces hold 22 bits (fuzz: 1.1e-5 / 2.4e-5 off at 40 capped updates).  The prep kernel flags such
                // a P, and this pass only forwards its pairs to the tracking kernel, which iterates f32 values.
                if (hand_all_over) {                                     // (wave-uniform)
                    const unsigned long long tm = ballot_b(take) & colmask;
                    if (tm) {
                        int base = 0;
                        if (lane == 0) base = __hip_atomic_fetch_add(p.track_count, (int)__popcll(tm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (take && grp == 0) p.track_list[base + __popcll(tm & ((1ull << col) - 1ull))] = q;
                    }
                    if (take) { active = false; want = true; }
                }
            }
        }
        if (ballot_b(active || want) == 0ull) break;

        T mx = T(0);
        if constexpr (C::HALF) {
            // ---- v = b / (G^T u), u = a / (G v): quotients straight into the pieces the products read ----
            // (the f32 quotients are temporaries; V is read again only by the marginal error of an update that ends with a
            // test -- small scaled values keep ABSOLUTE precision as pieces, which is all a product needs, but
            // d_k = v_k (G^T u)_k - b_k sees the relative one)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < NREG; ++r) V[t][r] = dead(t, r) ? T(0) : B[t][r] * M::rcp(ACC[t][r]);
            quot_panel(V, PV);
            // max(v) over the leading pieces, two elements per instruction; PV is dead once the product below is issued
            unsigned int m2 = 0u;
            {
                unsigned int pend = 0u;
                bool have = false;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        if (pieces_live(kb, h)) {
                            if (have) { m2 = pk_max3_f16(m2, pend, PV.p[0][kb][h]); have = false; }
                            else { pend = PV.p[0][kb][h]; have = true; }
                        }
                if (have) m2 = pk_max3_f16(m2, pend, pend);
            }
            product_h(a_g.img, PV, ACC);
            {
                acc_t X[RT];
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) X[t][r] = dead(t, r) ? T(0) : A[t][r] * M::rcp(ACC[t][r]);
                quot_panel(X, PU);
            }
            // ... joined by max(u) (a NaN piece makes the maximum NaN and the comparison below false: the pair is caught by the
            // error test instead, as in POT)
            {
                unsigned int pend = 0u;
                bool have = false;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        if (pieces_live(kb, h)) {
                            if (have) { m2 = pk_max3_f16(m2, pend, PU.p[0][kb][h]); have = false; }
                            else { pend = PU.p[0][kb][h]; have = true; }
                        }
                if (have) m2 = pk_max3_f16(m2, pend, pend);
            }
            const f16x2_t mh = __builtin_bit_cast(f16x2_t, m2);
            const float m0 = float(mh[0]), m1 = float(mh[1]);
            mx = (m0 <= tau && m1 <= tau) ? fmax(m0, m1) : __builtin_inff();      // a NaN half counts as over (see below)
        } else {
        // ---- v = b / (G^T u) --------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (dead(t, r)) continue;
                V[t][r] = B[t][r] * M::rcp(ACC[t][r]);
            }
        // ---- u = a / (G v) ----------------------------------------------------------------------------
        product(a_g, w_g, V, ACC, PADC);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (dead(t, r)) continue;
                const T un = A[t][r] * M::rcp(ACC[t][r]);
                U[t][r] = un;
                // scalings are positive, so max|.| needs no abs; NaN operands drop out of max (as in POT,
                // where `max(abs(u)) > tau` is False for a NaN) and are caught by the error test instead
                if constexpr (TRACK) mx = fmax(fmax(mx, un * RU[t][r]), V[t][r] * RV[t][r]);
                else mx = fmax(fmax(mx, un), V[t][r]);
            }
        }
        // POT: max|u| > tau or max|v| > tau  ->  absorb.  "any lane of the column over tau" == "column max over tau"
        // (fp16-split configuration: a scaling that jumps from below tau past the fp16 range in ONE update -- 531 -> 2071 at
        // K = 2 -- becomes the pieces (inf, -inf), the next product NaN, and a NaN maximum is not "> tau": such a pair used to
        // end at its next error test as "numerical errors" and take the POT-literal kernel, one workgroup per pair -- 65 of the
        // 67 ms of a 600 x 2 call.  There a NaN maximum counts as over (mx = inf, above): the tracking kernel, which iterates
        // f32 values, restarts the pair.)
        const unsigned long long omask = column_any_mask<C>(active && mx > tau);
        const bool over = (omask >> col) & 1ull;
        if constexpr (TRACK) {
            if (over) {
                // POT: alpha += reg log u, beta += reg log v, u = v = 1/K.  In total-scaling terms
                // (DESIGN.md): new references u_ref = u*K, v_ref = v/K; the stored u, v are unchanged.
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const bool pad = M::lidx(t, r, grp) >= K;
                        RU[t][r] = pad ? T(0) : M::rcp(U[t][r] * T(K));
                        RV[t][r] = pad ? T(0) : T(K) * M::rcp(V[t][r]);
                    }
                abs_at = ii;
                flags |= FLAG_ABSORBED;
            }
            if (omask) {    // (wave-uniform, rare)
                // An empty bin (a_k = 0 or b_k = 0, so u_k = 0 or v_k = 0 exactly) makes POT's absorption take log(0): the
                // potential becomes -inf, the rebuilt kernel row 0 and the next update 0/0 -- "Numerical errors", POT
                // returns the iterate before it.  The books kept here cannot express that; the pair is poisoned instead, ends
                // as NaN at its next test and is solved again by the POT-literal kernel (nan_list), which walks exactly
                // that path.  (The reference's proportions are strictly positive: Trajectory.py:405-430.)
                // The same exit for total scalings that leave the range in which these books are exact (2^+-100 in f32,
                // 2^+-900 in f64): histograms of unequal mass never converge and their scalings grow by the mass ratio at
                // every update -- POT carries that in its log-domain potentials, the fixed Gibbs image cannot.  Between two
                // absorptions a scaling moves by at most tau times one update's growth, so testing here is early enough.
                constexpr T BIG = sizeof(T) == 4 ? T(1.2676506e30) : T(8.452712498170644e270);     // 2^100, 2^900
                constexpr T SMALL = T(1) / BIG;
                bool z = false;
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int r = 0; r < NREG; ++r)
                        z = z || (M::lidx(t, r, grp) < K && !(U[t][r] >= SMALL && U[t][r] < BIG && V[t][r] >= SMALL && V[t][r] < BIG));
                const unsigned long long zmask = column_any_mask<C>(over && z);
                if ((zmask >> col) & 1ull) U[0][0] = __builtin_nanf("");
            }
        }
        // hand the pair to the tracking kernel (it restarts the pair from its first update)
        auto hand_over = [&]() {
            if (omask) {
                if (over) {
                    if (grp == 0) hb[hb_cnt + (int)__popcll(omask & ((1ull << col) - 1ull))] = q;
                    active = false;
                    want = true;
                }
                hb_cnt += (int)__popcll(omask);                         // (wave-uniform; <= HANDOVER_FLUSH - 1 + TILE <= HANDOVER_BUF)
                if (hb_cnt >= HANDOVER_FLUSH) hb_flush();
            }
        };
        // (fp16-split configuration: the hand-over -- a wave-uniform branch -- waits until the second product is issued, so that
        // quotients, both products and the tau maximum are ONE basic block and the scheduler may run the element-wise work of
        // the first tiles under the MFMAs of the later ones; the product of a column that is being handed over is wasted work)
        if constexpr (!TRACK && !(C::HALF && PILOT_DEFER_HANDOVER)) hand_over();
        ++ii;   // ii updates of (v, u) are done for this column

        // ---- ACC = G^T u: feeds the stopping test of this update and the next v ----------------------
        if constexpr (C::HALF) product_h(a_gt.img, PU, ACC);
        else product(a_gt, w_gt, U, ACC, PADC);
        if constexpr (!TRACK && C::HALF && PILOT_DEFER_HANDOVER) hand_over();

        // ---- POT's stopping rule: the error of update ii-1 is evaluated when (ii-1) % period == 0 ---
        const bool pending = active && ii == chk;
        if (pending) chk += p.period;
        const bool capped = active && ii >= p.max_iter;
        // squared marginal error of this lane's slots, per-tile partial sums added in tile order
        auto lane_e2 = [&](T sc) {
            T e2 = T(0);
            if constexpr (PILOT_E2_PACKED && sizeof(T) == 4 && !TRACK && NREG == 4) {
                // two slots per instruction (v_pk_fma_f32 on adjacent registers): d = v acc - b, e += d d -- 14 instead of 26 vector
                // instructions in a block that more than half the updates run; the two running sums are added at the end
                using P2 = pair_of<float>;
                P2 acc2 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r0 = 2 * h, r1 = 2 * h + 1;
                        if (dead(t, r0) && dead(t, r1)) continue;
                        if (dead(t, r0) || dead(t, r1)) {            // half a pair: the live slot alone
                            const int r = dead(t, r0) ? r1 : r0;
                            const float d = V[t][r] * ACC[t][r] - B[t][r];
                            acc2[0] = __builtin_fmaf(d, d, acc2[0]);
                            continue;
                        }
                        const P2 v2 = {V[t][r0], V[t][r1]}, a2 = {ACC[t][r0], ACC[t][r1]}, b2 = {B[t][r0], B[t][r1]};
                        const P2 d2 = __builtin_elementwise_fma(v2, a2, -b2);
                        acc2 = __builtin_elementwise_fma(d2, d2, acc2);
                    }
                return acc2[0] + acc2[1];
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                T et = T(0);
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    if (dead(t, r)) continue;      // (adds exactly 0 otherwise)
                    T d;
                    if constexpr (TRACK) d = V[t][r] * ACC[t][r] * sc - B[t][r];
                    else d = V[t][r] * ACC[t][r] - B[t][r];
                    et += d * d;
                }
                e2 += et;
            }
            return e2;
        };
        // (fast split kernels: the lane's part of the error is formed on EVERY update, in the block of the product that feeds it --
        // its ~26 vector instructions issue under the last MFMAs of that product, where the wave otherwise waits for the matrix
        // pipe -- instead of in the 56 % of the updates that test some column, where nothing covers them; same operations in
        // the same order: same bits)
        constexpr bool E2_EARLY = C::SPLIT && !TRACK && PILOT_E2_EARLY;
        T e2_early = T(0);
        if constexpr (E2_EARLY) e2_early = lane_e2(T(1));
        if (ballot_b(pending || capped)) {
            T sc = T(1);
            if constexpr (TRACK) sc = (abs_at == ii - 1) ? T(1) / kk : T(1);  // u, v were just reset to 1/K each
            T e2 = E2_EARLY ? e2_early : lane_e2(sc);
            e2 = group_sum<C>(e2);
            const T e = sqrt(e2);
            bool fin = capped;
            if (pending) {
                errv = e;
                if (e <= thr) { fin = true; flags |= FLAG_CONVERGED; }
                else if (e != e) { fin = true; flags |= FLAG_NAN; }   // POT: "Numerical errors at iteration"
            }
            // ---- retire finished pairs: (u, v) go to the wave's ring, the slot asks for the next pair ----
            unsigned long long fmask = ballot_b(fin) & colmask;
            if (fmask) {
                T scale = T(1);
                if constexpr (TRACK) {
                    if (fin && abs_at >= 0 && abs_at == ii - 1) { scale = T(1) / kk; flags |= FLAG_ABSORB_LAST; }
                }
                if constexpr (sizeof(T) == 8) flags |= FLAG_F64;
                if (fin && grp == 0) {
                    if (p.iters) p.iters[q] = ii;
                    if (p.err) p.err[q] = double(errv) * double(T(1) / IN_SCALE);
                }
                while (fmask) {     // wave-uniform: usually one pass; a second one when the ring fills up in between
                    if constexpr (PARK) {
                        // make room BEFORE finished columns are stored: the flush is inlined, U is parked in LDS meanwhile (with
                        // fewer ring slots than columns a second flush can follow in the next pass)
                        if (ring_cnt > 0 && ring_cnt + (int)__popcll(fmask) > p.ring) {      // wave-uniform
                            if constexpr (C::HALF) {
#pragma unroll
                                for (int part = 0; part < 2; ++part)
#pragma unroll
                                    for (int kb = 0; kb < KB; ++kb) *reinterpret_cast<u32x4_t *>(park + (part * KB + kb) * 4 * WAVE) = PU.p[part][kb];
                            } else {
#pragma unroll
                            for (int t = 0; t < RT; ++t) store_regs<C>(park + t * NREG * WAVE, U[t]);
                            }
                            ring_flush_body<C, RT>(ring, p, ring_cnt);
                            ring_cnt = 0;
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            const int i = p.row_begin + (q / N) * p.row_step, j = q % N;
                            const T *pa = Pt + (size_t)i * KP + grp * NREG, *pb = Pt + (size_t)j * KP + grp * NREG;
                            if constexpr (C::HALF) {
#pragma unroll
                                for (int part = 0; part < 2; ++part)
#pragma unroll
                                    for (int kb = 0; kb < KB; ++kb) PU.p[part][kb] = *reinterpret_cast<const u32x4_t *>(park + (part * KB + kb) * 4 * WAVE);
                            }
#pragma unroll
                            for (int t = 0; t < RT; ++t) {
                                if constexpr (!C::HALF) load_regs<C>(park + t * NREG * WAVE, U[t]);
#pragma unroll
                                for (int r = 0; r < NREG; ++r) { A[t][r] = T(0); B[t][r] = T(0); }
                                if (active) {
                                    load_regs<C>(pa + t * NGRP * NREG, A[t]);
                                    load_regs<C>(pb + t * NGRP * NREG, B[t]);
                                }
                            }
                            if constexpr (C::HALF) product_h(a_gt.img, PU, ACC);
                            else product(a_gt, w_gt, U, ACC, PADC);
                        }
                    }
                    const int space = p.ring - ring_cnt;
                    const int rank = (int)__popcll(fmask & ((1ull << col) - 1ull));
                    const bool put = fin && ((fmask >> col) & 1ull) && rank < space;
                    if (put) {
                        T *rec = ring + (ring_cnt + rank) * RSTRIDE;
                        constexpr int PE = ring_panel_elems<C>(RT);
                        if constexpr (C::HALF) {
#pragma unroll
                            for (int part = 0; part < 2; ++part)
#pragma unroll
                                for (int kb = 0; kb < KB; ++kb)
                                    *reinterpret_cast<u32x4_t *>(rec + ((part * KB + kb) * NGRP + grp) * 4) = PU.p[part][kb];
                            static_assert(!C::HALF || PE >= RT * C::TILE, "the v half of a ring slot holds f32 values");
#pragma unroll
                            for (int t = 0; t < RT; ++t) store_regs<C>(rec + PE + (t * NGRP + grp) * NREG, V[t]);   // (f32: see ring_flush_body)
                        } else {
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            store_regs<C>(rec + (t * NGRP + grp) * NREG, U[t]);
                            store_regs<C>(rec + KP + (t * NGRP + grp) * NREG, V[t]);
                        }
                        }
                        if (grp == 0) {
                            rec[2 * PE] = scale;
                            int *meta = reinterpret_cast<int *>(rec + 2 * PE + 1);
                            meta[0] = q;
                            meta[1] = flags;
                        }
                    }
                    const unsigned long long taken = ballot_b(put) & colmask;
                    fmask &= ~taken;
                    ring_cnt = __builtin_amdgcn_readfirstlane(ring_cnt + (int)__popcll(taken));
                    if constexpr (!PARK) {
                        if (ring_cnt >= p.ring) { flush(ring_cnt); ring_cnt = 0; }
                    }
                }
                if (fin) { active = false; want = true; }
            }
        }
    }
    if (ring_cnt > 0) {
        if constexpr (PARK) ring_flush_body<C, RT>(ring, p, ring_cnt);
        else flush(ring_cnt);
    }
    if constexpr (!TRACK) {
        if (hb_cnt > 0) hb_flush();
    }
}

// One-launch setup: Gibbs kernel images in MFMA operand order, first-product table, P converted to T.
__device__ inline unsigned short bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
}
__device__ inline unsigned short f16_bits(float x) {
    return __builtin_bit_cast(unsigned short, static_cast<_Float16>(x));
}
// split operand forms of configuration C at `img`: [part][k-block][out tile][lane][8 x 16 bit] pieces of G^T, G, G o M
// (bf16: exact 3-way split of the f32 value; fp16: 2 pieces of 2^15 times the value -- the rounding chains of split_pair)
template <class C>
__device__ inline void write_split_forms(const double *__restrict__ Msrc, int K, int RT, double reg, float *__restrict__ img,
                                         bool two_bands, int tid, int nthr) {
    using M = C;
    const int nimg = form_elems<C>(RT);
    const int KB = split_kblocks(RT);
    unsigned short *im16 = reinterpret_cast<unsigned short *>(img);
    for (int idx = tid; idx < KB * RT * WAVE * 8; idx += nthr) {
        const int e = idx % 8;
        int rest = idx / 8;
        const int lane = rest % WAVE; rest /= WAVE;
        const int t = rest % RT;
        const int kb = rest / RT;
        const int orow = M::lidx_of_row(t, lane % M::TILE);
        const int kt = 2 * kb + e / 4;                            // row-tile whose accumulator register e % 4 is this k-slot
        const int k = kt < RT ? M::lidx(kt, e % 4, lane / M::TILE) : K;
        float f[3] = {0.f, 0.f, 0.f}, f1[3] = {0.f, 0.f, 0.f};
        if (orow < K && k < K) {
            const double m_ko = Msrc[(size_t)k * K + orow], m_ok = Msrc[(size_t)orow * K + k];
            const double sc = C::HALF ? double(H_GIBBS_SCALE) : 1.0;
            f[0] = float(exp(-m_ko / reg) * sc);
            f[1] = float(exp(-m_ok / reg) * sc);
            f[2] = float(exp(-m_ok / reg) * m_ok * sc);
            if (two_bands) {       // two exponent bands: entries below the safe minimum move to band 1, times 2^128
                if (f[0] < SPLIT_SAFE_MIN) { f1[0] = float(exp(-m_ko / reg + BAND1_UP_LN)); f[0] = 0.f; }
                if (f[1] < SPLIT_SAFE_MIN) {
                    f1[1] = float(exp(-m_ok / reg + BAND1_UP_LN));
                    f1[2] = float(exp(-m_ok / reg + BAND1_UP_LN) * m_ok);
                    f[1] = 0.f; f[2] = 0.f;
                }
            }
        }
        if (two_bands) {
            unsigned short *im1 = reinterpret_cast<unsigned short *>(img + band1_offset<C>(RT));
#pragma unroll
            for (int form = 0; form < 3; ++form) {
                float x = f1[form];
#pragma unroll
                for (int part = 0; part < C::NP; ++part) {
                    const unsigned short hb = bf16_bits(x);
                    im1[(size_t)form * nimg * 2 + ((((size_t)part * KB + kb) * RT + t) * WAVE + lane) * 8 + e] = hb;
                    x -= __uint_as_float((unsigned int)hb << 16);
                }
            }
        }
#pragma unroll
        for (int form = 0; form < 3; ++form) {
            float x = f[form];
#pragma unroll
            for (int part = 0; part < C::NP; ++part) {
                unsigned short hb;
                if constexpr (C::HALF) { hb = f16_bits(x); x -= float(__builtin_bit_cast(_Float16, hb)); }
                else { hb = bf16_bits(x); x -= __uint_as_float((unsigned int)hb << 16); }
                im16[(size_t)form * nimg * 2 + ((((size_t)part * KB + kb) * RT + t) * WAVE + lane) * 8 + e] = hb;
            }
        }
    }
}
// first product of every pair: (G^T u0)[j] = (1/K) * sum_k G[k][j] (times `scale`), in accumulator-slot order
// (four adjacent lanes share a slot, each sums every fourth k: the K serial fp64 exponentials of a slot were the longest
// chain of the whole preparation)
template <class C>
__device__ inline void write_first_product(const double *__restrict__ Msrc, int K, int RT, double reg, typename C::T *__restrict__ tab,
                                           double scale, int tid, int nthr) {
    using M = C;
    using T = typename C::T;
    const int KP = RT * M::TILE;
    for (int idx = tid; idx < 4 * KP; idx += nthr) {              // (4 KP and nthr are multiples of the wave size)
        const int part = idx & 3, slot = idx >> 2;
        const int r = slot % M::NREG;                             // slot order [tile][group][reg]
        const int g = (slot / M::NREG) % M::NGRP;
        const int t = slot / (M::NGRP * M::NREG);
        const int j = M::lidx(t, r, g);
        double s = 0.0;
        if (j < K)
            for (int k = part; k < K; k += 4) s += exp(-Msrc[(size_t)k * K + j] / reg);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0) tab[slot] = T(j < K ? s / K * scale : 1.0);   // padded slots: 1 keeps b/acc = 0/1 finite
    }
}

template <class C>
__device__ inline void setup_body(const double *__restrict__ Msrc, int K, int RT, double reg,
                                  typename C::T *__restrict__ img, const double *__restrict__ Psrc,
                                  typename C::T *__restrict__ Pdst, long n_p, int write_tail, double stop_thr,
                                  double floor_ulps, int tid, int nthr, int *unequal = nullptr) {
    using M = C;
    using T = typename C::T;
    const int KP = RT * M::TILE;
    const int nimg = form_elems<C>(RT);
    if constexpr (C::SPLIT) {
        write_split_forms<C>(Msrc, K, RT, reg, img, (write_tail & 4) != 0, tid, nthr);
        if constexpr (C::HALF) {     // the bf16-split block of the tracking kernel
            write_split_forms<CfgS32x16>(Msrc, K, RT, reg, img + track_img_offset<C>(RT), false, tid, nthr);
            write_first_product<CfgS32x16>(Msrc, K, RT, reg, img + track_img_offset<C>(RT) + acc0_offset<CfgS32x16>(RT), 1.0, tid, nthr);
        }
    } else {
    for (int idx = tid; idx < nimg; idx += nthr) {
        const int lane = idx % WAVE;
        int rest = idx / WAVE;
        const int t = rest % RT; rest /= RT;
        const int r = rest % M::NREG;
        const int tp = rest / M::NREG;
        const int orow = M::lidx_of_row(t, lane % M::TILE);      // cell type of the output row of the product
        const int k = M::lidx(tp, r, lane / M::TILE);            // cell type of the contraction index
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
    }
    write_first_product<C>(Msrc, K, RT, reg, img + acc0_offset<C>(RT), C::HALF ? double(H_GIBBS_SCALE) * double(H_PANEL_SCALE) : 1.0, tid, nthr);
    // tail-row weights for the VALU variant (see tail_rows): [form][chain][k-step][lane] pairs behind the table
    if (write_tail & 2) {     // plain tables of solo_pairs: G[lane][k], G[k][lane], (G o M)[lane][k]
        T *plain = img + plain_offset<C>(RT);
        for (int idx = tid; idx < 3 * 64 * WAVE; idx += nthr) {
            const int lane = idx % WAVE, k = (idx / WAVE) % 64, tbl = idx / (64 * WAVE);
            double v = 0.0;
            if (k < K && lane < K) {
                if (tbl == 1) v = exp(-Msrc[(size_t)k * K + lane] / reg);
                else v = exp(-Msrc[(size_t)lane * K + k] / reg);
                if (tbl == 2) v *= Msrc[(size_t)lane * K + k];
            }
            plain[idx] = T(v);
        }
    }
    if (write_tail & 1) {
        const int nst = (RT - 1) * 4 + 1;
        T *tail = img + tail_offset<C>(RT);
        for (int idx = tid; idx < 2 * 2 * nst * WAVE * 2; idx += nthr) {
            const int h = idx & 1;
            int rest = idx >> 1;
            const int lane = rest % WAVE; rest /= WAVE;
            const int st = rest % nst; rest /= nst;
            const int c = rest & 1, form = rest >> 1;
            const int row = M::lidx(RT - 1, 0, 2 * c + h);
            const int k = M::lidx(st / M::NREG, st % M::NREG, lane / M::TILE);
            double v = 0.0;
            if (row < K && k < K) v = form == 0 ? exp(-Msrc[(size_t)k * K + row] / reg) : exp(-Msrc[(size_t)row * K + k] / reg);
            tail[idx] = T(v);
        }
    }
    // stop threshold per (column) patient behind the proportions: POT's stopThr, floored in f32 at floor_ulps * eps * ||b||_2
    // (the marginal error cannot get below the rounding of b itself)
    double mass0 = 0.0;
    for (int k = 0; k < K; ++k) mass0 += Psrc[k];
    for (long row = tid; row < n_p / KP; row += nthr) {
        double n2 = 0.0, mass = 0.0;
        for (int k = 0; k < K; ++k) { n2 += Psrc[row * K + k] * Psrc[row * K + k]; mass += Psrc[row * K + k]; }
        // histograms of unequal mass: the scalings of a pair drift apart by the mass ratio at every update (see the refill of
        // the stream kernel); 1e-6 relative is a drift of 0.1 % over 1000 updates
        if (unequal && !(fabs(mass - mass0) <= 1e-6 * fabs(mass0))) atomicOr(unequal, 1);
        double thr = stop_thr;
        if (sizeof(T) == 4) { const double fl = floor_ulps * 1.1920928955078125e-07 * sqrt(n2); thr = thr > fl ? thr : fl; }
        Pdst[n_p + row] = T(thr);
        if constexpr (C::HALF) Pdst[(n_p + n_p / KP) + n_p + row] = T(thr) * T(H_IN_SCALE);
    }
    // proportions: N rows of KP values in slot order, zero in padding (n_p = N * KP)
    for (long idx = tid; idx < n_p; idx += nthr) {
        const long row = idx / KP;
        const int sidx = (int)(idx % KP);
        const int r = sidx % M::NREG, g = (sidx / M::NREG) % M::NGRP, t = sidx / (M::NGRP * M::NREG);
        const int l = M::lidx(t, r, g);
        Pdst[idx] = l < K ? T(Psrc[row * K + l]) : T(0);
        // fp16-split configuration: a second copy in the scaled domain of its tile kernel (a 2^25, b 2^25, thresholds 2^25 -- powers of
        // two: the same bits the kernel used to form at every refill, 32 multiplies per lane in a block that 40 % of the updates
        // run) behind the first one (the buffer is sized for f64); the one-wave path and the tracking pass read the plain copy
        if constexpr (C::HALF) Pdst[(n_p + n_p / KP) + idx] = l < K ? T(Psrc[row * K + l]) * T(H_IN_SCALE) : T(0);
    }
}

// ---- longest-first work order ------------------------------------------------------------------------
// Sinkhorn needs more updates the closer the two histograms are (the diagonal pairs a == b are the slowest
// by far), and a pair's updates are a serial chain, so the long pairs must START first or they become the
// tail of the launch.  Pairs are bucketed by -log2 of their L1 distance (NB buckets, 4 per octave, exact
// duplicates in the last one) and the work list is emitted from the highest bucket down.  Three tiny
// launches: bucket ids + histogram, (prefix is folded into) scatter.
constexpr int ORDER_NB = 48;
// control block of a call (ints, zeroed per call by the host): [0 .. CTRL_INTS) counters and queue heads, then the order
// histograms.  Slot CTRL_UNEQUAL is set by the prep kernel when the rows of P do not all carry the same mass.
constexpr int CTRL_INTS = 16, CTRL_UNEQUAL = 12;

// wave-aggregated LDS counter: lanes with equal `b` share one atomic; returns the lane's slot (base + rank among equals)
__device__ inline int lds_count_aggregated(int *counters, int b, bool valid) {
    int slot = 0;
    unsigned long long todo = ballot_b(valid);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int b0 = __builtin_amdgcn_readlane(b, leader);
        const unsigned long long same = ballot_b(valid && b == b0);
        const int lane = threadIdx.x % WAVE;
        int base = 0;
        if (lane == leader) base = atomicAdd(&counters[b0], (int)__popcll(same));
        base = __builtin_amdgcn_readlane(base, leader);
        if (valid && b == b0) slot = base + (int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

// Tile of ORDER_RI selected rows x ORDER_JW columns per workgroup: the column histograms are staged transposed in LDS
// (coalesced global reads, conflict-free LDS reads), the row histograms are LDS broadcasts; 2 VALU per |a_k - b_k|.
constexpr int ORDER_JW = 128, ORDER_RI = 8;

__device__ inline void bucket_body(int tile, const double *__restrict__ Psrc, int N, int K, int n_rows, int row_begin,
                                   int row_step, unsigned char *__restrict__ bucket, int *__restrict__ hist, int collapse,
                                   unsigned char *order_smem) {
    // rows of the caller's N x K proportions (float is plenty for a sort key; L1 does not care about slot order)
    const int K4 = (K + 3) & ~3;
    float *Bt = reinterpret_cast<float *>(order_smem);           // [K4][ORDER_JW + 1]
    float *Ar = Bt + (size_t)K4 * (ORDER_JW + 1);                // [ORDER_RI][K4]
    int *lh = reinterpret_cast<int *>(Ar + ORDER_RI * K4);       // [ORDER_NB]
    const int n_jb = (N + ORDER_JW - 1) / ORDER_JW;
    const int jb = (tile % n_jb) * ORDER_JW, rb = (tile / n_jb) * ORDER_RI;
    const int nj = N - jb < ORDER_JW ? N - jb : ORDER_JW;
    const int nr = n_rows - rb < ORDER_RI ? n_rows - rb : ORDER_RI;
    for (int e = threadIdx.x; e < ORDER_NB; e += blockDim.x) lh[e] = 0;
    for (int e = threadIdx.x; e < nj * K; e += blockDim.x) {
        const int row = e / K, k = e % K;
        Bt[k * (ORDER_JW + 1) + row] = float(Psrc[(size_t)jb * K + e]);
    }
    for (int e = threadIdx.x; e < (K4 - K) * ORDER_JW; e += blockDim.x) Bt[(K + e / ORDER_JW) * (ORDER_JW + 1) + e % ORDER_JW] = 0.f;
    for (int e = threadIdx.x; e < ORDER_RI * K4; e += blockDim.x) {
        const int r = e / K4, k = e % K4;
        Ar[e] = (r < nr && k < K) ? float(Psrc[(size_t)(row_begin + (rb + r) * row_step) * K + k]) : 0.f;
    }
    __syncthreads();
    constexpr int R = ORDER_RI / 2;                               // rows per thread: the two halves of the block split the rows
    const int jl = threadIdx.x % ORDER_JW, half = threadIdx.x / ORDER_JW;
    float l1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) l1[r] = 0.f;
    using f4 = float __attribute__((ext_vector_type(4)));
    for (int k = 0; k < K4; k += 4) {
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = Bt[(k + e) * (ORDER_JW + 1) + jl];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f4 av = *reinterpret_cast<const f4 *>(Ar + (half * R + r) * K4 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) l1[r] += fabsf(av[e] - bv[e]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = rb + half * R + r;
        const bool valid = jl < nj && half * R + r < nr;
        int b = ORDER_NB - 1;
        if (l1[r] > 0.f) {
            const float v = 4.f * (1.f - log2f(l1[r]));      // l1 = 2 -> 0, halves add 4
            b = v < 0.f ? 0 : (v > float(ORDER_NB - 2) ? ORDER_NB - 2 : int(v));
            if (collapse) b = 0;       // experiment switch: natural order, only the duplicates are told apart
        } else if (row_begin + rr * row_step != jb + jl) {
            // a == b bit for bit but two DIFFERENT patients: stays in the tiles, at the head of their queue.  Only the diagonal
            // takes the one-wave path -- it is sized for one pair per row, and a cohort with a handful of cell types has
            // thousands of duplicate patients (K = 2, 200 cells each: 6 700 of 360 000 pairs, a quarter of them running to the
            // cap), which 16 to a wave in the tiles take a fraction of the time they queue for on solo waves
            // (profiles/r04/small_k_probe.txt).  A rule by VALUE (the two indices), like the old one: a row shard and the
            // full grid still send the same pair down the same path.
            b = ORDER_NB - 2;
        }
        if (valid) bucket[(size_t)rr * N + jb + jl] = (unsigned char)b;
        (void)lds_count_aggregated(lh, b, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ORDER_NB; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// One launch prepares a call: workgroups [0, n_tiles) compute the order keys (bucket_body), the rest build the operand
// images, tables and the slot-ordered proportions (setup_body).  The two halves are independent.
constexpr int PREP_SETUP_BLOCKS = 64;
template <class C>
__global__ void __launch_bounds__(256) sinkhorn_prep_kernel(const double *__restrict__ Msrc, int K, int RT, double reg,
                                                            typename C::T *__restrict__ img, const double *__restrict__ Psrc,
                                                            typename C::T *__restrict__ Pdst, int N, int write_tail, double stop_thr,
                                                            double floor_ulps, int n_tiles,
                                                            int n_rows, int row_begin, int row_step,
                                                            unsigned char *__restrict__ bucket, int *__restrict__ hist, int collapse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char order_smem[];
    if ((int)blockIdx.x < n_tiles) {
        bucket_body(blockIdx.x, Psrc, N, K, n_rows, row_begin, row_step, bucket, hist, collapse, order_smem);
    } else {
        setup_body<C>(Msrc, K, RT, reg, img, Psrc, Pdst, (long)N * RT * C::TILE, write_tail, stop_thr, floor_ulps,
                      ((int)blockIdx.x - n_tiles) * (int)blockDim.x + (int)threadIdx.x, PREP_SETUP_BLOCKS * (int)blockDim.x,
                      hist - (CTRL_INTS - CTRL_UNEQUAL));
    }
}

// list position of bucket b = (number of items in higher buckets) + a range reserved per workgroup.
// Every workgroup owns a contiguous chunk of items: LDS histogram of the chunk, ONE global atomic per
// (workgroup, bucket) to reserve the range, then LDS cursors -- a handful of hot global addresses would
// otherwise serialise all N^2 atomics.
// `split` (4 ints, written by workgroup 0): [0], [1] = n_dup, the number of leading list items that the solo waves take (the
// exact duplicates a == b: top bucket, first in the list) and the initial head of the tile queue; [2], [3] the same count for
// the solo waves' own queue.  solo_mode bit 1: the launch carries solo waves.
static __global__ void order_scatter_kernel(const unsigned char *__restrict__ bucket, int n_items, const int *__restrict__ hist,
                                     int *__restrict__ cursor, int *__restrict__ list, int *__restrict__ split,
                                     int *__restrict__ main_queue_head, int solo_mode) {
    __shared__ int offs[ORDER_NB], lh[ORDER_NB], lbase[ORDER_NB], gh[ORDER_NB];
    if (threadIdx.x < ORDER_NB) { lh[threadIdx.x] = 0; gh[threadIdx.x] = hist[threadIdx.x]; }
    __syncthreads();
    if (threadIdx.x < ORDER_NB) {          // items in higher buckets come first
        int run = 0;
        for (int b = ORDER_NB - 1; b > (int)threadIdx.x; --b) run += gh[b];
        offs[threadIdx.x] = run;
    }
    if (blockIdx.x == 0 && threadIdx.x == 64) {
        const int n_dup = (solo_mode & 2) ? gh[ORDER_NB - 1] : 0;
        split[0] = n_dup; split[1] = n_dup; split[2] = n_dup; split[3] = n_dup;
        *main_queue_head = n_dup;          // the tile queue starts behind the duplicates
    }
    __syncthreads();
    const int chunk = (n_items + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * chunk, q1 = (q0 + chunk < n_items) ? q0 + chunk : n_items;
    const int n_it = (q1 - q0 + (int)blockDim.x - 1) / (int)blockDim.x;      // whole waves stay in the loop: ballots inside
    for (int it = 0; it < n_it; ++it) {
        const int q = q0 + it * blockDim.x + threadIdx.x;
        const bool valid = q < q1;
        (void)lds_count_aggregated(lh, valid ? bucket[q] : 0, valid);
    }
    __syncthreads();
    if (threadIdx.x < ORDER_NB) {
        const int c = lh[threadIdx.x];
        lbase[threadIdx.x] = offs[threadIdx.x] + (c ? atomicAdd(&cursor[threadIdx.x], c) : 0);
        lh[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int q = q0 + it * blockDim.x + threadIdx.x;
        const bool valid = q < q1;
        const int b = valid ? bucket[q] : 0;
        const int slot = lds_count_aggregated(lh, b, valid);
        if (valid) list[lbase[b] + slot] = q;
    }
}

}  // namespace pilot
