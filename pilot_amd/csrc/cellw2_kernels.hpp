// Cell-level W2 pair grid (BASELINE config 5).  NOT in the reference: PILOT only ever solves
// K x K problems on cell-type proportions; this extension compares two patients by their raw cell clouds.
//
// Pair (p, q): a = 1/n_p, b = 1/n_q, C_ij = |x_i - y_j|^2 / scale, entropic OT in the log domain with the control
// flow of POT's ot.bregman.sinkhorn_log (v-update, then u-update, marginal error every `period` updates, strict <):
//     v = log b - LSE_i(-C_ij/eps + u_i);   u = log a - LSE_j(-C_ij/eps + v_j);   value = sum(exp(logT) * C)
// With alpha = 1/(scale*eps) and the shifted potentials hu_i = u_i - alpha|x_i|^2, hv_j = v_j - alpha|y_j|^2 both
// updates are the same "flash" pass   out_r = LSE_c( 2 alpha <A_r, B_c> + h_c ):   hv = log b - out(A=Y, B=X, h=hu),
// hu = log a - out(A=X, B=Y, h=hv).  The n_p x n_q cost matrix (100 MB at 5000 cells) is never materialised: 16 x 16
// tiles of dot products are recomputed every half-update on the matrix pipe, the log-sum-exp is kept online per lane
// (running max + rescaled sum, base 2 so v_exp_f32 / v_log_f32 are used directly) and reduced across the 16 lanes of a
// row with DPP at the end of a row block.  One workgroup (16 waves) per pair; the potentials of the pair live in LDS.
//
// Dot products: the exponent 2 alpha <x, y> + h needs f32-level absolute accuracy, but both operands are STATIC (the
// cells), so each coordinate is split ONCE by the setup kernel into three bf16 pieces (x = x1 + x2 + x3 exactly) and a
// 16 x 16 x 32 tile is six v_mfma_f32_16x16x32_bf16 (x1y1 + x1y2 + x2y1 + x2y2 + x1y3 + x3y1, f32 accumulation: terms of
// order 2^-24 dropped) = 96 matrix-pipe cycles instead of the 8 x 32 = 256 of the f32-input MFMA it replaces -- and the
// bf16 pipe leaves the vector unit free for the exponentials.
//
// HALF (default when the scaled coordinates fit fp16's range): TWO fp16 pieces per coordinate (11 + 11 significant bits) and
// three v_mfma_f32_16x16x32_f16 per tile (x1y1 + x1y2 + x2y1) -- half the matrix work; the dropped x2y2 and the 2^-22
// relative representation error of the static operands perturb the exponent by <= 3 * 2^-22 |x||y| 2 alpha log2 e, i.e. the
// COST by <= ~7e-7 |x||y| / scale (a fixed perturbation of the ground cost: the OT value moves by at most that).  The
// column potential h_col cannot be rounded to 22 bits and is added on the vector unit; the row shift m_row still rides a
// spare k-slot, rounded to what its two fp16 pieces represent -- the SAME rounded value is added back after the log, so
// the rounding cancels exactly.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {

constexpr float CELL_NEG_BIG = -1.0e30f;
// The pipelined fp16 column sweep requests tile tb + 1 (16 cells) unclamped while it works on tile tb, so after a partial last
// tile it reads up to 2 * 16 - 1 = 31 cells past a patient's last one: the operand buffer ends in a zeroed pad of
// CELL_PREFETCH_CELLS cells of 32 bytes per plane (a weight of 2^(finite - big) = 0 makes whatever is read there harmless,
// but the read itself must stay inside the allocation).  Host allocation and kernel share this constant.
constexpr int CELL_TILE_CELLS = 16, CELL_PREFETCH_TILES = 1;
constexpr int CELL_PREFETCH_CELLS = (CELL_PREFETCH_TILES + 1) * CELL_TILE_CELLS;
constexpr size_t CELL_PAD_BYTES = (size_t)CELL_PREFETCH_CELLS * 32;
static_assert(CELL_PAD_BYTES >= (size_t)((CELL_PREFETCH_TILES + 1) * CELL_TILE_CELLS - 1) * 32, "operand pad shorter than the sweep's read-ahead");
constexpr int CELL_WG = 1024;   // 16 waves share one pair: the self-pairs converge slowly and set the critical path
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;

// the operand pieces of the cohort (see CellParams::Xb)
struct CellXb {
    const uint4 *p;
    long long C;
    int np;                 // pieces per coordinate: 3 (bf16) or 2 (fp16)
    __device__ inline const uint4 &at(long point, int kb, int piece, int g) const {
        return p[((((long)kb * np + piece) * 2 + (g >> 1)) * C + point) * 2 + (g & 1)];
    }
};

struct CellParams {
    const uint4 *Xb;        // bf16 operand pieces, 16 bytes = the 8 k-slots 32 kb + 8 g .. + 7 of one cell and piece, laid out
                            // [kb][piece][g >> 1][cell][g & 1] (CellXb::at): the lanes of one load instruction cover consecutive
                            // cells and so consecutive, fully used cache lines
    long long C;            // cells in the cohort
    const float *nrm;       // C: |x_c|^2
    const long long *offs;  // N + 1: first cell of every patient
    int N;
    int n_rows, row_begin, row_step;
    float two_alpha2;       // 2 * alpha * log2(e): the operand pieces are scaled by its square root, so a dot product of
                            // two operands IS the exponent term 2 alpha log2(e) <x, y>
    float dot_unscale;      // 1 / two_alpha2: back to <x, y>
    float alpha;            // 1 / (scale * eps)
    float inv_scale;        // 1 / scale
    int max_iter, period;
    float stop_thr, floor_ulps;
    int max_n;              // largest patient (LDS layout: hu[max_n], hv[max_n], hvn[max_n])
    double *w2;             // n_rows x N
    int *iters;
    double *err;
    int *queue;             // dynamic pair queue
};

// one pre-pass over the cells: the three bf16 pieces of every (scaled) coordinate in MFMA operand order + squared norms
// (half: two fp16 pieces instead of three bf16 ones)
// one_slot (fp16 pieces with spare k-slots): the last k-slot of every cell holds 1 -- the B side of the row-shift slot of
// cell_patch_a, so the column sweep patches nothing (the A side overwrites or clears it: cell_clear_slot)
__global__ void cell_setup_kernel(const float *__restrict__ X, long C, int D, int KB, float op_scale, int half, int one_slot,
                                  unsigned short *__restrict__ Xb, float *__restrict__ nrm) {
    const int np = half ? 2 : 3;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < C * KB * 32; idx += (long)gridDim.x * blockDim.x) {
        const long c = idx / (KB * 32);
        const int o = (int)(idx % (KB * 32)), kb = o / 32, d = o;      // coordinate d sits in k-block d / 32, slot d % 32
        float x = d < D ? X[c * D + d] * op_scale : 0.f;
        if (one_slot && o == KB * 32 - 1) x = 1.f;
        for (int piece = 0; piece < np; ++piece) {
            unsigned short hb;
            if (half) {
                const _Float16 h = static_cast<_Float16>(x);
                hb = __builtin_bit_cast(unsigned short, h);
                x -= static_cast<float>(h);
            } else {
                hb = __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
                x -= __uint_as_float((unsigned int)hb << 16);
            }
            const int g = (o % 32) / 8;
            Xb[((((long)kb * np + piece) * 2 + (g >> 1)) * C + c) * 16 + (g & 1) * 8 + (o % 8)] = hb;
        }
    }
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < C; c += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += X[c * D + d] * X[c * D + d];
        nrm[c] = s;
    }
}

using cell_bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using cell_f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using cell_f4 = float __attribute__((ext_vector_type(4)));
// the lane's operand pieces of one point: p[kb][piece] = 8 x 16 bit (k-slots 8 g .. 8 g + 7 of k-block kb); NP = 3 bf16 / 2 fp16
template <int KB, int NP = 3> struct CellOperand { uint4 p[KB][NP]; };
template <int KB, int NP> __device__ inline void cell_load(const CellXb &Xb, long point, int g, CellOperand<KB, NP> &o) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) o.p[kb][piece] = Xb.at(point, kb, piece, g);
}
// fp16 pieces of a value (hi + lo) and what they add up to
__device__ inline void cell_split2h(float x, unsigned int &hi, unsigned int &lo) {
    const _Float16 h = static_cast<_Float16>(x);
    const _Float16 l = static_cast<_Float16>(x - static_cast<float>(h));
    hi = __builtin_bit_cast(unsigned short, h); lo = __builtin_bit_cast(unsigned short, l);
}
__device__ inline float cell_round2h(float x) {
    const _Float16 h = static_cast<_Float16>(x);
    const _Float16 l = static_cast<_Float16>(x - static_cast<float>(h));
    return static_cast<float>(h) + static_cast<float>(l);
}
// exact 3-way split of an f32 into bf16 pieces by truncation (8 + 8 + 8 mantissa bits): x = hi + mid + lo
__device__ inline void cell_split3(float x, unsigned int &hi, unsigned int &mid, unsigned int &lo) {
    const unsigned int xh = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(xh);
    const unsigned int xm = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(xm);
    hi = xh >> 16; mid = xm >> 16; lo = __float_as_uint(r2) >> 16;
}
// The two spare k-slots of the last k-block (D <= 32 KB - 2) carry the exponent's additive terms through the MFMA:
// slot 32 KB - 2 holds (1 on the A side, h_col on the B side), slot 32 KB - 1 holds (-m_row, 1), each value as its three
// bf16 pieces -- with the 1 only in the leading piece, the six piece products of cell_dot_tile add exactly h_col - m_row,
// and the tile comes out of the matrix pipe as the finished exponent.  Both slots sit in the .w word of the g = 3 lanes.
template <int KB> __device__ inline void cell_patch_a(CellOperand<KB, 3> &a, int g, float minus_m) {
    unsigned int h, m, l;
    cell_split3(minus_m, h, m, l);
    if (g == 3) {
        a.p[KB - 1][0].w = 0x3f80u | (h << 16);
        a.p[KB - 1][1].w = m << 16;
        a.p[KB - 1][2].w = l << 16;
    }
}
template <int KB> __device__ inline void cell_patch_b(CellOperand<KB, 3> &b, int g, float hcol) {
    unsigned int h, m, l;
    cell_split3(hcol, h, m, l);
    if (g == 3) {
        b.p[KB - 1][0].w = h | (0x3f80u << 16);
        b.p[KB - 1][1].w = m;
        b.p[KB - 1][2].w = l;
    }
}
// fp16 pieces: only the row shift rides the last spare slot (A side: the two pieces of -m_row, B side: 1); the result tile is
// <a, b> - cell_round2h(m_row).  The other spare slot stays zero on both sides.
template <int KB> __device__ inline void cell_patch_a(CellOperand<KB, 2> &a, int g, float minus_m) {
    unsigned int h, l;
    cell_split2h(minus_m, h, l);
    if (g == 3) {
        a.p[KB - 1][0].w = (a.p[KB - 1][0].w & 0xffffu) | (h << 16);
        a.p[KB - 1][1].w = (a.p[KB - 1][1].w & 0xffffu) | (l << 16);
    }
}
// (B side: the 1 of that slot is part of the stored operand, cell_setup_kernel's one_slot)
// a plain A operand: the stored 1 of the last slot is taken out again (products with unpatched operands on both sides)
template <int KB> __device__ inline void cell_clear_slot(CellOperand<KB, 2> &a, int g) {
    if (g == 3) a.p[KB - 1][0].w &= 0xffffu;
}
template <int KB> __device__ inline void cell_clear_slot(CellOperand<KB, 3> &, int) {}
// 16 x 16 tile of <A_row, B_col>: six bf16 MFMAs per k-block, smallest terms first
template <int KB> __device__ inline cell_f4 cell_dot_tile(const CellOperand<KB, 2> &a, const CellOperand<KB, 2> &b) {
    cell_f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto mm = [](const uint4 &x, const uint4 &y, cell_f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cell_f16x8, x), __builtin_bit_cast(cell_f16x8, y), c, 0, 0, 0);
    };
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        acc = mm(a.p[kb][1], b.p[kb][0], acc);
        acc = mm(a.p[kb][0], b.p[kb][1], acc);
        acc = mm(a.p[kb][0], b.p[kb][0], acc);
    }
    return acc;
}
template <int KB> __device__ inline cell_f4 cell_dot_tile(const CellOperand<KB, 3> &a, const CellOperand<KB, 3> &b) {
    cell_f4 acc = {0.f, 0.f, 0.f, 0.f};

    auto mm = [](const uint4 &x, const uint4 &y, cell_f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cell_bf16x8, x), __builtin_bit_cast(cell_bf16x8, y), c, 0, 0, 0);
    };
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        acc = mm(a.p[kb][2], b.p[kb][0], acc);
        acc = mm(a.p[kb][0], b.p[kb][2], acc);
        acc = mm(a.p[kb][1], b.p[kb][1], acc);
        acc = mm(a.p[kb][1], b.p[kb][0], acc);
        acc = mm(a.p[kb][0], b.p[kb][1], acc);
        acc = mm(a.p[kb][0], b.p[kb][0], acc);
    }
    return acc;
}

template <int CTRL> __device__ inline float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// combine the online (max, sum) pairs of the 16 lanes of a row group; every lane of the group gets the result
__device__ inline void row16_lse_combine(float &m, float &l) {
#define PILOT_LSE_STEP(CTRL)                                                       \
    {                                                                              \
        const float mo = dpp_f32<CTRL>(m), lo = dpp_f32<CTRL>(l);                  \
        const float mn = fmaxf(m, mo);                                             \
        l = l * __builtin_amdgcn_exp2f(m - mn) + lo * __builtin_amdgcn_exp2f(mo - mn); \
        m = mn;                                                                    \
    }
    PILOT_LSE_STEP(0xB1)    // lane ^ 1
    PILOT_LSE_STEP(0x4E)    // lane ^ 2
    PILOT_LSE_STEP(0x141)   // the other quad of the octet
    PILOT_LSE_STEP(0x140)   // the other octet of the row
#undef PILOT_LSE_STEP
}
__device__ inline float row16_sum(float x) {
    x += dpp_f32<0xB1>(x); x += dpp_f32<0x4E>(x); x += dpp_f32<0x141>(x); x += dpp_f32<0x140>(x);
    return x;
}

// out2[r] = log2 sum_c 2^( two_alpha2 * <A_r, B_c> + h2[c] )  for every row r of the A side  (base-2 LSE)
// FN(row, lse2) is called by one lane per row with the result.
//
// The exponent shift of a row is the value this same LSE had one update ago (ref_r = logw2 - hprev[r]: the potential the
// previous update wrote IS log w - LSE): between two Sinkhorn updates it moves by a fraction of a unit, so the sum of
// 2^(t - ref_r) is of order one and neither a running maximum nor its rescaling exponentials are needed -- per element
// fma, sub, v_exp, add instead of fma, max, sub, v_exp, add plus one v_exp per row and step (the pass is bound by the
// vector unit, not by the six MFMAs per tile).  use_ref = false (first update of a pair: no previous value), or a block
// whose shifted sums leave [2^-64, 2^64] for any row (never seen in practice), takes the online-maximum form.
template <int KB, bool AUG, int NP, class FN>
__device__ inline void lse_pass(const CellXb &Xb, long a0, int na, long b0, int nb,
                                const float *h2 /* LDS, nb */, const float *hprev /* LDS, na */, float logw2,
                                bool use_ref, int wave, int n_waves, int lane, FN &&fn) {
    using f4 = cell_f4;
    const int col = lane & 15, g = lane >> 4;
    // a wave keeps the A operands of RB row blocks in registers and sweeps the B side once for all of them: every B tile
    // it loads (3 x 1 KB per k-block, from L1/L2) feeds RB output tiles -- the sweep is bound by the vector-memory path,
    // not by the MFMAs
#ifndef CELL_RB
#define CELL_RB 4
#endif
    constexpr int RB = KB >= 2 ? 2 : CELL_RB, TB = KB >= 2 ? 1 : (CELL_RB >= 4 ? 1 : 2);
    constexpr int TBS = KB >= 2 ? 2 : 4;                  // column tiles per step of the online-maximum form
    for (int unit = wave; unit * 16 * RB < na; unit += n_waves) {
        CellOperand<KB, NP> a[RB];
        float m[RB][4], l[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            int arow = (unit * RB + rb) * 16 + col;       // A operand: lane holds row (lane & 15), k-slots of group g
            if (arow >= na) arow = na - 1;
            cell_load<KB, NP>(Xb, a0 + arow, g, a[rb]);
            if constexpr (AUG && NP == 2) { if (!use_ref) cell_clear_slot<KB>(a[rb], g); }
        }
        bool done = false;
        if (use_ref) {                                    // (wave-uniform)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (unit * RB + rb) * 16 + 4 * g + r;
                    m[rb][r] = logw2 - hprev[row < na ? row : na - 1];
                    if constexpr (AUG && NP == 2) m[rb][r] = -cell_round2h(-m[rb][r]);      // what the fp16 slot subtracts
                    l[rb][r] = 0.f;
                }
            if constexpr (AUG) {                          // the row shift goes into the spare slot of the A operands
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    int arow = (unit * RB + rb) * 16 + col;
                    if (arow >= na) arow = na - 1;
                    cell_patch_a<KB>(a[rb], g, hprev[arow] - logw2);
                }
            }
            if constexpr (AUG && NP == 2) {
                // fp16 pieces: the B side needs no patch (its slot value is stored) and no clamp -- columns beyond the patient
                // weigh 2^(finite + CELL_NEG_BIG) = 0 whatever the operand (the next patient's cells, or the zeroed pad behind
                // the last plane), so the address is the lane's base plus 512 bytes per tile.  The tile of step tb + 1 is
                // requested before the MFMAs of step tb; all MFMAs of a step come first, each output tile into registers of
                // its own (one shared accumulator made every tile's vector work wait for its MFMAs: `s_nop 7` four times per
                // step), then the vector work with the adds two at a time (v_pk_add_f32 is full rate; exponentials are not).
                using f2 = float __attribute__((ext_vector_type(2)));
                CellOperand<KB, NP> bn;
                cell_load<KB, NP>(Xb, b0 + col, g, bn);
                for (int tb = 0; tb * 16 < nb; ++tb) {
                    const CellOperand<KB, NP> bc = bn;
                    cell_load<KB, NP>(Xb, b0 + (tb + 1) * 16 + col, g, bn);
                    const int bcol = tb * 16 + col;
                    const float hv = bcol < nb ? h2[bcol] : CELL_NEG_BIG;
                    f4 acc[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb] = cell_dot_tile<KB>(a[rb], bc);     // 2 alpha log2e <A, B> - m_row (rounded m)
                    const f2 hh = {hv, hv};
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const f2 x01 = f2{acc[rb][0], acc[rb][1]} + hh, x23 = f2{acc[rb][2], acc[rb][3]} + hh;
                        const f2 e01 = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1])};
                        const f2 e23 = {__builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
                        const f2 l01 = f2{l[rb][0], l[rb][1]} + e01, l23 = f2{l[rb][2], l[rb][3]} + e23;
                        l[rb][0] = l01[0]; l[rb][1] = l01[1]; l[rb][2] = l23[0]; l[rb][3] = l23[1];
                    }
                }
            } else {
            for (int tb = 0; tb * 16 < nb; tb += TB) {
                CellOperand<KB, NP> b[TB];
                float h[TB];
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    int bcol = (tb + u) * 16 + col;
                    const bool okc = bcol < nb;
                    if (!okc) bcol = nb - 1;
                    cell_load<KB, NP>(Xb, b0 + bcol, g, b[u]);
                    h[u] = okc ? h2[bcol] : CELL_NEG_BIG;
                    if constexpr (AUG) cell_patch_b<KB>(b[u], g, h[u]);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int u = 0; u < TB; ++u) {
                        if constexpr (AUG) {
                            const f4 acc = cell_dot_tile<KB>(a[rb], b[u]);     // acc[r] = the exponent of (row 4g+r, column u)
#pragma unroll
                            for (int r = 0; r < 4; ++r) l[rb][r] += __builtin_amdgcn_exp2f(acc[r]);
                        } else {
                            const f4 acc = cell_dot_tile<KB>(a[rb], b[u]);     // acc[r] = 2 alpha log2e <A_{4g+r}, B_col(u)>
#pragma unroll
                            for (int r = 0; r < 4; ++r) l[rb][r] += __builtin_amdgcn_exp2f(acc[r] + h[u] - m[rb][r]);
                        }
                    }
            }
            }
            bool bad = false;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l[rb][r] = row16_sum(l[rb][r]);
                    bad = bad || !(l[rb][r] > 5.4e-20f && l[rb][r] < 1.8e19f);
                }
            done = __ballot(bad) == 0ull;
        }
        if (!done) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if ((unit * RB + rb) * 16 >= na) continue;            // (wave-uniform)
                if (AUG && use_ref) {                                 // the plain operand again
                    int arow = (unit * RB + rb) * 16 + col;
                    if (arow >= na) arow = na - 1;
                    cell_load<KB, NP>(Xb, a0 + arow, g, a[rb]);
                    if constexpr (AUG && NP == 2) cell_clear_slot<KB>(a[rb], g);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { m[rb][r] = CELL_NEG_BIG; l[rb][r] = 0.f; }
                // TBS column tiles per step: one running-max rescale per row and step instead of one per element
                for (int tb = 0; tb * 16 < nb; tb += TBS) {
                    f4 acc[TBS];
                    float h[TBS];
#pragma unroll
                    for (int u = 0; u < TBS; ++u) {
                        int bcol = (tb + u) * 16 + col;
                        const bool okc = bcol < nb;
                        if (!okc) bcol = nb - 1;
                        CellOperand<KB, NP> b;
                        cell_load<KB, NP>(Xb, b0 + bcol, g, b);
                        h[u] = okc ? h2[bcol] : CELL_NEG_BIG;
                        acc[u] = cell_dot_tile<KB>(a[rb], b);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t[TBS], mn = m[rb][r];
#pragma unroll
                        for (int u = 0; u < TBS; ++u) { t[u] = acc[u][r] + h[u]; mn = fmaxf(mn, t[u]); }
                        float add = 0.f;
#pragma unroll
                        for (int u = 0; u < TBS; ++u) add += __builtin_amdgcn_exp2f(t[u] - mn);
                        l[rb][r] = fmaf(l[rb][r], __builtin_amdgcn_exp2f(m[rb][r] - mn), add);
                        m[rb][r] = mn;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) row16_lse_combine(m[rb][r], l[rb][r]);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (unit * RB + rb) * 16 + 4 * g + r;
                if (col == 0 && row < na) fn(row, m[rb][r] + __builtin_amdgcn_logf(l[rb][r]));   // v_log_f32 is log2
            }
    }
}

// sum_ij 2^(two_alpha2 <x_i, y_j> + hu2_i + hv2_j) * C_ij  for the rows handled by this wave (lane-local partial)
template <int KB, int NP, bool ONE_SLOT = false>
__device__ inline float value_pass(const CellXb &Xb, long a0, const float *__restrict__ na2, int na,
                                   long b0, const float *__restrict__ nb2, int nb,
                                   const float *hA2, const float *hB2, float dot_unscale, float inv_scale, int wave,
                                   int n_waves, int lane) {
    using f4 = cell_f4;
    const int col = lane & 15, g = lane >> 4;
    float total = 0.f;
    for (int blk = wave; blk * 16 < na; blk += n_waves) {
        int arow = blk * 16 + col;
        if (arow >= na) arow = na - 1;
        CellOperand<KB, NP> a;
        cell_load<KB, NP>(Xb, a0 + arow, g, a);
        if constexpr (ONE_SLOT) cell_clear_slot<KB>(a, g);
        float hr[4], nr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = blk * 16 + 4 * g + r;
            const bool ok = row < na;
            hr[r] = ok ? hA2[row] : CELL_NEG_BIG;
            nr[r] = ok ? na2[row] : 0.f;
        }
        for (int tb = 0; tb * 16 < nb; ++tb) {
            int bcol = tb * 16 + col;
            const bool okc = bcol < nb;
            if (!okc) bcol = nb - 1;
            CellOperand<KB, NP> b;
            cell_load<KB, NP>(Xb, b0 + bcol, g, b);
            const float h = okc ? hB2[bcol] : CELL_NEG_BIG;
            const float nc = nb2[bcol];
            const f4 acc = cell_dot_tile<KB>(a, b);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wgt = __builtin_amdgcn_exp2f(acc[r] + h + hr[r]);                    // Gamma_ij
                const float c = fmaxf(fmaf(-2.f * dot_unscale, acc[r], nr[r] + nc), 0.f) * inv_scale;   // C_ij >= 0
                total = fmaf(wgt, c, total);
            }
        }
    }
    return total;
}

// AUG: the embedding leaves two spare k-slots (D <= 32 KB - 2), see cell_patch_a / cell_patch_b
// HALF: two fp16 operand pieces (three piece products per tile) instead of three bf16 ones (six)
template <int KB, bool AUG, bool HALF = false>
__global__ void __launch_bounds__(CELL_WG) cell_w2_kernel(CellParams p) {
    constexpr int NP = HALF ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *hu = reinterpret_cast<float *>(smem_raw);        // [max_n] shifted potentials of the row patient (base 2)
    float *hv = hu + p.max_n;                                // [max_n] ... of the column patient
    float *hvn = hv + p.max_n;                               // [max_n] spare buffer of the v-update
    float *red = hvn + p.max_n;                              // [32] per-wave partial sums
    int *qslot = reinterpret_cast<int *>(red + 32);
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64, n_waves = blockDim.x / 64;
    const long total = (long)p.n_rows * p.N;
    const CellXb Xb = {p.Xb, p.C, NP};

    for (;;) {
        if (threadIdx.x == 0) qslot[0] = atomicAdd(p.queue, 1);
        __syncthreads();
        const long q = qslot[0];
        __syncthreads();
        if (q >= total) break;
        // self-pairs first (slowest to converge), then the rest row by row
        int r_idx, j;
        if (q < p.n_rows) { r_idx = (int)q; j = p.row_begin + r_idx * p.row_step; }
        else {
            const long qq = q - p.n_rows;
            r_idx = (int)(qq / (p.N - 1));
            j = (int)(qq % (p.N - 1));
            j += j >= p.row_begin + r_idx * p.row_step;
        }
        const int i = p.row_begin + r_idx * p.row_step;
        const long out = (long)r_idx * p.N + j;
        const long o_p = p.offs[i], o_q = p.offs[j];
        const int np = (int)(p.offs[i + 1] - o_p), nq = (int)(p.offs[j + 1] - o_q);
        const float *nxp = p.nrm + o_p, *nyq = p.nrm + o_q;
        const float loga2 = -__builtin_amdgcn_logf((float)np), logb2 = -__builtin_amdgcn_logf((float)nq);
        const float bval = 1.f / (float)nq;
        const float a2l = p.alpha * LOG2E_F;
        // u = v = 0  ->  hu2_i = -alpha |x_i|^2 log2 e
        for (int t = threadIdx.x; t < np; t += blockDim.x) hu[t] = -a2l * nxp[t];
        for (int t = threadIdx.x; t < nq; t += blockDim.x) hv[t] = -a2l * nyq[t];
        __syncthreads();
        // f32 floor of the stop threshold (|b|_2 = 1/sqrt(nq)), as in the proportion-level kernel
        float thr = p.stop_thr;
        {
            const float fl = p.floor_ulps * 1.1920929e-07f * __builtin_amdgcn_rsqf((float)nq);
            thr = thr > fl ? thr : fl;
        }
        int iters = 0;
        float err = 1.f;
        float *hv_cur = hv, *hv_new = hvn;
        for (int ii = 0; ii < p.max_iter; ++ii) {
            // ---- v-update:  hv_j = log b - LSE_i(2 alpha <y_j, x_i> + hu_i), written to the spare buffer.  The column
            //      sums of the plan BEFORE this update are 2^(hv_old_j + out_j), so the marginal error POT evaluates
            //      after update ii-1 comes for free; if it stops the pair, (hu, hv_old) is exactly the plan POT returns.
            const bool check = ii > 0 && ((ii - 1) % p.period == 0);
            float e2 = 0.f;
            lse_pass<KB, AUG, NP>(Xb, o_q, nq, o_p, np, hu, hv_cur, logb2, ii > 0, wave, n_waves, lane, [&](int row, float lse2) {
                if (check) {
                    const float d = __builtin_amdgcn_exp2f(hv_cur[row] + lse2) - bval;
                    e2 = fmaf(d, d, e2);
                }
                hv_new[row] = logb2 - lse2;
            });
            if (check) {   // wave-uniform
                float s = e2;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0) red[wave] = s;
            }
            __syncthreads();
            if (check) {
                float e = 0.f;
                for (int w = 0; w < n_waves; ++w) e += red[w];   // fixed order: results do not depend on scheduling
                err = sqrtf(e);
                __syncthreads();
                if (err < thr || err != err) break;          // POT: `if err < stopThr: break` (strict)
            }
            { float *t = hv_cur; hv_cur = hv_new; hv_new = t; }
            // ---- u-update:  hu_i = log a - LSE_j(2 alpha <x_i, y_j> + hv_j) -------------------------------------------
            lse_pass<KB, AUG, NP>(Xb, o_p, np, o_q, nq, hv_cur, hu, loga2, ii > 0, wave, n_waves, lane,
                         [&](int row, float lse2) { hu[row] = loga2 - lse2; });
            __syncthreads();
            iters = ii + 1;
        }
        // ---- value <Gamma, C> ---------------------------------------------------------------------------------------
        float part = value_pass<KB, NP, (AUG && NP == 2)>(Xb, o_p, nxp, np, o_q, nyq, nq, hu, hv_cur, p.dot_unscale, p.inv_scale, wave, n_waves, lane);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) red[16 + wave] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int w = 0; w < n_waves; ++w) tot += red[16 + w];
            p.w2[out] = (double)tot;
            if (p.iters) p.iters[out] = iters;
            if (p.err) p.err[out] = (double)err;
        }
        __syncthreads();
    }
}

}  // namespace pilot
