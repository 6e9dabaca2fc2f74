// Sinkhorn pair-grid kernel for 96 < K <= 128 cell types (symmetric cost, fp16-split range): FOUR waves per 16-pair tile.
//
// Why.  From 7 row-tiles on the one-wave-per-tile kernel (sinkhorn_stream_kernel) reads its operand image from LDS for every
// pair of MFMAs, keeps 5 panels of 7 x 4 registers and spills (400 B per lane at K = 100): c4 (2000 x 100, the shape BASELINE gives
// to 8 GPUs) ran 23.2 ms with the matrix pipe half busy.  Here, like in sinkhorn_wide_kernel (128 < K <= 256, eight waves), the cell
// types of a tile are spread over the waves of a 256-thread workgroup -- wave w owns the output row-tiles 2 w and 2 w + 1:
//   * its rows of the image live in REGISTERS (2 pieces x 4 k-blocks x 2 tiles x 16 B = 64 VGPRs), loaded once per wave;
//     G^T = G serves both products, nothing but panels moves in the update loop;
//   * the accumulator registers of tiles 2 w, 2 w + 1 are exactly k-block w of the next product's B operand, so after the
//     element-wise step a wave publishes ONE k-block of packed pieces (2 KB) in LDS and reads all four: two workgroup
//     barriers per update;
//   * control state is replicated in every wave and moves only on values every wave reads identically from LDS (the tau
//     flags of the columns, the four partial squared errors added in wave order), so the waves never diverge and a pair's bits
//     do not depend on its slot, its workgroup or the row subset of the call;
//   * the operand block, the slot-ordered proportions, the longest-first work list, the hand-over list of pairs in which POT
//     would tau-absorb (solved by the bf16 tracking kernel) and the NaN list are those of the stream kernel's fast pass: this
//     kernel takes its place in the same call sequence;
//   * THE COSTS ARE FORMED INSIDE THE KERNEL (round 5's four-wave experiment left records for a second kernel and lost its
//     18 - 26 % again in that round trip): a finished pair parks its (u, v) pieces -- every wave its own k-block -- in a ring
//     of 16 slots in LDS; when the ring is full the four waves form <Gamma, M> = u^T (G o M) v for all 16 pairs with ONE more
//     panel product (each wave its rows of G o M, read from L2), the per-wave partial sums meet in LDS in wave order.
// The last k-block of an odd row-tile count (K <= 112) holds one row-tile and runs on v_mfma_f32_16x16x16_f16 (see
// mfma_pieces<C, true> and tail16_gap).
// Same scaled domain, stopping rule (f32 floor of the threshold) and tolerance as the fp16-split stream kernel.
#pragma once
#include "sinkhorn_kernels.hpp"

namespace pilot {

constexpr int QUAD_WAVES = 4, QUAD_KB = 4, QUAD_RING = 16, QUAD_MIN_K = 97, QUAD_MAX_K = 128;

// The costs of the cnt pairs in the ring: one more panel product by all four waves (cnt is the same in every wave).  Deliberately
// NOT inlined: it runs once per 16 finished pairs, and as a call its registers (an accumulator pair, eight operand registers of
// G o M in flight) are paid at the call site instead of in the update loop, which must fit 168 VGPRs for three workgroups per CU.
template <int RT>
__device__ __attribute__((noinline)) void quad_flush(const GridParams &p, int cnt, const u32x4_t (*ring_pu)[QUAD_KB][2][CfgH32x16::NGRP],
                                                     const u32x4_t (*ring_pv)[QUAD_KB][2][CfgH32x16::NGRP], const int (*ring_meta)[2],
                                                     float (*red_val)[CfgH32x16::TILE]) {
    using C = CfgH32x16;
    using acc_t = C::acc_t;
    constexpr int TILE = C::TILE, NREG = C::NREG, KB = QUAD_KB;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, col = lane % TILE, grp = lane / TILE;
    constexpr int KBL = (RT + 1) / 2;
    constexpr bool tail16 = PILOT_TAIL16 && (RT & 1);
    constexpr int FORM = form_elems<C>(RT);
    const float *img = static_cast<const float *>(p.img);
    __syncthreads();                                                    // every wave's ring stores are visible
    const int s = col < cnt ? col : cnt - 1;                            // columns beyond the fill level redo the last slot, unused
    acc_t W[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) W[tl][r] = 0.f;
    {
        // my rows of G o M (form 2) from L2, one tile at a time
        const u32x4_t *gm = reinterpret_cast<const u32x4_t *>(img + 2 * FORM);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            if (2 * wave + tl >= RT) continue;                          // (wave-uniform: tile 7 is dead up to K = 112)
            u32x4_t G[2][KB];
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    G[part][kb] = u32x4_t{0u, 0u, 0u, 0u};
                    if (kb < KBL) G[part][kb] = gm[((part * KBL + kb) * RT + (2 * wave + tl)) * WAVE + lane];
                }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb >= KBL) continue;
                const u32x4_t b0 = ring_pv[s][kb][0][grp], b1 = ring_pv[s][kb][1][grp];
                if (tail16 && kb == KBL - 1) {
                    tail16_gap();                                       // (the tile's full MFMAs are right in front)
                    W[tl] = mfma_pieces<C, true>(G[1][kb], b0, W[tl]);
                    W[tl] = mfma_pieces<C, true>(G[0][kb], b1, W[tl]);
                    W[tl] = mfma_pieces<C, true>(G[0][kb], b0, W[tl]);
                } else {
                    W[tl] = mfma_pieces<C>(G[1][kb], b0, W[tl]);
                    W[tl] = mfma_pieces<C>(G[0][kb], b1, W[tl]);
                    W[tl] = mfma_pieces<C>(G[0][kb], b0, W[tl]);
                }
            }
        }
    }
    // u of my rows = hi + lo of my k-block of the slot's u pieces
    const u32x4_t uh = ring_pu[s][wave][0][grp], ul = ring_pu[s][wave][1][grp];
    float val = 0.f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        val += pieces_sum_lo(uh[h], ul[h]) * W[h / 2][2 * (h & 1)];
        val += pieces_sum_hi(uh[h], ul[h]) * W[h / 2][2 * (h & 1) + 1];
    }
    val = group_sum<C>(val);
    if (grp == 0) red_val[wave][col] = val;
    __syncthreads();
    if (wave == 0 && grp == 0 && col < cnt) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < QUAD_WAVES; ++w) tot += red_val[w][col];             // wave order: one sum, whatever the timing
        tot *= 1.f / H_IN_SCALE;                                        // u~^T (2^15 G o M) v~ = 2^25 u^T (G o M) v
        const int qq = ring_meta[col][0];
        int fl = ring_meta[col][1];
        if (p.nan_list && (!(tot - tot == 0.f) || (fl & FLAG_NAN))) {               // NaN or inf: the POT-literal kernel solves the pair again
            p.nan_list[__hip_atomic_fetch_add(p.nan_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = qq;
        } else {
            if (tot != tot) fl |= FLAG_NAN;
            p.emd[qq] = double(tot);
            p.flags[qq] = fl;
        }
    }
    __syncthreads();                                                    // the slots are reused only after every lane has read them
}

template <int RT>
__global__ void __launch_bounds__(WAVE * QUAD_WAVES, 3) sinkhorn_quad_kernel(GridParams p) {
    static_assert(RT == 7 || RT == 8, "four waves own two row-tiles each");
    using C = CfgH32x16;
    using acc_t = C::acc_t;
    constexpr int TILE = C::TILE, NREG = C::NREG, NGRP = C::NGRP, KB = QUAD_KB;
    __shared__ u32x4_t PB[2][KB][2][WAVE];                  // [v panel, u panel][k-block][piece][lane]: the B operands of the two products
    __shared__ u32x4_t ring_pu[QUAD_RING][KB][2][NGRP];     // finished pairs: [slot][k-block][piece][lane group] u pieces ...
    __shared__ u32x4_t ring_pv[QUAD_RING][KB][2][NGRP];     // ... and v pieces (the B operand of the cost product as it lies)
    __shared__ int ring_meta[QUAD_RING][2];                 // q, flags
    __shared__ int ovc[2][TILE];                            // [iteration parity][column]: some scaling of the column is over tau
    __shared__ float red_e2[2][QUAD_WAVES][TILE];           // [parity][wave][column]: partial squared marginal errors
    __shared__ float red_val[QUAD_WAVES][TILE];             // [wave][slot]: partial costs of a flush
    __shared__ int sh_base[2];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, col = lane % TILE, grp = lane / TILE;
    const int K = p.K, N = p.N;
    constexpr int KP = RT * TILE;                                       // 7 or 8 row-tiles: the layout the prep kernel wrote
    constexpr int KBL = (RT + 1) / 2;                                   // live k-blocks (4 here)
    constexpr bool tail16 = PILOT_TAIL16 && (RT & 1);                   // the last k-block holds one row-tile
    const int n_items = p.list_len ? *p.list_len : p.n_pairs;
    const float *img = static_cast<const float *>(p.img);               // form 0: G^T == G (symmetric cost)
    constexpr int FORM = form_elems<C>(RT);
    const float *Pt = static_cast<const float *>(p.P) + (size_t)N * KP + N;     // the copy in the scaled domain (prep kernel)
    const float *acc0 = img + 3 * FORM;
    const float tau = float(p.tau) * H_PANEL_SCALE;
    const unsigned long long colmask = (1ull << TILE) - 1ull;
    bool tile_live[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) tile_live[tl] = 2 * wave + tl < RT;  // (wave-uniform; tile 7 is dead up to K = 112)

    // my rows of the operand image: [piece][k-block][local tile]
    u32x4_t AR[2][KB][2];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                AR[part][kb][tl] = u32x4_t{0u, 0u, 0u, 0u};
                if (tile_live[tl] && kb < KBL)                       // (a dead tile keeps a zero image)
                    AR[part][kb][tl] = reinterpret_cast<const u32x4_t *>(img)[((part * KBL + kb) * RT + (2 * wave + tl)) * WAVE + lane];
            }
    // padded slots (cell types beyond K; every slot of a dead tile) as a bit mask: their accumulators start at 1, which keeps 0 / OUT finite
    unsigned int padmask = 0u;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) padmask |= (C::lidx(2 * wave + tl, r, grp) >= K ? 1u : 0u) << (tl * NREG + r);

    // OUT (my two tiles) = image rows x the panel in PB[panel]; piece products smallest first: a2 b1, a1 b2, a1 b1.  Straight-line code:
    // a dead tile (tile 7 up to K = 112, wave 3 only) multiplies a zero image -- the other three waves set the pace anyway.
    auto product = [&](int panel, acc_t (&OUT)[2]) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < NREG; ++r) OUT[tl][r] = (padmask >> (tl * NREG + r)) & 1u ? 1.f : 0.f;
#pragma unroll
        for (int kb = 0; kb < KBL; ++kb) {
            const u32x4_t b0 = PB[panel][kb][0][lane], b1 = PB[panel][kb][1][lane];
            if constexpr (tail16) {
                if (kb == KBL - 1) {                                     // one row-tile of k-slots: the 16-deep instruction
                    // (a tile's tail MFMAs follow its last full MFMA behind three other MFMAs; the gap keeps the 16x16x32 -> 16x16x16
                    // accumulator hazard out whatever the distance, see tail16_gap)
                    tail16_gap();
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        OUT[tl] = mfma_pieces<C, true>(AR[1][kb][tl], b0, OUT[tl]);
                        OUT[tl] = mfma_pieces<C, true>(AR[0][kb][tl], b1, OUT[tl]);
                        OUT[tl] = mfma_pieces<C, true>(AR[0][kb][tl], b0, OUT[tl]);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                OUT[tl] = mfma_pieces<C>(AR[1][kb][tl], b0, OUT[tl]);
                OUT[tl] = mfma_pieces<C>(AR[0][kb][tl], b1, OUT[tl]);
                OUT[tl] = mfma_pieces<C>(AR[0][kb][tl], b0, OUT[tl]);
            }
        }
    };
    // X (my two tiles) -> the packed pieces of my k-block
    auto pieces_of = [&](const acc_t (&X)[2], u32x4_t &hi, u32x4_t &lo) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            unsigned int a, b;
            quot_pieces(X[h / 2][2 * (h & 1)], X[h / 2][2 * (h & 1) + 1], a, b);
            hi[h] = a; lo[h] = b;
        }
    };

    auto flush = [&](int cnt) { quad_flush<RT>(p, cnt, ring_pu, ring_pv, ring_meta, red_val); };

    bool active = false, want = true, exhausted = false;
    int q = 0, ii = 0, chk = 1, flags = 0, ring_cnt = 0;
    float errv = 1.f, thr = 0.f;
    acc_t A[2], B[2], V[2], ACC[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) { A[tl][r] = B[tl][r] = V[tl][r] = 0.f; ACC[tl][r] = 1.f; }
    int res_next = 0, res_end = 0, res_base = 0, qbatch = 0, ibatch = 0, jbatch = 0, draws = 0;
    const bool all_over = p.unequal && *p.unequal != 0;                  // histograms of unequal mass: see the stream kernel
    if (threadIdx.x < 2 * TILE) (&ovc[0][0])[threadIdx.x] = 0;
    __syncthreads();
    if ((int)blockIdx.x * TILE >= n_items) return;                       // more workgroups than work

    for (int it = 0;; ++it) {
        const int par = it & 1;
        // ---- (re)fill columns: every wave runs the same logic on the same replicated state; the queue atomic is wave 0's ----
        const unsigned long long wmask = __ballot(want) & colmask;
        if (wmask) {
            if (res_next >= res_end && !exhausted) {
                int base;
                if (draws == 0) {                   // the first batch is the workgroup's own number: no atomic, no barrier
                    base = (int)blockIdx.x * TILE;
                } else {
                    if (threadIdx.x == 0)
                        sh_base[draws & 1] = (int)gridDim.x * TILE + __hip_atomic_fetch_add(p.queue_head, TILE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    base = __builtin_amdgcn_readfirstlane(sh_base[draws & 1]);
                }
                ++draws;
                exhausted = base >= n_items;
                res_next = exhausted ? n_items : base;
                res_end = (base + TILE < n_items) ? base + TILE : n_items;
                if (exhausted) res_end = n_items;
                res_base = base;
                const int bi = base + col;
                qbatch = (p.list && bi < n_items) ? p.list[bi] : bi;
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
            const int qsel = __builtin_amdgcn_ds_bpermute(bsel, qbatch);
            const int isel = __builtin_amdgcn_ds_bpermute(bsel, ibatch), jsel = __builtin_amdgcn_ds_bpermute(bsel, jbatch);
            res_next = __builtin_amdgcn_readfirstlane(res_next + (n_want < avail ? n_want : avail));
            if (want && !take && exhausted) {       // no work left: the slot goes dark
                want = false;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) { A[tl][r] = B[tl][r] = 0.f; ACC[tl][r] = 1.f; }
            }
            if (take) {
                want = false; active = true;
                q = qsel;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int t = 2 * wave + tl;
#pragma unroll
                    for (int r = 0; r < NREG; ++r) { A[tl][r] = B[tl][r] = 0.f; ACC[tl][r] = 1.f; }
                    if (tile_live[tl]) {
                        load_regs<C>(Pt + (size_t)isel * KP + (t * NGRP + grp) * NREG, A[tl]);
                        load_regs<C>(Pt + (size_t)jsel * KP + (t * NGRP + grp) * NREG, B[tl]);
                        load_regs<C>(acc0 + (t * NGRP + grp) * NREG, ACC[tl]);
                    }
                }
                thr = Pt[(size_t)N * KP + jsel];                       // (u0 = 1/K enters through the first-product table ACC = G^T u0)
                chk = 1; ii = 0; flags = 0; errv = 1.f;
                if (all_over) {                     // not a problem for the scaled fp16 domain: straight to the tracking kernel
                    if (wave == 0 && grp == 0) p.track_list[__hip_atomic_fetch_add(p.track_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = q;
                    active = false; want = true;
                }
            }
        }
        if (__ballot(active || want) == 0ull) break;

        // ---- v = b / (G^T u): my k-block of the v panel ----------------------------------------------------------------
        float mx = 0.f;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < NREG; ++r) { V[tl][r] = B[tl][r] * C::rcp(ACC[tl][r]); mx = fmaxf(mx, V[tl][r]); }
        {
            u32x4_t hi, lo;
            pieces_of(V, hi, lo);
            PB[0][wave][0][lane] = hi; PB[0][wave][1][lane] = lo;
        }
        if (active && !(mx <= tau)) ovc[par][col] = 1;                   // (NaN counts as over: the tracking kernel restarts the pair)
        __syncthreads();
        // ---- u = a / (G v) ------------------------------------------------------------------------------------------------
        product(0, ACC);
        mx = 0.f;
        {
            acc_t U[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int r = 0; r < NREG; ++r) { U[tl][r] = A[tl][r] * C::rcp(ACC[tl][r]); mx = fmaxf(mx, U[tl][r]); }
            u32x4_t hi, lo;
            pieces_of(U, hi, lo);
            PB[1][wave][0][lane] = hi; PB[1][wave][1][lane] = lo;
        }
        if (active && !(mx <= tau)) ovc[par][col] = 1;
        if (threadIdx.x < TILE) ovc[par ^ 1][threadIdx.x] = 0;           // next iteration's flags (nobody reads them before barrier 2 of it)
        __syncthreads();
        // POT: max|u| > tau or max|v| > tau -> absorb: the pair leaves the scaled domain; the tracking kernel restarts it
        const bool over = active && ovc[par][col] != 0;
        if (over) {
            if (wave == 0 && grp == 0) p.track_list[__hip_atomic_fetch_add(p.track_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = q;
            active = false; want = true;
        }
        ++ii;
        // ---- ACC = G^T u: the stopping test of this update and the next v ------------------------------------------------
        product(1, ACC);
        const bool pending = active && ii == chk;
        if (pending) chk += p.period;
        const bool capped = active && ii >= p.max_iter;
        if (__ballot(pending || capped)) {                               // (the same in every wave)
            float e2 = 0.f;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                float et = 0.f;
#pragma unroll
                for (int r = 0; r < NREG; ++r) { const float d = V[tl][r] * ACC[tl][r] - B[tl][r]; et += d * d; }
                e2 += et;
            }
            e2 = group_sum<C>(e2);
            if (grp == 0) red_e2[par][wave][col] = e2;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < QUAD_WAVES; ++w) tot += red_e2[par][w][col];     // wave order: the same sum in every wave
            const float e = sqrtf(tot);
            bool fin = capped;
            if (pending) {
                errv = e;
                if (e <= thr) { fin = true; flags |= FLAG_CONVERGED; }
                else if (e != e) { fin = true; flags |= FLAG_NAN; }
            }
            // ---- retire finished pairs into the ring (every wave its own k-block); a full ring is flushed first ----
            unsigned long long fmask = __ballot(fin) & colmask;
            if (fmask) {
                if (fin && wave == 0 && grp == 0) {
                    if (p.iters) p.iters[q] = ii;
                    if (p.err) p.err[q] = double(errv) * double(1.f / H_IN_SCALE);
                }
                while (fmask) {                                         // (the same in every wave; a second pass when the ring fills up in between)
                    if (ring_cnt >= QUAD_RING) { flush(ring_cnt); ring_cnt = 0; }
                    const int space = QUAD_RING - ring_cnt;
                    const int rank = (int)__popcll(fmask & ((1ull << col) - 1ull));
                    const bool put = fin && ((fmask >> col) & 1ull) && rank < space;
                    if (put) {
                        const int s = ring_cnt + rank;
                        // (my k-block of this update's u and v pieces still lies in PB: nothing writes it before the next update)
                        ring_pu[s][wave][0][grp] = PB[1][wave][0][lane]; ring_pu[s][wave][1][grp] = PB[1][wave][1][lane];
                        ring_pv[s][wave][0][grp] = PB[0][wave][0][lane]; ring_pv[s][wave][1][grp] = PB[0][wave][1][lane];
                        if (wave == 0 && grp == 0) { ring_meta[s][0] = q; ring_meta[s][1] = flags; }
                    }
                    const unsigned long long taken = __ballot(put) & colmask;
                    fmask &= ~taken;
                    ring_cnt = __builtin_amdgcn_readfirstlane(ring_cnt + (int)__popcll(taken));
                }
                if (fin) { active = false; want = true; }
            }
        }
    }
    if (ring_cnt > 0) flush(ring_cnt);
}

}  // namespace pilot
