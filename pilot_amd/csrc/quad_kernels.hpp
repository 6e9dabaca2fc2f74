// Sinkhorn pair-grid kernel for 112 < K <= 128 cell types (8 row-tiles; symmetric cost, fp16-split range): FOUR waves per 16-pair tile.
//
// Why.  At 8 row-tiles the one-wave-per-tile kernel (sinkhorn_stream_kernel) needs 347 registers, runs one wave per SIMD and reads
// its operand image from LDS for every pair of MFMAs: K = 128 at N = 600 took 1.85 ms against 1.0 ms at K = 96.  Here, like in
// sinkhorn_wide_kernel (128 < K <= 256, eight waves), the cell types of a tile are spread over the waves of a 256-thread workgroup --
// wave w owns the output row-tiles 2 w and 2 w + 1:
//   * its rows of the image AND of G o M live in REGISTERS (2 x 2 pieces x 4 k-blocks x 2 tiles x 16 B = 128 VGPRs), loaded once
//     per wave; G^T = G serves both products, nothing but panels moves in the update loop; two workgroups per CU;
//   * the accumulator registers of tiles 2 w, 2 w + 1 are exactly k-block w of the next product's B operand, so after the
//     element-wise step a wave publishes ONE k-block of packed pieces (2 KB) in LDS and reads all four: two workgroup
//     barriers per update (LDS-only barriers: lds_barrier);
//   * control state is replicated in every wave and moves only on values every wave reads identically from LDS (the tau
//     flags of the columns, the four partial squared errors added in wave order), so the waves never diverge and a pair's bits
//     do not depend on its slot, its workgroup or the row subset of the call;
//   * the operand block, the slot-ordered proportions, the longest-first work list, the hand-over list of pairs in which POT
//     would tau-absorb (solved by the bf16 tracking kernel) and the NaN list are those of the stream kernel's fast pass: this
//     kernel takes its place in the same call sequence;
//   * the costs are formed inside the kernel: a finished pair parks its (u, v) pieces -- every wave its own k-block -- in a ring
//     of 32 slots in LDS; when the ring is full the four waves form <Gamma, M> = u^T (G o M) v with one more panel product per 16
//     pairs and the per-wave partial sums meet in LDS in wave order.
// Why not from 7 row-tiles (K = 97 .. 112, BASELINE's c4 among them), where the update loop alone (140 VGPRs, three workgroups per CU)
// runs K = 100 at N = 600 in 1.32 ms against the stream kernel's 1.60: every way of giving the cost flush its rows of G o M costs more
// than that -- a second register image means two workgroups per CU (1.69 ms); a call that fetches the rows pays its register saves
// (1.61 ms); streaming them through spare registers, or swapping the register image for the flush, spills the update loop (1.77 /
// 2.15 ms); records for a second kernel are round 5's 2.3 ms at c4.  The table is in profiles/r06/ab_experiments.md section 5.
// Same scaled domain, stopping rule (f32 floor of the threshold) and tolerance as the fp16-split stream kernel.
#pragma once
#include "sinkhorn_kernels.hpp"

namespace pilot {

constexpr int QUAD_WAVES = 4, QUAD_KB = 4, QUAD_RING = 32, QUAD_MIN_K = 113, QUAD_MAX_K = 128;

// Workgroup barrier for data that travels through LDS only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0): the outputs of finished pairs, a microsecond or two until L2 acknowledges them) -- in a kernel that meets at
// two barriers per update and writes outputs now and then that wait was 6 us per cost flush (K = 128 at N = 600: 1.58 ms with
// __syncthreads, see profiles/r06/ab_experiments.md).  Nothing the waves of a workgroup tell each other here goes through global memory.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RT>
__global__ void __launch_bounds__(WAVE * QUAD_WAVES, 2) sinkhorn_quad_kernel(GridParams p) {
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

    // my rows of the operand image: [piece][k-block][local tile] (loaded below, and again after every cost flush)
    u32x4_t AR[2][KB][2];
    // padded slots (cell types beyond K; every slot of a dead tile) as a bit mask: their accumulators start at 1, which keeps 0 / OUT finite
    unsigned int padmask = 0u;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) padmask |= (C::lidx(2 * wave + tl, r, grp) >= K ? 1u : 0u) << (tl * NREG + r);

    // OUT (my two tiles) = image rows x the panel in PB[panel]; piece products smallest first: a2 b1, a1 b2, a1 b1.  Straight-line code:
    // a dead tile (tile 7 up to K = 112, wave 3 only) multiplies a zero image -- the other three waves set the pace anyway.
    auto product_of = [&](const u32x4_t (&IMG)[2][KB][2], auto bget, acc_t (&OUT)[2]) {
#pragma unroll
        for (int kb = 0; kb < KBL; ++kb) {
            const u32x4_t b0 = bget(kb, 0), b1 = bget(kb, 1);
            if constexpr (tail16) {
                if (kb == KBL - 1) {                                     // one row-tile of k-slots: the 16-deep instruction
                    // (a tile's tail MFMAs follow its last full MFMA behind three other MFMAs; the gap keeps the 16x16x32 -> 16x16x16
                    // accumulator hazard out whatever the distance, see tail16_gap)
                    tail16_gap();
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        OUT[tl] = mfma_pieces<C, true>(IMG[1][kb][tl], b0, OUT[tl]);
                        OUT[tl] = mfma_pieces<C, true>(IMG[0][kb][tl], b1, OUT[tl]);
                        OUT[tl] = mfma_pieces<C, true>(IMG[0][kb][tl], b0, OUT[tl]);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                OUT[tl] = mfma_pieces<C>(IMG[1][kb][tl], b0, OUT[tl]);
                OUT[tl] = mfma_pieces<C>(IMG[0][kb][tl], b1, OUT[tl]);
                OUT[tl] = mfma_pieces<C>(IMG[0][kb][tl], b0, OUT[tl]);
            }
        }
    };
    auto product = [&](int panel, acc_t (&OUT)[2]) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < NREG; ++r) OUT[tl][r] = (padmask >> (tl * NREG + r)) & 1u ? 1.f : 0.f;
        product_of(AR, [&](int kb, int part) { return PB[panel][kb][part][lane]; }, OUT);
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

    // my rows of G (form 0, the update loop's) and of G o M (form 2, the cost flush's): both stay in registers
    // (branch-free loads: a dead tile -- tile 7 at 7 row-tiles, wave 3 only -- reads its neighbour's rows and masks them to zero)
    u32x4_t AGM[2][KB][2];
    auto load_image = [&](int form, u32x4_t (&IMG)[2][KB][2]) {
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(img + form * FORM) + lane;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int kb = 0; kb < KBL; ++kb)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int t = 2 * wave + tl < RT ? 2 * wave + tl : RT - 1;
                    u32x4_t x = src[((part * KBL + kb) * RT + t) * WAVE];
                    if constexpr (RT & 1) {
                        const unsigned int m = tile_live[tl] ? 0xffffffffu : 0u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] &= m;
                    }
                    IMG[part][kb][tl] = x;
                }
    };
    // ---- the costs of the cnt pairs in the ring: one more panel product per 16 pairs, all four waves (cnt is the same in every wave) ----
    auto flush = [&](int cnt) {
        lds_barrier();                                                  // every wave's ring stores are visible
        for (int base = 0; base < cnt; base += TILE) {                  // (wave-uniform)
            const int n = cnt - base < TILE ? cnt - base : TILE;
            const int s = base + (col < n ? col : n - 1);               // columns beyond the fill level redo the last slot, unused
            acc_t W[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int r = 0; r < NREG; ++r) W[tl][r] = 0.f;
            product_of(AGM, [&](int kb, int part) { return ring_pv[s][kb][part][grp]; }, W);      // W = my rows of (G o M) v
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
            lds_barrier();
            if (wave == 0 && grp == 0 && col < n) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < QUAD_WAVES; ++w) tot += red_val[w][col];     // wave order: one sum, whatever the timing
                tot *= 1.f / H_IN_SCALE;                                // u~^T (2^15 G o M) v~ = 2^25 u^T (G o M) v
                const int qq = ring_meta[s][0];
                int fl = ring_meta[s][1];
                if (p.nan_list && (!(tot - tot == 0.f) || (fl & FLAG_NAN))) {       // NaN or inf: the POT-literal kernel solves the pair again
                    p.nan_list[__hip_atomic_fetch_add(p.nan_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = qq;
                } else {
                    if (tot != tot) fl |= FLAG_NAN;
                    p.emd[qq] = double(tot);
                    p.flags[qq] = fl;
                }
            }
            lds_barrier();                                              // red_val and the slots are reused only after every lane has read them
        }
    };

    load_image(0, AR);
    load_image(2, AGM);
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
    lds_barrier();
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
                    lds_barrier();
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
        lds_barrier();
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
        lds_barrier();
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
            lds_barrier();
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
