// Sinkhorn pair-grid kernel for 128 < K <= 256 cell types (symmetric cost, max(M)/reg <= 16): the fp16-split formulation of
// sinkhorn_stream_kernel<CfgH32x16, ...> with the cell types of a 16-pair tile spread over the EIGHT waves of a workgroup.
//
// Why a kernel of its own: one wave per tile needs 5 panels x RT x 4 registers per lane and the stationary operand image
// (2 fp16 pieces x K^2 x 2 B = 256 KB at K = 256) in LDS -- neither fits beyond 8 row-tiles, and round 3 sent every K > 128
// to the POT-literal kernel (one workgroup per pair, K' streamed from L2 twice per update): 151 ms for 200 patients at K = 130
// against 0.7 ms at K = 128.  Here wave w owns the OUTPUT row-tiles 2w and 2w + 1:
//   * its rows of the operand image live in REGISTERS (2 pieces x 8 k-blocks x 2 tiles x 16 B = 128 VGPRs per lane, loaded
//     once per wave; G^T = G serves both products), so no image is ever read in the update loop;
//   * the accumulator registers of tiles 2w, 2w + 1 are exactly k-block w of the next product's B operand (the layout rule
//     of the stream kernel), so after the element-wise step a wave publishes ONE k-block of packed pieces (2 KB) in LDS
//     and every wave reads all eight: two workgroup barriers per update, 16 KB of panel per product;
//   * per-column decisions (tau test, marginal error, stop) are taken from LDS reductions that every wave reads in the same
//     order, so the replicated control state never diverges and a pair's bits do not depend on its slot or workgroup;
//   * a finished pair leaves its (u, v) pieces in a global record (the ring-slot format of the stream kernel); the costs
//     <Gamma, M> are formed afterwards by sinkhorn_wide_value_kernel, 16 records per wave with ring_flush_body;
//   * a pair in which POT would tau-absorb, or that ends as NaN, is marked in its record and forwarded by the value kernel to
//     the POT-literal kernel (nan_list) -- at max(M)/reg <= 16 and K > 128 those are a handful.
// Same scaled domain, same stopping rule (f32 floor of the threshold) and the same tolerance as the fp16-split stream kernel.
#pragma once
#include "sinkhorn_kernels.hpp"

namespace pilot {

constexpr int WIDE_WAVES = 8, WIDE_RT = 16, WIDE_KB = 8, WIDE_KP = 256, WIDE_MAX_K = 256;
constexpr int WIDE_PE = ring_panel_elems<CfgH32x16>(WIDE_RT);        // 256 4-byte words per panel of a record
constexpr int WIDE_REC = ring_slot_stride<CfgH32x16>(WIDE_RT);       // words per record: u pieces, v pieces, scale, q, flags, pad

__global__ void __launch_bounds__(WAVE * WIDE_WAVES, 2) sinkhorn_wide_kernel(GridParams p, float *__restrict__ rec_base) {
    using C = CfgH32x16;
    using acc_t = C::acc_t;
    constexpr int TILE = C::TILE, NREG = C::NREG, NGRP = C::NGRP, KB = WIDE_KB, RT = WIDE_RT, KP = WIDE_KP;
    __shared__ u32x4_t PB[2][KB][2][WAVE];          // [v panel, u panel][k-block][piece][lane]: the B operands of the two products
    __shared__ int ovc[2][TILE];                    // [iteration parity][column]: some scaling of the column is over tau
    __shared__ float red_e2[2][WIDE_WAVES][TILE];   // [parity][wave][column]: partial squared marginal errors
    __shared__ int sh_base[2];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, col = lane % TILE, grp = lane / TILE;
    const int K = p.K, N = p.N, n_items = p.n_pairs;
    const int KBL = (K + 31) / 32;                                      // live k-blocks
    const bool live_wave = 2 * wave * TILE < K;                         // this wave's tiles hold cell types at all
    const float *img = static_cast<const float *>(p.img);               // form 0: G^T == G (symmetric cost)
    const float *Pt = static_cast<const float *>(p.P);
    const float *acc0 = img + acc0_offset<C>(RT);
    const float uinit = H_PANEL_SCALE / float(K), tau = float(p.tau) * H_PANEL_SCALE;
    const unsigned long long colmask = (1ull << TILE) - 1ull;

    // my rows of the operand image: [piece][k-block][local tile]
    u32x4_t AR[2][KB][2];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
                AR[part][kb][tl] = reinterpret_cast<const u32x4_t *>(img)[((part * KB + kb) * RT + (2 * wave + tl)) * WAVE + lane];
    acc_t PADC[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) PADC[tl][r] = C::lidx(2 * wave + tl, r, grp) >= K ? 1.f : 0.f;

    auto product = [&](int panel, acc_t (&OUT)[2]) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) OUT[tl] = PADC[tl];              // 1 in padded slots keeps 0 / OUT finite there
        if (live_wave) {                                                // (wave-uniform)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                if (kb < KBL) {                                         // (wave-uniform)
                    const u32x4_t b0 = PB[panel][kb][0][lane], b1 = PB[panel][kb][1][lane];
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {                    // piece products smallest first: a2 b1, a1 b2, a1 b1
                        OUT[tl] = mfma_pieces<C>(AR[1][kb][tl], b0, OUT[tl]);
                        OUT[tl] = mfma_pieces<C>(AR[0][kb][tl], b1, OUT[tl]);
                        OUT[tl] = mfma_pieces<C>(AR[0][kb][tl], b0, OUT[tl]);
                    }
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

    bool active = false, want = true, exhausted = false;
    int q = 0, ii = 0, chk = 1, flags = 0;
    float errv = 1.f, thr = 0.f;
    acc_t A[2], B[2], V[2], U[2], ACC[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < NREG; ++r) { A[tl][r] = B[tl][r] = V[tl][r] = U[tl][r] = 0.f; ACC[tl][r] = 1.f; }
    u32x4_t pu_hi = {0u, 0u, 0u, 0u}, pu_lo = pu_hi, pv_hi = pu_hi, pv_lo = pu_hi;
    int res_next = 0, res_end = 0, res_base = 0, qbatch = 0, ibatch = 0, jbatch = 0, draws = 0;
    const bool all_over = p.unequal && *p.unequal != 0;                  // histograms of unequal mass: see the stream kernel
    if (threadIdx.x < 2 * TILE) (&ovc[0][0])[threadIdx.x] = 0;
    __syncthreads();

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
                pu_hi = pu_lo = u32x4_t{0u, 0u, 0u, 0u};
            }
            if (take) {
                want = false; active = true;
                q = qsel;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int t = 2 * wave + tl;
                    load_regs<C>(Pt + (size_t)isel * KP + (t * NGRP + grp) * NREG, A[tl]);
                    load_regs<C>(Pt + (size_t)jsel * KP + (t * NGRP + grp) * NREG, B[tl]);
                    load_regs<C>(acc0 + (t * NGRP + grp) * NREG, ACC[tl]);
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        A[tl][r] *= H_IN_SCALE; B[tl][r] *= H_IN_SCALE;
                        U[tl][r] = uinit - uinit * PADC[tl][r];             // u0 = 1/K, 0 in padded slots
                    }
                }
                pieces_of(U, pu_hi, pu_lo);
                thr = Pt[(size_t)N * KP + jsel] * H_IN_SCALE;
                chk = 1; ii = 0; flags = 0; errv = 1.f;
                if (all_over) {                     // not a problem for the scaled fp16 domain: straight to the POT-literal kernel
                    if (wave == 0 && grp == 0) {
                        int *meta = reinterpret_cast<int *>(rec_base + (size_t)q * WIDE_REC + 2 * WIDE_PE + 1);
                        meta[0] = q; meta[1] = FLAG_NAN;
                    }
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
        pieces_of(V, pv_hi, pv_lo);
        PB[0][wave][0][lane] = pv_hi; PB[0][wave][1][lane] = pv_lo;
        if (active && !(mx <= tau)) ovc[par][col] = 1;                   // (NaN counts as over: caught below as a hand-over)
        __syncthreads();
        // ---- u = a / (G v) ------------------------------------------------------------------------------------------------
        product(0, ACC);
        mx = 0.f;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < NREG; ++r) { U[tl][r] = A[tl][r] * C::rcp(ACC[tl][r]); mx = fmaxf(mx, U[tl][r]); }
        pieces_of(U, pu_hi, pu_lo);
        PB[1][wave][0][lane] = pu_hi; PB[1][wave][1][lane] = pu_lo;
        if (active && !(mx <= tau)) ovc[par][col] = 1;
        if (threadIdx.x < TILE) ovc[par ^ 1][threadIdx.x] = 0;           // next iteration's flags (nobody reads them before barrier 2 of it)
        __syncthreads();
        // POT: max|u| > tau or max|v| > tau -> absorb: such a pair leaves the scaled domain and goes to the POT-literal kernel
        const bool over = active && ovc[par][col] != 0;
        if (over) {
            if (wave == 0 && grp == 0) {
                int *meta = reinterpret_cast<int *>(rec_base + (size_t)q * WIDE_REC + 2 * WIDE_PE + 1);
                meta[0] = q; meta[1] = FLAG_NAN;
            }
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
            for (int w = 0; w < WIDE_WAVES; ++w) tot += red_e2[par][w][col];     // wave order: the same sum in every wave
            const float e = sqrtf(tot);
            bool fin = capped;
            if (pending) {
                errv = e;
                if (e <= thr) { fin = true; flags |= FLAG_CONVERGED; }
                else if (e != e) { fin = true; flags |= FLAG_NAN; }
            }
            if (fin) {
                // the pair's record: my k-block of the u and v pieces; wave 0 adds the scale, the output index and the flags
                float *rec = rec_base + (size_t)q * WIDE_REC;
                *reinterpret_cast<u32x4_t *>(rec + ((0 * KB + wave) * NGRP + grp) * 4) = pu_hi;
                *reinterpret_cast<u32x4_t *>(rec + ((1 * KB + wave) * NGRP + grp) * 4) = pu_lo;
                *reinterpret_cast<u32x4_t *>(rec + WIDE_PE + ((0 * KB + wave) * NGRP + grp) * 4) = pv_hi;
                *reinterpret_cast<u32x4_t *>(rec + WIDE_PE + ((1 * KB + wave) * NGRP + grp) * 4) = pv_lo;
                if (wave == 0 && grp == 0) {
                    rec[2 * WIDE_PE] = 1.f;
                    int *meta = reinterpret_cast<int *>(rec + 2 * WIDE_PE + 1);
                    meta[0] = q; meta[1] = flags;
                    if (p.iters) p.iters[q] = ii;
                    if (p.err) p.err[q] = double(errv) * double(1.f / H_IN_SCALE);
                }
                active = false; want = true;
            }
        }
    }
}

// costs of the finished pairs: 16 consecutive records per wave through the flush of the stream kernel (one panel product with
// the G o M image from L2); records marked FLAG_NAN (hand-overs) go to nan_list there
__global__ void __launch_bounds__(WAVE * WAVES_PER_WG) sinkhorn_wide_value_kernel(GridParams p, const float *__restrict__ rec_base) {
    const int n_tiles = (p.n_pairs + 15) / 16;
    const int n_waves = (int)gridDim.x * WAVES_PER_WG;
    for (int tile = (int)blockIdx.x * WAVES_PER_WG + (int)threadIdx.x / WAVE; tile < n_tiles; tile += n_waves) {
        const int cnt = p.n_pairs - tile * 16 < 16 ? p.n_pairs - tile * 16 : 16;
        ring_flush_body<CfgH32x16, WIDE_RT, true>(rec_base + (size_t)tile * 16 * WIDE_REC, p, cnt);
    }
}

}  // namespace pilot
