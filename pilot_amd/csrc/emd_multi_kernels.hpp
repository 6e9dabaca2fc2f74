// Exact optimal transport for SMALL numbers of cell types: several pairs per wavefront.
// Replaces the ot.emd2 loop of pilotpy/tools/Trajectory.py:507-511 (the reference's DEFAULT mode) where real cohorts live:
// the reference test's own cohort has 14 clusters (test/test_pilot.py:9-23), PILOT's tutorials 10 - 30 cell types.
//
// emd_grid_kernel (emd_kernels.hpp) gives a pair a whole wavefront: at K = 14 that is 14 of 64 lanes.  Here a wavefront is cut
// into 64 / G GROUPS of G lanes and each group solves a pair of its own with the same algorithm -- successive shortest
// augmenting paths from every row with supply left, node potentials, every column tied at the smallest label scanned in one
// step, several augmentations per search while the tree is valid, diagonal warm start, initial labels of a search cached while
// the source set stands -- and the same fp64 arithmetic for labels and potentials (labels kept plus the column potential,
// clamped to the previous step's label): a pair takes the same augmentations and comes out equal to rounding.
// G = 16 (four pairs per wave, K <= 16) is what is instantiated; G = 32 (two pairs, K <= 32) compiles and measured slower than one
// pair per wave at every K (profiles/r05/ab_experiments.md #1).
//   * lane c of a group owns row c and column c of its pair: supply, demand, potentials, label, predecessor links in registers;
//   * a group's minimum is a four-stage DPP butterfly inside the 16-lane row (quad_perm, row_half_mirror, row_mirror;
//     v_permlane16_swap joins two rows for G = 32), a value of lane i of the group is fetched with ds_bpermute;
//   * lane = node number for all groups at once, so the node sets of a search are 64-bit WAVE masks on the scalar unit and the
//     control flow is one loop with wave-uniform branches and group-uniform predicates (see emd_multi_kernel below): the groups
//     run their steps in lockstep, and a group that starts a new search or a new pair does so at the top of the loop while the
//     others go on;
//   * an augmentation walks nothing: labels carry the bit mask of their tree path (see below);
//   * a group draws its pairs one by one from the sharded device-wide queue of emd_grid_kernel, so a pair's bits do not depend
//     on its slot mates, its wave or the row subset of the call;
//   * flow values: K x K doubles per group in LDS (up to K = 15) or in the L2-resident global slab, zero outside the support,
//     put back to zero through the support masks when a pair is done; the support itself is a bit mask per row in registers.
#pragma once
#include <type_traits>
#include "emd_kernels.hpp"

namespace pilot {

// Lanes of one wave hand data to each other through memory (the flow block, the arc scratch) without a workgroup barrier: DS
// operations of a wave retire in order, but the COMPILER must not reorder or forward across the hand-over either.  Flows in LDS: a
// wavefront-scope release fence + wave barrier (no instructions on the hardware); flows in the global slab: a workgroup-scope
// release, as before.
template <bool FLDS> __device__ inline void emd_wave_sync() {
    if constexpr (FLDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    else EMD_FENCE();
    __builtin_amdgcn_wave_barrier();
}

// the partner's x under a DPP control that gives every lane of a row a partner
template <int CTRL> __device__ inline unsigned int dppx_u32(unsigned int x) {
    return (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, 0xf, 0xf, false);    // (old = the identity of min: the move folds into v_min_u32_dpp)
}
// minimum over the G lanes of a group, in every lane of the group
template <int G> __device__ inline unsigned int gmin_u32(unsigned int x) {
    unsigned int y;
    y = dppx_u32<0xB1>(x); x = y < x ? y : x;        // quad_perm [1,0,3,2]
    y = dppx_u32<0x4E>(x); x = y < x ? y : x;        // quad_perm [2,3,0,1]
    y = dppx_u32<0x141>(x); x = y < x ? y : x;       // row_half_mirror
    y = dppx_u32<0x140>(x); x = y < x ? y : x;       // row_mirror
    if constexpr (G == 32) {
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r[0] < r[1] ? r[0] : r[1];
    }
    return x;
}
// minimum of doubles whose bit patterns order like unsigned integers (non-negative values, +inf; see wave_min_f64)
template <int G> __device__ inline double gmin_f64(double x) {
    union { double d; unsigned int u[2]; } v, o;
    v.d = x;
    const unsigned int mh = gmin_u32<G>(v.u[1]);
    const unsigned int ml = gmin_u32<G>(v.u[1] == mh ? v.u[0] : 0xffffffffu);
    o.u[0] = ml; o.u[1] = mh;
    return o.d;
}
template <int G> __device__ inline double gsum_f64(double x) {
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    if constexpr (G == 32) { double a, b; swap16_f64(x, a, b); x = a + b; }
    return x;
}
// lane `src` (a wave lane number) of x
__device__ inline int bperm_i32(int x, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, x); }
__device__ inline double bperm_f64(double x, int src) {
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[0]);
    r.i[1] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[1]);
    return r.d;
}

constexpr int emd_multi_pitch(int G) { return G + 1; }      // row pitch of M in LDS (doubles): the rows two groups read fall on different banks
// dynamic LDS of emd_multi_kernel: M and its row minima, then (FLDS) one K x K flow block per group of every wave
__host__ __device__ constexpr size_t emd_multi_lds_bytes(int K, int G, int waves, bool flds) {
    return sizeof(double) * ((size_t)K * emd_multi_pitch(G) + K + 8 * (size_t)waves + (flds ? (size_t)waves * (64 / G) * K * K : 0));   // (+ 64 B per wave: scr)
}

// EMD_MSTAT = n (diagnostic builds): n_aug reports 1 steps, 2 relaxed rows, 4 searches, 5 source rows in rebuilds
//
// How the groups of a wave share it.  The loop below is ONE loop for the whole wave with wave-uniform branches; what a group
// does in an iteration is decided by group-uniform predicates that select results (no divergent regions on the hot path, so
// no exec-mask bookkeeping).  Lane = node number for all groups at once, so the node sets of a search -- scanned columns,
// reached rows, columns with demand -- are 64-bit WAVE masks kept on the scalar unit (a group's part is its 16 / 32-bit field);
// a field is moved into a vector register only where its lowest set bit becomes an index.
// An augmentation walks nothing: every label carries the bit mask of the rows and columns on its tree path (a relaxed column
// takes its row's mask plus its own bit, a reached row its column's), so the target's mask says which lanes take part; a path
// column tells its parent row "your forward arc goes to me" through a byte of LDS, the bottleneck is one group minimum over
// the rows' backward-arc flows (the source row contributes its supply, the target its demand), and every row on the path
// updates its own two flow entries and its own support mask.
template <int G, bool FLDS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) emd_multi_kernel(EmdParams p) {
    static_assert(G == 16 || G == 32, "groups are one or two DPP rows");
    constexpr int NG = 64 / G;
    constexpr unsigned int FM = G == 32 ? 0xffffffffu : 0xffffu;
    constexpr int MP = emd_multi_pitch(G);
    using u64 = unsigned long long;
    using pm_t = typename std::conditional<G == 16, unsigned int, u64>::type;     // path mask: rows << G | columns
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K, N = p.N;
    const int waves = (int)blockDim.x >> 6;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int c = lane & (G - 1), gsh = lane & ~(G - 1), g = lane / G;
    const bool valid = c < K;
    const u64 validM = ballot_b(valid);
    const pm_t colbit = (pm_t)1 << c, rowbit = (pm_t)1 << (G + c);
    double *Msh = reinterpret_cast<double *>(smem_raw);          // K x MP, pad columns zero
    double *rowmin = Msh + (size_t)K * MP;                        // K: min_j M_ij (initial row potentials)
    unsigned char *scr = reinterpret_cast<unsigned char *>(rowmin + K) + wave * 64;      // a byte per lane: forward column of a path row
    for (int t = threadIdx.x; t < K * MP; t += blockDim.x) {
        const int i = t / MP, j = t % MP;
        Msh[t] = j < K ? p.M[(size_t)i * K + j] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        double m = __builtin_inf();
        for (int j = 0; j < K; ++j) { const double v = Msh[(size_t)i * MP + j]; m = v < m ? v : m; }
        rowmin[i] = m;
    }
    __syncthreads();
    // this group's flow values F[i * K + j]
    double *F = FLDS ? rowmin + K + 8 * waves + ((size_t)wave * NG + g) * K * K
                     : p.f_slab + (((size_t)blockIdx.x * waves + wave) * NG + g) * K * K;
    for (int t = c; t < K * K; t += G) F[t] = 0.0;
    emd_wave_sync<FLDS>();

    auto gfield = [&](u64 b) -> unsigned int { return (unsigned int)(b >> gsh) & FM; };
    auto in_mask = [](u64 m) -> bool { return __builtin_amdgcn_inverse_ballot_w64(m); };
    auto bperm_pm = [&](pm_t x, int src) -> pm_t {
        if constexpr (G == 16) return (pm_t)bperm_i32((int)x, src);
        else return (pm_t)(unsigned int)bperm_i32((int)(unsigned int)x, src) | ((pm_t)(unsigned int)bperm_i32((int)(unsigned int)(x >> 32), src) << 32);
    };
    const double INF = __builtin_inf();
    const long total = (long)p.n_rows * N;
    const long n_items = p.upper_only ? (long)p.n_rows * (N - p.row_begin) - (long)p.row_step * p.n_rows * (p.n_rows - 1) / 2 : total;
    auto row_offset = [&](long r) { return r * (N - p.row_begin) - (long)p.row_step * r * (r - 1) / 2; };
    const int q_max_tries = (long)gridDim.x * waves * NG >= EMD_NQ ? 6 : EMD_NQ;
    int qc = (int)((((long)blockIdx.x * waves + wave) * NG + g) % EMD_NQ), q_tries = 0;
    const int aug_guard = 64 * K + 64;

    // ---- state of the group's pair (group-uniform values are replicated in its lanes) ----
    bool alive = true, need_search = true, have_pair = false, tripped = false, rebuild = false;
    int trip_code = 0, n_aug = 0, n_search = 0, step = 0;
    long q = 0;
    double ra = 0.0, rb = 0.0, pu = 0.0, pv = 0.0, tol = 0.0;    // lane c: row c / column c
    unsigned int ship = 0u;                                        // row c ships to these columns
    double A = INF, dC = INF, puN = 0.0, step_bd = 0.0;
    int Apar = 0, parC = 0, parR = -1;
    pm_t pmC = 0, pmR = 0;                                         // path masks of column c / row c
    unsigned int prev_src = 0u, Rf = 0u;
    u64 reachedM = 0ull, closedM = 0ull, demandM = 0ull;           // wave masks: rows reached, columns scanned (and lanes beyond K), columns with demand
#ifdef EMD_MSTAT
    int n_stat = 0;
#endif

    for (;;) {
        if (ballot_b(need_search)) {
            bool srcb = need_search && ra > tol;
            unsigned int srcF = gfield(ballot_b(srcb));
            if (need_search && have_pair && (n_aug > aug_guard || n_search > aug_guard)) { tripped = true; trip_code = 5; }
            if (ballot_b(need_search && (srcF == 0u || tripped))) {
                // (rare: once per pair) pairs that are done leave, new pairs come in -- plain divergent code
                if (need_search) {
                    while (srcF == 0u || tripped) {
                        if (have_pair) {
                            // cost = sum over the support of F_ij * M_ij (lane c walks the bits of row c); the flow block goes back to zeros
                            double cost = 0.0;
                            unsigned int m = ship;
                            while (m) {
                                const int j = __builtin_ctz(m);
                                m &= m - 1u;
                                cost = __builtin_fma(F[c * K + j], Msh[c * MP + j], cost);
                                F[c * K + j] = 0.0;
                            }
                            cost = gsum_f64<G>(cost);
                            if (tripped) for (int t = c; t < K * K; t += G) F[t] = 0.0;
                            if (c == 0) {
                                p.emd[q] = tripped ? __builtin_nan("") : cost;
#ifdef EMD_MSTAT
                                if (p.n_aug) p.n_aug[q] = n_stat;
#else
                                if (p.n_aug) p.n_aug[q] = tripped ? -(n_aug * 8 + trip_code) : n_aug;
#endif
                            }
                            emd_wave_sync<FLDS>();
                            have_pair = false;
                        }
                        // next pair of the group (queue: see emd_grid_kernel)
                        long t = -1;
                        for (;;) {
                            int ti = 0;
                            if (c == 0) ti = __hip_atomic_fetch_add(p.queue + qc * EMD_Q_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ti = bperm_i32(ti, gsh);
                            const long tt = (long)qc + (long)EMD_NQ * ti;
                            if (tt < n_items) { t = tt; q_tries = 0; break; }
                            if (++q_tries >= q_max_tries) break;
                            qc = qc + 1 == EMD_NQ ? 0 : qc + 1;
                        }
                        if (t < 0) { alive = false; need_search = false; ra = 0.0; rb = 0.0; Rf = 0u; break; }
                        int r, j_s;
                        if (p.upper_only) {
                            const double a = 0.5 * p.row_step, b = (double)(N - p.row_begin) + a;      // offset(r) = b r - a r^2
                            const double disc = b * b - 4.0 * a * (double)t;
                            long rr = (long)((b - __builtin_sqrt(disc > 0.0 ? disc : 0.0)) / (2.0 * a));
                            rr = rr < 0 ? 0 : (rr > p.n_rows - 1 ? p.n_rows - 1 : rr);
                            while (rr + 1 < p.n_rows && row_offset(rr + 1) <= t) ++rr;                 // (the float estimate is off by at most one)
                            while (rr > 0 && row_offset(rr) > t) --rr;
                            r = (int)rr;
                            j_s = p.row_begin + r * p.row_step + (int)(t - row_offset(rr));
                        } else {
                            r = (int)(t / N); j_s = (int)(t % N);
                        }
                        q = (long)r * N + j_s;
                        const int i_s = p.row_begin + r * p.row_step;
                        // POT pre-step: b *= sum(a) / sum(b)   (ot/lp/__init__.py::emd2)
                        double a_ = valid ? p.P[(size_t)i_s * K + c] : 0.0, b_ = valid ? p.P[(size_t)j_s * K + c] : 0.0;
                        const double sa = gsum_f64<G>(a_), sb = gsum_f64<G>(b_);
                        tol = 1e-15 * (sa > 0.0 ? sa : 1.0);
                        b_ *= sa / sb;
                        pv = 0.0;
                        pu = valid ? rowmin[c] : 0.0;        // pu_i = min_j M_ij keeps every reduced cost >= 0 at the start
                        ship = 0u;
                        // warm start: min(a_i, b_i) along every diagonal arc of zero reduced cost (see emd_grid_kernel)
                        if (valid && Msh[c * MP + c] - pu == 0.0) {
                            const double f = a_ < b_ ? a_ : b_;
                            if (f > 0.0) { F[c * K + c] = f; ship = 1u << c; a_ -= f; b_ -= f; }
                        }
                        emd_wave_sync<FLDS>();
                        ra = a_; rb = b_;
                        prev_src = 0u; n_aug = 0; n_search = 0; tripped = false; trip_code = 0; have_pair = true;
#ifdef EMD_MSTAT
                        n_stat = 0;
#endif
                        srcF = gfield(ballot_b(ra > tol));
                    }
                }
                if (!ballot_b(alive)) break;
                srcb = need_search && ra > tol;
            }
            // a new search: shortest paths from ANY row with supply left.  The initial label of column j is min over the source
            // rows i of M_ij - pu_i (a source's potential never moves), kept in A while the source set stands.
            const bool bs = need_search;
            const u64 bsM = ballot_b(bs);
            rebuild = bs ? srcF != prev_src : rebuild;
            prev_src = bs ? srcF : prev_src;
            reachedM = (reachedM & ~bsM) | ballot_b(srcb);
            closedM = (closedM & ~bsM) | (~validM & bsM);
            demandM = (demandM & ~bsM) | ballot_b(bs && rb > 0.0);
            parR = bs ? -1 : parR;
            puN = bs ? pu : puN;
            pmR = bs ? rowbit : pmR;
            step_bd = bs ? 0.0 : step_bd;
            step = bs ? 0 : step;
            dC = bs ? (rebuild ? INF : A) : dC;
            parC = bs ? (rebuild ? 0 : Apar) : parC;
            pmC = bs ? (colbit | ((pm_t)1 << (G + Apar))) : pmC;       // (a rebuilt label gets its mask with its relaxation)
            Rf = bs ? (rebuild ? srcF : 0u) : Rf;
            need_search = false;
#if defined(EMD_MSTAT) && EMD_MSTAT == 4
            if (bs) ++n_stat;
#endif
        }
        // rows reached in the previous step (or, when the label cache is rebuilt, the sources): forward arcs to every open column
        while (ballot_b(Rf != 0u)) {
            const bool on = Rf != 0u;
#if defined(EMD_MSTAT)
            if (EMD_MSTAT == 2 && on && !rebuild) ++n_stat;
            if (EMD_MSTAT == 5 && on && rebuild) ++n_stat;
#endif
            const int i = on ? __builtin_ctz(Rf) : 0;              // (row 0 where the group has no row left: a valid address, selected away)
            Rf &= Rf - 1u;
            const double t_i = bperm_f64(puN, gsh + i);                  // pu_i - d_i
            const pm_t pm_i = bperm_pm(pmR, gsh + i);
            const double nd = Msh[i * MP + c] - t_i;
            const bool lt = on && nd < dC && !in_mask(closedM);
            dC = lt ? nd : dC;
            parC = lt ? i : parC;
            pmC = lt ? (pm_i | colbit) : pmC;
        }
        A = rebuild ? dC : A;
        Apar = rebuild ? parC : Apar;
        rebuild = false;
        step += alive ? 1 : 0;
        bool trip_now = alive && step > 2 * K + 2;          // cannot happen: a step closes >= one column
#if defined(EMD_MSTAT) && EMD_MSTAT == 1
        if (alive) ++n_stat;
#endif
        // smallest open label; ALL columns that carry it are final and are scanned in this one step
        union { double d; unsigned int u[2]; } cur, bdu;
        cur.d = __builtin_fmax(dC - pv, step_bd);
        cur.u[1] = in_mask(closedM) ? 0x7ff00000u : cur.u[1];         // closed: never the minimum
        const double bd = gmin_f64<G>(cur.d);
        bdu.d = bd;
        const bool exh = alive && bdu.u[1] >= 0x7ff00000u;              // nothing (more) reachable: only rounding dust is left
        ra = exh ? 0.0 : ra;
        const bool go = alive && !exh && !trip_now;
        step_bd = go ? bd : step_bd;
        const u64 tieM = ballot_b(go && cur.u[1] == bdu.u[1] && cur.u[0] == bdu.u[0]);
        closedM |= tieM;
        const unsigned int tieF = gfield(tieM);
        // tied columns with demand left are targets: augment right away, without touching the potentials, and go on while the tree
        // is valid (its root still has supply, no arc on a path ran empty, the target is satisfied)
        bool ended = false;
        if (tieM & demandM) {
            unsigned int tgtF = gfield(tieM & demandM);
            while (ballot_b(tgtF != 0u && !ended)) {
                const bool aug = tgtF != 0u && !ended;
                const int t = __builtin_ctz(tgtF | (1u << (G - 1)));
                tgtF = aug ? tgtF & (tgtF - 1u) : tgtF;
                const pm_t pm_t_ = bperm_pm(pmC, gsh + t);
                const bool onC = aug && ((pm_t_ >> c) & 1u), onR = aug && ((pm_t_ >> (G + c)) & 1u);
                const bool is_t = aug && c == t;
                if (onC) scr[gsh + parC] = (unsigned char)c;          // "your forward arc goes to me"
                emd_wave_sync<true>();                                 // (other lanes of the wave read what this one stored)
                const int fwd = scr[lane], bwd = parR;                 // (bwd < 0: the source row of the path)
                double fb = INF;
                if (onR && bwd >= 0) fb = F[c * K + bwd];
                double val = onR ? (bwd >= 0 ? fb : ra) : INF;
                val = is_t && rb < val ? rb : val;
                const double delta = gmin_f64<G>(val);               // min(demand of the target, supply of the root, backward-arc flows)
                if (onR) {
                    F[c * K + fwd] += delta;
                    if (bwd >= 0) F[c * K + bwd] = fb - delta;
                }
                emd_wave_sync<FLDS>();
                const bool emp = onR && bwd >= 0 && fb == delta;      // a backward arc ran empty: it leaves the support
                ship = onR ? ((ship | (1u << fwd)) & ~(emp ? 1u << bwd : 0u)) : ship;
                const bool isrc = onR && bwd < 0;
                ra = isrc ? ra - delta : ra;
                rb = is_t ? rb - delta : rb;
                // a dry root, an emptied arc, or a target that keeps demand (its path gave out first): the tree is out of date
                const bool stale = emp || (isrc && !(ra > tol)) || (is_t && rb > 0.0);
                n_aug += aug ? 1 : 0;
                demandM = (demandM & ~ballot_b(aug)) | ballot_b(aug && rb > 0.0);
                ended = ended || (aug && gfield(ballot_b(stale)) != 0u);
            }
        }
        if (trip_now) { tripped = true; trip_code = 1; }
        const bool endS = go && ended;
        // potentials: rc'(i,j) = rc(i,j) + min(d_i, d*) - min(d_j, d*) >= 0 with d* = this step's label, 0 on the tree
        {
            const double fC = in_mask(closedM) && valid ? __builtin_fmax(dC - pv, 0.0) : INF;
            const double puE = in_mask(reachedM) ? puN : pu - bd;
            pu = endS ? puE : pu;
            pv = endS ? pv + __builtin_fmin(fC, bd) : pv;
            n_search += endS ? 1 : 0;
        }
        need_search = alive && (endS || exh || trip_now);
        // backward arcs: the rows that ship to ANY of the tied columns (reduced cost 0) are reached with this step's label
        const bool cont = go && !ended;
        const unsigned int hit = ship & tieF;
        const u64 newM = ballot_b(cont && hit != 0u) & ~reachedM;
        reachedM |= newM;
        const bool newr = in_mask(newM);
        const int pr = __builtin_ctz(hit | (1u << (G - 1)));
        const pm_t pm_par = bperm_pm(pmC, gsh + pr);
        parR = newr ? pr : parR;
        puN = newr ? pu - bd : puN;
        pmR = newr ? (pm_par | rowbit) : pmR;
        Rf = cont ? gfield(newM) : 0u;
    }
}

// host side: geometry of a launch
struct EmdMultiGeom { int G, waves, wgs_per_cu; bool flds; size_t lds; };
inline EmdMultiGeom emd_multi_geom(int K, bool flds_wanted) {
    EmdMultiGeom m;
    m.G = 16;
    m.flds = flds_wanted;
    // the workgroup size that puts most waves on a CU (160 KB of LDS, 32 waves); ties: the larger workgroup (one copy of M serves more)
    int best_w = 1, best_total = 0, best_wg = 1;
    for (int w = 1; w <= 8; ++w) {
        const size_t lds = emd_multi_lds_bytes(K, m.G, w, m.flds);
        int wg = (int)((size_t)160 * 1024 / lds);
        if (wg * w > 32) wg = 32 / w;
        if (wg < 1) continue;
        if (wg * w >= best_total) { best_total = wg * w; best_w = w; best_wg = wg; }
    }
    m.waves = best_w; m.wgs_per_cu = best_wg;
    m.lds = emd_multi_lds_bytes(K, m.G, m.waves, m.flds);
    return m;
}

}  // namespace pilot
