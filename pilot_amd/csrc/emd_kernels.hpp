// Exact optimal-transport pair grid for gfx950: one wavefront per ordered pair.
// Replaces the ot.emd2 loop of pilotpy/tools/Trajectory.py:507-511 (the reference's DEFAULT mode).
//
// POT solves each K x K transportation LP with a LEMON-derived network simplex on the CPU.  The LP
// optimum VALUE is unique, so any exact algorithm returns the same number up to rounding; on the
// GPU the problem is solved by successive shortest augmenting paths (multi-source: from any row with supply left
// to any column with demand left) with node potentials
// (complementary slackness is kept after every augmentation, so the final flow is optimal):
//   * lane l owns row l and column l (and l + 64, l + 128, l + 192 for K up to 256): supplies/demands, potentials,
//     Dijkstra labels and predecessor links live in registers;
//   * one Dijkstra step = wave-wide minimum over the open column labels (a six-stage v_min_u32_dpp scan) and ONE
//     parallel relaxation per scanned row: it relaxes all K columns at once (row of M from LDS, coalesced); a scanned
//     column reaches the rows that currently ship to it (one AND against the support masks), and they are scanned in the
//     same step -- every node tied at the minimum is handled in that one step;
//   * a target is augmented at once, without touching the potentials, and the search goes on while its tree is valid
//     (several augmentations per search); the potentials are brought up to date when a new search starts;
//   * flow values in an L2-resident global slab per wave, flow support as bit masks in registers (see below);
//   * path tracing / bottleneck / flow update walk the predecessor links with wave-uniform indices
//     (v_readlane / v_writelane), touching one flow entry per hop;
//   * pairs are drawn from a device-wide counter over the SOLVED pairs (upper triangle only for a symmetric cost).
// All arithmetic is fp64 like POT's.
#pragma once
#include <hip/hip_runtime.h>

namespace pilot {


struct EmdParams {
    const double *P;    // N x K
    const double *M;    // K x K
    int N, K;
    int n_rows, row_begin, row_step;
    int upper_only;     // 1: only pairs with j >= i are solved (symmetric cost); others left untouched
    double *emd;        // n_rows x N
    int *n_aug;         // nullable: augmentations per pair (diagnostic; negative = guard tripped)
    double *f_slab;     // global flow slabs (one K*K block per resident wave) when !F_IN_LDS
    int *queue;         // dynamic pair queue (zeroed before the launch)
};

// the lanes where `b` holds, as a wave-uniform mask (HIP's __ballot takes an int: the bool is first materialised as 0 / 1 in a
// register and compared again -- two vector instructions per call in a kernel that ballots three times per Dijkstra step)
#ifndef PILOT_BALLOT_B_DEFINED
#define PILOT_BALLOT_B_DEFINED
__device__ inline unsigned long long ballot_b(bool b) { return __builtin_amdgcn_ballot_w64(b); }
#endif
__device__ inline double rl_f64(double x, int lane) {
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return r.d;
}
__device__ inline int rl_i32(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
// make a value that IS wave-uniform also LOOK uniform to the compiler (scalar branches, no exec-mask loops)
__device__ inline double uni_f64(double x) {
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    r.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return r.d;
}
__device__ inline int uni_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Wave-wide reductions of a double, every lane gets the result.  Lanes are exchanged with DPP inside a 16-lane row
// (quad_perm for the xor-1 / xor-2 partners, row_half_mirror and row_mirror to pair quads and octets) and with
// gfx950's v_permlane16_swap / v_permlane32_swap across rows: ~20 VALU instructions instead of six LDS-crossbar
// (ds_bpermute) round trips -- the arg-min of every Dijkstra step sits on the critical path of the solver.
template <int CTRL> __device__ inline double dpp_f64(double x) {
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(u.i[0], u.i[0], CTRL, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(u.i[1], u.i[1], CTRL, 0xf, 0xf, false);
    return r.d;
}
__device__ inline void swap16_f64(double x, double &a, double &b) {
    union { double d; unsigned int u[2]; } v, p, q;
    v.d = x;
    auto lo = __builtin_amdgcn_permlane16_swap(v.u[0], v.u[0], false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(v.u[1], v.u[1], false, false);
    p.u[0] = lo[0]; p.u[1] = hi[0]; q.u[0] = lo[1]; q.u[1] = hi[1];
    a = p.d; b = q.d;
}
__device__ inline void swap32_f64(double x, double &a, double &b) {
    union { double d; unsigned int u[2]; } v, p, q;
    v.d = x;
    auto lo = __builtin_amdgcn_permlane32_swap(v.u[0], v.u[0], false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(v.u[1], v.u[1], false, false);
    p.u[0] = lo[0]; p.u[1] = hi[0]; q.u[0] = lo[1]; q.u[1] = hi[1];
    a = p.d; b = q.d;
}
// x of the lane the DPP control selects, ~0u (the identity of an unsigned minimum) where it selects none: in this form the
// compiler folds the move into the consumer (one v_min_u32_dpp per stage)
template <int CTRL, int ROW_MASK = 0xf> __device__ inline unsigned int dpp_u32(unsigned int x) {
    return (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, ROW_MASK, 0xf, false);
}
// wave-wide unsigned minimum (wave-uniform result): row_shr 1 / 2 / 4 / 8 scan inside each 16-lane row, row_bcast:15 and
// row_bcast:31 across the rows -- six v_min_u32_dpp; lane 63 ends up with the minimum of all 64
__device__ inline unsigned int wave_min_u32(unsigned int x) {
    unsigned int y;
    y = dpp_u32<0x111>(x); x = y < x ? y : x;
    y = dpp_u32<0x112>(x); x = y < x ? y : x;
    y = dpp_u32<0x114>(x); x = y < x ? y : x;
    y = dpp_u32<0x118>(x); x = y < x ? y : x;
    y = dpp_u32<0x142, 0xa>(x); x = y < x ? y : x;
    y = dpp_u32<0x143, 0xc>(x); x = y < x ? y : x;
    return (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
}
// Wave-wide minimum of doubles whose bit patterns order like unsigned integers (non-negative values, +inf included;
// "done" entries are -1.0, whose pattern is above every non-negative one): the minimum is the lexicographic minimum of
// (high word, low word).  The high words take one 32-bit reduction with the DPP modifier fused into v_min_u32; the low
// word of the winner is then read straight from its lane when every lane that shares the minimal high word also shares
// the low word (a unique minimum, or exact ties -- the usual cases: labels of a Dijkstra step either differ in their
// leading 32 bits or are the same number); only otherwise a second reduction runs.
__device__ inline double wave_min_f64(double x) {
    union { double d; unsigned int u[2]; } v, o;
    v.d = x;
    const unsigned int mh = wave_min_u32(v.u[1]);
    const unsigned long long top = ballot_b(v.u[1] == mh);
    unsigned int ml = (unsigned int)__builtin_amdgcn_readlane((int)v.u[0], __builtin_ctzll(top));
    if (ballot_b(v.u[1] == mh && v.u[0] != ml)) ml = wave_min_u32(v.u[1] == mh ? v.u[0] : 0xffffffffu);   // wave-uniform
    o.u[0] = ml; o.u[1] = mh;
    return o.d;
}
__device__ inline double wave_sum_f64(double x) {
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    double a, b;
    swap16_f64(x, a, b); x = a + b;
    swap32_f64(x, a, b); x = a + b;
    return x;
}

// BF (NK >= 3, or 1 with padded rows): the loops over a row of M -- the relaxation, the rebuild of the source minima -- read
// without a bounds test (a lane beyond K reads column K - 1; its label is closed for good, its minimum never used) and write
// through selects.  K = 160: 333 -> 246 ms per 600 x 600 grid.  Not for NK = 2: at its 64 registers the two clamped column
// indices cost more spills than the branches they remove (K = 100: 36.1 -> 41.6 ms), profiles/r04/ab_experiments.md.
constexpr int EMD_BF_MIN_NK = 3;
// UL (template parameter; EMD_ULAB = 0 / 1 forces it off / on for A/B builds, default: by K, see emd_ul()): a column's label is
// kept PLUS its potential, L_j = d_j + pv_j = min over the scanned rows i of M_ij + (d_i - pu_i): relaxing a row is one add and one
// compare per column (was: two subtracts, a clamp, an add, a compare), and the label itself, max(L_j - pv_j, 0), is formed once
// per step for the arg-min.  The scanned columns are a wave-uniform bit mask (lane = column), so closing a column costs no
// vector instruction and the final labels are read off L once per search.  It pays when a step relaxes rows over many
// columns: K = 50 -2.4 %, K = 100 -9 %; the 14 clusters of the kidney cohort +4.5 % (profiles/r04/ab_experiments.md)
__host__ __device__ constexpr bool emd_ul(int K) {
#ifdef EMD_ULAB
    return EMD_ULAB != 0;
#else
    return K > 32;
#endif
}
// lane `lane` of `old` <- the wave-uniform `value` (v_writelane_b32: one instruction instead of compare + select; gfx950
// reads at most one SGPR per VALU instruction, so the lane select travels in M0)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // M0 is reserved (never live across statements); the clobber is declared anyway
__device__ inline int wl_i32(int old, int value, int lane) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(value), "s"(lane) : "m0");
    return old;
}
#pragma clang diagnostic pop
// m with bit `bit` cleared, in ONE scalar instruction (m &= m - 1 is s_add_u32 + s_addc_u32 + s_and_b64; the scalar unit is as
// busy as the vector unit in the exact-OT kernel, profiles/r04/rocprofv3_pmc_summary_emd_c3.txt)
__device__ inline unsigned long long clear_bit(unsigned long long m, int bit) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));
    return m;
}
__device__ inline unsigned int hi_word(double x) {
    union { double d; unsigned int u[2]; } v;
    v.d = x;
    return v.u[1];
}
// order of the bit patterns (see wave_min_f64)
__device__ inline bool bits_less(double a, double b) {
    union { double d; unsigned long long u; } x, y;
    x.d = a; y.d = b;
    return x.u < y.u;
}


// NK = rows/columns per lane (1: K <= 64, 2: K <= 128, 3 / 4: K <= 192 / 256 with MG).
//   * the cost matrix M and its row minima live in LDS (shared by the workgroup);
//   * the flow VALUES live in an L2-resident global slab of K*K doubles per resident wave (row-major, zero outside the support)
//     and are touched only along augmenting paths and for the final cost;
//   * the flow SUPPORT lives in registers: lane i keeps a bit mask of the columns row i currently ships to, so "which
//     rows ship to the columns being scanned" is one 64-bit AND against the ballot mask of those columns.
// Nothing per wave is in LDS, so occupancy is bounded by registers (and, from K = 91 on, by the LDS copy of M).
// (Round 4 also built a variant with the flow values in LDS slots, K <= 60: a seventh of the HBM traffic and 5 - 7 % MORE time on
// this instruction-bound kernel -- removed in round 5, profiles/r04/ab_experiments.md #11; the small-K kernel of
// emd_multi_kernels.hpp keeps its flows in LDS.)
// waves per workgroup: 8 (two per SIMD) in general; 16 for 64 < K <= 128, where the LDS copy of M (up to 128 KB) allows
// at most two workgroups per CU -- with 8 waves each that is 4 waves per SIMD, too few to hide the latencies of this
// kernel; 2 x 16 waves of <= 64 registers run c4 in 0.47 s instead of 0.55 s (profiles/r03/ab_experiments.md)
__host__ __device__ constexpr int emd_waves(int NK) { return NK == 2 ? 16 : 8; }
// Row pitch of the LDS copy of M (doubles).  K <= 64: 64, one lane per column of a padded row -- the relaxation of a row reads
// its 64 entries without a bounds test (the pad columns' labels are closed for good) and addresses it by a shift: two scalar
// instructions and a branch less per relaxed row in a kernel whose scalar unit is as busy as its vector unit.
__host__ __device__ constexpr int emd_m_pitch(int K) { return K <= 64 ? 64 : K; }
// dynamic LDS of emd_grid_kernel: M (K <= 128), the row minima, and for K <= 64 the two K x K byte tables of the source order
__host__ __device__ constexpr size_t emd_lds_bytes(int K) {
    return K > 128 ? sizeof(double) * (size_t)K
                   : sizeof(double) * ((size_t)K * emd_m_pitch(K) + K) + (K <= 64 ? (size_t)K * K + (size_t)K * emd_m_pitch(K) : 0);
}
#define EMD_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
// Build switches (diagnostics and experiments only):
//   EMD_ULAB = 0 / 1   forces the label form (see emd_ul) for A/B builds
//   EMD_PROF = n       n_aug reports the wave's clock64() ticks / 16 spent in section n of a pair: 1 whole pair, 2 label set-up of a
//                      search, 3 arg-min, 4 ties + targets (augmentations included), 5 augmentations alone, 6 row reach +
//                      relaxations, 7 final cost, 8 pair set-up, 9 potentials
//   EMD_STAT = n       n_aug reports a count: 1 Dijkstra steps, 2 tied-row relaxations, 3 path hops, 4 source rows in A rebuilds, 5 searches
//   EMD_PERTURB = eps  robustness experiment: every reduced cost below 1e-13 -- the tight arcs -- is replaced by eps, i.e. the
//                      labels that tie at a step's minimum no longer do; the LP values must not move (profiles/r04/ab_experiments.md #9)
//   EMD_NO_CLOCK       no wall-clock watchdog (timing experiment)
#ifdef EMD_PROF
#define PROF_BEGIN(n) if constexpr (EMD_PROF == (n)) { prof_t0 = clock64(); }
#define PROF_END(n) if constexpr (EMD_PROF == (n)) { prof_acc += clock64() - prof_t0; }
#else
#define PROF_BEGIN(n) do {} while (0)
#define PROF_END(n) do {} while (0)
#endif
// dynamic pair queue: EMD_NQ counters, EMD_Q_STRIDE ints apart (one 128-byte line each)
constexpr int EMD_NQ = 64, EMD_Q_STRIDE = 32;
#define EMD_WPE_ATTR __attribute__((amdgpu_waves_per_eu(NK == 1 ? 8 : (NK == 2 ? 8 : 2), 8)))

// MG: the cost matrix is read from global memory (L2) instead of LDS -- K > 128, where K*K doubles no longer fit LDS
template <int NK, bool MG = false, bool UL = true>
__global__ void __launch_bounds__(64 * emd_waves(NK)) EMD_WPE_ATTR emd_grid_kernel(EmdParams p) {
    constexpr int EMD_WAVES = emd_waves(NK);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K, N = p.N;
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    double *Msh = reinterpret_cast<double *>(smem_raw);   // K*K (not with MG)
    constexpr bool PAD = NK == 1 && !MG;                  // rows of M padded to 64 entries (zeros)
    const int MP = PAD ? 64 : K;                          // row pitch of Mrd
    double *rowmin = MG ? Msh : Msh + (size_t)K * MP;     // K: min_j M_ij (initial row potentials)
    if constexpr (!MG) {
        for (int t = threadIdx.x; t < K * MP; t += blockDim.x) {
            const int i = t / MP, j = t % MP;
            Msh[t] = j < K ? p.M[(size_t)i * K + j] : 0.0;
        }
        __syncthreads();
    }
    const double *Mrd = MG ? p.M : Msh;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        double m = __builtin_inf();
        for (int j = 0; j < K; ++j) { const double v = Mrd[(size_t)i * MP + j]; m = v < m ? v : m; }
        rowmin[i] = m;
    }
    __syncthreads();
    // ORD (K <= 64): the initial label of column j is min over the SOURCE rows i of c_ij = M_ij - pu_i, and a source's pu_i never
    // moves (its distance is 0 in every search; rows only ever stop being sources), so c is the same matrix for every pair
    // and every search: M - rowmin.  Each column's rows are ranked by c_ij once per workgroup (ties: lower row first, the
    // order the rebuild loop finds them in): a column's label source is its source row of smallest RANK -- one byte read and
    // an integer minimum per source row instead of a readlane pair, a row read, an fp64 subtract, an fp64 compare and three
    // selects (315 source visits per c3 pair).
    constexpr bool ORD = NK == 1 && !MG;
    unsigned char *ord = reinterpret_cast<unsigned char *>(rowmin + K);      // [column][rank] -> row
    unsigned char *rnk = ord + (size_t)K * K;                                // [row][column] -> rank of the row in that column
                                                                             // (lanes = columns read consecutive bytes: no bank conflict)
    if constexpr (ORD) {
        for (int t = threadIdx.x; t < K * K; t += blockDim.x) {
            const int i = t / K, j = t % K;
            const double c = Msh[(size_t)i * MP + j] - rowmin[i];
            int rank = 0;
            for (int i2 = 0; i2 < K; ++i2) {
                const double c2 = Msh[(size_t)i2 * MP + j] - rowmin[i2];
                rank += (c2 < c || (c2 == c && i2 < i)) ? 1 : 0;
            }
            ord[j * K + rank] = (unsigned char)i;
            rnk[i * MP + j] = (unsigned char)rank;
        }
        __syncthreads();
    }
    double *F = p.f_slab + ((size_t)blockIdx.x * EMD_WAVES + wave) * K * K;   // F[i*K + j]
    constexpr bool BF = NK >= EMD_BF_MIN_NK;
    int mcol[NK];                                                             // BF: the lane's columns, clamped into the matrix
#pragma unroll
    for (int e = 0; e < NK; ++e) mcol[e] = PAD || lane + 64 * e < K ? lane + 64 * e : K - 1;
    const double INF = __builtin_inf(), NEG = -1.0;
    // LAZY: keep a search going after an augmentation dried its root / emptied an arc / left its target open, and restart
    // only when a later path turns out to be unusable.  At K = 50 the searches drop from 41 to 22 per pair but the steps do
    // not, and the extra state costs registers (10.5 -> 11.7 ms); at K = 100 it paid at 4 waves per SIMD (0.73 -> 0.66 s) but
    // the eager form at 8 waves per SIMD and <= 64 registers is faster still (0.47 s): kept for the K > 128 variants only
    constexpr bool LAZY = NK >= 3;
    const long total = (long)p.n_rows * N;

    // Waves draw pairs from one device-wide counter.  (A static deal leaves the waves with very different numbers of SOLVED
    // pairs when only the upper triangle is solved -- about 22 +- 5 of a wave's 44 at c3 -- and the launch ends with its
    // unluckiest wave.)  With upper_only the counter runs over the solved pairs alone: local row r = rows row_begin +
    // r row_step holds the N - i_s pairs j >= i_s, so item t sits in the row with offset(r) <= t < offset(r + 1),
    // offset(r) = r (N - row_begin) - row_step r (r - 1) / 2.  Rows are drawn in order: the long rows start first.
    const long n_items = p.upper_only ? (long)p.n_rows * (N - p.row_begin) - (long)p.row_step * p.n_rows * (p.n_rows - 1) / 2 : total;
    auto row_offset = [&](long r) { return r * (N - p.row_begin) - (long)p.row_step * r * (r - 1) / 2; };
    // (a wave's first item is its own number -- no burst of atomics on one address at the start of a small grid --, the
    // following ones come from the counter, which starts behind the resident waves)
    // The flow slab holds zeros outside the support of the pair being solved: zeroed ONCE per wave here, and a finished pair
    // puts back zeros in the (<= 2K - 1) entries of its final support while it sums its cost.  (Round 3 zeroed the K x K slab
    // per pair: 20 KB written and later fetched again per c3 pair -- 7.5 GB of HBM traffic per launch for 0.29 GB of inputs
    // and outputs, profiles/r03/rocprofv3_pmc_summary_emd_c3.txt.)
    for (int t = lane; t < K * K; t += 64) F[t] = 0.0;
    EMD_FENCE();
    __builtin_amdgcn_wave_barrier();
    // Pairs are drawn from EMD_NQ device-wide counters, each on a cache line of its own: counter c hands out the items
    // c, c + EMD_NQ, c + 2 EMD_NQ, ..  One counter for all was 11.4 ns of one L2 atomic unit per pair whatever K is -- ALL of the
    // 2.06 ms of a 600 x 600 grid at K <= 16 (profiles/r04/small_k_probe_before.txt); drawing batches from it instead left
    // the launch waiting for the waves with the longest batches (K = 50: +5 %, the 634-patient kidney cohort +13 %).  A wave
    // starts at the counter of its own number and moves on to the next one when a counter runs out, so the items still go
    // out one by one, in increasing order per counter (the long rows of the upper triangle first).  Every counter has waves
    // that start on it and drain it, so a wave may give up after a few empty neighbours (all counters run out at about the
    // same time: a full round of 64 atomics per wave at the end would cost more than the queue itself); a launch with
    // fewer waves than counters makes the full round.
    const int q_max_tries = (int)gridDim.x * EMD_WAVES >= EMD_NQ ? 6 : EMD_NQ;
    int qc = ((int)blockIdx.x * EMD_WAVES + wave) % EMD_NQ, q_tries = 0;
    for (;;) {
        int ti = 0;
        if (lane == 0) ti = __hip_atomic_fetch_add(p.queue + qc * EMD_Q_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long t = (long)qc + (long)EMD_NQ * uni_i32(ti);
        if (t >= n_items) {
            if (++q_tries >= q_max_tries) break;
            qc = qc + 1 == EMD_NQ ? 0 : qc + 1;
            continue;
        }
        q_tries = 0;
        int r, j_s;
        if (p.upper_only) {
            const double a = 0.5 * p.row_step, b = (double)(N - p.row_begin) + a;      // offset(r) = b r - a r^2
            const double disc = b * b - 4.0 * a * (double)t;
            long rr = (long)((b - __builtin_sqrt(disc > 0.0 ? disc : 0.0)) / (2.0 * a));
            rr = rr < 0 ? 0 : (rr > p.n_rows - 1 ? p.n_rows - 1 : rr);
            while (rr + 1 < p.n_rows && row_offset(rr + 1) <= t) ++rr;                 // (the float estimate is off by at most one)
            while (rr > 0 && row_offset(rr) > t) --rr;
            r = uni_i32((int)rr);
            j_s = uni_i32(p.row_begin + r * p.row_step + (int)(t - row_offset(rr)));
        } else {
            r = (int)(t / N); j_s = (int)(t % N);
        }
        const long q = (long)r * N + j_s;
        const int i_s = p.row_begin + r * p.row_step;
#ifdef EMD_PROF
        long long prof_t0 = 0, prof_acc = 0;
#endif
        PROF_BEGIN(1); PROF_BEGIN(8);

        double pu[NK], pv[NK], ra[NK], rb[NK], dC[NK];
        int parR[NK], parC[NK];
        unsigned long long ship[NK][NK];           // ship[e][w] bit b: row (lane + 64e) ships to column 64w + b
        // POT pre-step: b *= sum(a) / sum(b)   (ot/lp/__init__.py::emd2)
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int e = 0; e < NK; ++e) {
            const int idx = lane + 64 * e;
            ra[e] = idx < K ? p.P[(size_t)i_s * K + idx] : 0.0;
            rb[e] = idx < K ? p.P[(size_t)j_s * K + idx] : 0.0;
            sa += ra[e]; sb += rb[e];
        }
        sa = uni_f64(wave_sum_f64(sa)); sb = uni_f64(wave_sum_f64(sb));
        const double scale = sa / sb;
        const double tol = 1e-15 * (sa > 0.0 ? sa : 1.0);
        // warm start: wherever the diagonal arc (i, i) has zero reduced cost (always, for a metric-like cost with a
        // zero diagonal) ship min(a_i, b_i) along it.  Flow only on zero-reduced-cost arcs keeps complementary
        // slackness, so the augmenting-path phase continues from an optimal partial flow and only has to move the
        // mass that really differs between the two histograms (about 3x fewer augmentations on PILOT-like inputs).
#pragma unroll
        for (int e = 0; e < NK; ++e) {
            const int idx = lane + 64 * e;
            rb[e] *= scale;
            pv[e] = 0.0;
            pu[e] = idx < K ? rowmin[idx] : 0.0;   // pu_i = min_j M_ij keeps every reduced cost >= 0 at the start
#pragma unroll
            for (int w = 0; w < NK; ++w) ship[e][w] = 0ull;
            if (idx < K && Mrd[(size_t)idx * MP + idx] - pu[e] == 0.0) {
                const double f = ra[e] < rb[e] ? ra[e] : rb[e];
                if (f > 0.0) {
                    F[(size_t)idx * K + idx] = f;
                    ship[e][e] = 1ull << lane; ra[e] -= f; rb[e] -= f;
                }
            }
        }
        EMD_FENCE();
        __builtin_amdgcn_wave_barrier();
        PROF_END(8);
        int n_aug = 0, n_search = 0;
#ifdef EMD_STAT     // diagnostic builds: 1 Dijkstra steps, 2 tied-row relaxations, 3 path hops, 4 source rows in A rebuilds, 5 searches
        int n_stat = 0;
#endif
        const int aug_guard = 64 * K + 64;   // far above the O(K) augmentations SSP needs; bounds every loop
        bool tripped = false;
        int trip_code = 0;
#ifdef EMD_NO_CLOCK
        const unsigned long long t_start = 0;
#else
        const unsigned long long t_start = wall_clock64();
#endif
        const unsigned long long watchdog_ticks = 400000000ull;  // 4 s of the 100 MHz constant clock per pair

        // One search per round: shortest paths from ANY row that still has supply to the columns that still have demand
        // (multi-source Dijkstra on the reduced costs), with as many augmentations as the tree survives.  All source rows
        // are relaxed up front without an arg-min each, and every step scans ALL nodes tied at the smallest label.
        unsigned long long prev_src[NK];
        double A[NK];
        int Apar[NK];
#pragma unroll
        for (int e = 0; e < NK; ++e) { prev_src[e] = 0ull; A[e] = INF; Apar[e] = -1; }
        for (;;) {
            unsigned long long srcmask[NK];
            bool any_src = false;
#pragma unroll
            for (int e = 0; e < NK; ++e) {
                srcmask[e] = ballot_b(ra[e] > tol);                 // (ra is 0 beyond K)
                any_src = any_src || srcmask[e] != 0ull;
            }
            if (!any_src) break;
            if (n_aug > aug_guard || n_search > aug_guard) { tripped = true; trip_code = 5; break; }
#ifndef EMD_NO_CLOCK
            if ((n_search & 15) == 15 && wall_clock64() - t_start > watchdog_ticks) { tripped = true; trip_code = 6; break; }
#endif
            // initial column labels min over sources i of rc(i, j) = (A_j - pv_j)+ with A_j = min_i (M_ij - pu_i): a source's
            // potential never moves (its distance is 0), so A and its arg-min only change when a source runs dry
            PROF_BEGIN(2);
            bool src_changed = false;
#pragma unroll
            for (int e = 0; e < NK; ++e) { src_changed = src_changed || srcmask[e] != prev_src[e]; prev_src[e] = srcmask[e]; }
            if constexpr (ORD) {
                if (src_changed) {
                    const unsigned long long sm = srcmask[0];
                    if (ballot_b(lane < K && (Apar[0] < 0 || !((sm >> Apar[0]) & 1ull)))) {     // some column lost its source
                        unsigned int best = 0xffu;
                        const unsigned char *rl = rnk + (lane < K ? lane : 0);
                        unsigned long long m = sm;
                        while (m) {                                        // wave-uniform
#if defined(EMD_STAT) && EMD_STAT == 4
                            ++n_stat;
#endif
                            const int i = __builtin_ctzll(m);
                            m = clear_bit(m, i);
                            const unsigned int r = rl[i * MP];
                            best = r < best ? r : best;
                        }
                        if (lane < K) {
                            Apar[0] = ord[lane * K + best];
                            A[0] = Msh[(size_t)Apar[0] * MP + lane] - rowmin[Apar[0]];
                        }
                    }
                }
            } else if (src_changed) {
#pragma unroll
                for (int e = 0; e < NK; ++e) { A[e] = INF; Apar[e] = -1; }
#pragma unroll
                for (int e = 0; e < NK; ++e) {
                    unsigned long long m = srcmask[e];
                    while (m) {                                        // wave-uniform
#if defined(EMD_STAT) && EMD_STAT == 4
                        ++n_stat;
#endif
                        const int l = __builtin_ctzll(m);
                        m = clear_bit(m, l);
                        const int in = l + 64 * e;
                        const double pu_i = rl_f64(pu[e], l);
#pragma unroll
                        for (int e2 = 0; e2 < NK; ++e2) {
                            if constexpr (BF) {
                                const double v = Mrd[(size_t)in * MP + mcol[e2]] - pu_i;
                                const bool lt = __builtin_amdgcn_inverse_ballot_w64(ballot_b(v < A[e2]));
                                A[e2] = lt ? v : A[e2];
                                Apar[e2] = lt ? in : Apar[e2];
                            } else {
                                const int idx = lane + 64 * e2;
                                if (idx < K) {
                                    const double v = Mrd[(size_t)in * MP + idx] - pu_i;
                                    if (v < A[e2]) { A[e2] = v; Apar[e2] = in; }
                                }
                            }
                        }
                    }
                }
            }
            // Labels.  Columns, !UL: dC is the tentative distance while the column is open and NEG (-1) once it is scanned (or
            // beyond K) -- as bit patterns NEG sorts above every distance and above +inf, so the arg-min, the tie test and
            // the relaxation need no "scanned" flag; fC keeps the final distance (+inf: never scanned).  UL: dC is distance +
            // column potential and closedm the scanned columns (see emd_ul above).  Rows are never pending: a row is reached
            // only over a zero-reduced-cost backward arc from a column being scanned, takes that column's label d_i and is
            // scanned in the same step.  reachedm = the rows reached so far (sources included), a wave-uniform mask (lane =
            // row); puN = the potential a reached row takes when the search ends, pu_i - d_i, formed when the row is reached
            // (the labels never exceed d*, the label the search ends at).
            double puN[NK], fC[NK];
            unsigned long long reachedm[NK];
            unsigned long long closedm[NK];         // UL: columns scanned in this search, and the lanes beyond K
            unsigned long long demand[NK];
#pragma unroll
            for (int e = 0; e < NK; ++e) {
                const bool valid = lane + 64 * e < K;
#ifdef EMD_PERTURB
                double rc = __builtin_fmax(A[e] - pv[e], 0.0);
                if (rc < 1e-13) rc = EMD_PERTURB;
#else
                const double rc = __builtin_fmax(A[e] - pv[e], 0.0);        // (one v_max_f64; rc is never NaN)
#endif
                if constexpr (UL) { dC[e] = A[e]; closedm[e] = ~ballot_b(valid); }
                else { dC[e] = valid ? rc + 0.0 : NEG; closedm[e] = 0ull; }   // (+ 0.0: never -0, whose pattern would sort last)
                fC[e] = INF; parC[e] = Apar[e];
                puN[e] = pu[e]; reachedm[e] = srcmask[e]; parR[e] = -1;
                demand[e] = ballot_b(rb[e] > 0.0);                          // (rb is 0 beyond K)
            }
            PROF_END(2);
            double dstar = 0.0, last_bd = 0.0;
            double step_bd = 0.0;                   // label of the previous step of this search (0: the sources)
            bool exhausted = false, stale = false;
            for (int step = 0;; ++step) {
                if (step > 2 * K + 2) { tripped = true; trip_code = 1; break; }  // cannot happen: >= one node is scanned per step
#if defined(EMD_STAT) && EMD_STAT == 1
                ++n_stat;
#endif
                // smallest open label; ALL columns that carry it are final and are scanned in this one step (after the
                // first augmentations most arcs around the sources are tight, so dozens of nodes tie at the same label)
                PROF_BEGIN(3);
                double cur[NK];
#pragma unroll
                for (int e = 0; e < NK; ++e) {
                    if constexpr (UL) {
                        union { double d; unsigned int u[2]; } c;
                        // (a label that rounds a hair below zero IS zero: the clamp carries that rounding error from search to
                        // search; anything that turns -eps into a positive label, |.| for one, doubles it per search through the
                        // potential update -- ab_experiments.md r04 #9)
                        c.d = __builtin_fmax(dC[e] - pv[e], step_bd);
                        c.u[1] = __builtin_amdgcn_inverse_ballot_w64(closedm[e]) ? 0x7ff00000u : c.u[1];   // closed: never the minimum (>= +inf)
                        cur[e] = c.d;
                    } else {
                        cur[e] = dC[e];
                    }
                }
                // Labels never fall below the label of the previous step (step_bd; the labels are clamped to it), so if an open
                // column carries exactly that label -- the rows reached in the previous step had tight arcs, the usual case once
                // a few augmentations are done -- it IS the minimum: no wave-wide reduction, and its compare is the tie mask.
                unsigned long long tieC[NK], tieR[NK];
                bool same_label = false;
#pragma unroll
                for (int e = 0; e < NK; ++e) { tieC[e] = ballot_b(cur[e] == step_bd); same_label = same_label || tieC[e] != 0ull; }
                double bd = step_bd;
                if (!same_label) {
                    double best = cur[0];
#pragma unroll
                    for (int e = 1; e < NK; ++e) best = bits_less(cur[e], best) ? cur[e] : best;
                    bd = uni_f64(wave_min_f64(best));
                }
                PROF_END(3);
                if (hi_word(bd) >= 0x7ff00000u) {   // +inf or NEG: nothing (more) reachable
                    if (LAZY && stale) dstar = last_bd;     // ... from a tree that is out of date: search again
                    else exhausted = true;                  // ... at all: only rounding dust is left
                    break;
                }
                if constexpr (LAZY) last_bd = bd;
                step_bd = bd;
                PROF_BEGIN(4);
#pragma unroll
                for (int e = 0; e < NK; ++e) {
                    if (!same_label) tieC[e] = ballot_b(cur[e] == bd);
                    if constexpr (UL) closedm[e] |= tieC[e];
                    else if (__builtin_amdgcn_inverse_ballot_w64(tieC[e])) { fC[e] = bd; dC[e] = NEG; }
                }
                // Tied columns with demand left are targets: augment along the tree path right away, WITHOUT touching the
                // potentials, and let the search go on with the target as one more scanned column.  The labels are exact
                // distances from the sources the search started with, in a residual graph that only gains tight arcs
                // between scanned nodes, so every tree path stays a shortest (tight) path as long as its root still has
                // supply and its backward arcs still carry flow; a path that fails this test (a root that ran dry or an arc
                // that ran empty in an earlier augmentation of this search) is not used: the potentials are brought up to
                // date with the labels (d* = this step's label -- valid for ANY scanned set) and a new search starts.
                bool broke = false;
#pragma unroll
                for (int et = 0; et < NK; ++et) {
                    unsigned long long dm = broke || tripped ? 0ull : (tieC[et] & demand[et]);
                    while (dm) {                                   // wave-uniform
                        const int target = __builtin_ctzll(dm) + 64 * et;
                        dm &= dm - 1ull;
                        PROF_BEGIN(5);
                        // walk target <- ... <- source row once with wave-uniform indices; hop h is recorded in lane h (h % 64, slot
                        // h / 64; v_writelane): forward arc (hi -> hj) gains flow, backward arc (hi -> hb) loses it (hb < 0 at the
                        // source row).  The forward arcs enter the support right here (the bottleneck is positive: support arcs carry
                        // flow > 0, the source has supply > tol, the target demand > 0).
                        int hi[NK], hj[NK], hb[NK];
#pragma unroll
                        for (int e = 0; e < NK; ++e) { hi[e] = 0; hj[e] = 0; hb[e] = -1; }
                        int n_hops = 0, src_row = -1;
                        for (int j = target;;) {
                            if (n_hops >= 64 * NK || j < 0) { tripped = true; trip_code = 2; break; }
                            int i = 0, jb = 0;
                            if constexpr (NK == 1) {        // (0 <= j < K <= 64 and n_hops < 64 here: no slot selects -- scalar work)
                                i = rl_i32(parC[0], j);
                                if (i < 0) { tripped = true; trip_code = 3; break; }
                                jb = rl_i32(parR[0], i);
                                hi[0] = wl_i32(hi[0], i, n_hops);
                                hj[0] = wl_i32(hj[0], j, n_hops);
                                hb[0] = wl_i32(hb[0], jb, n_hops);
                                ship[0][0] |= lane == i ? 1ull << j : 0ull;
                            } else {
#pragma unroll
                            for (int e = 0; e < NK; ++e) if (e == j / 64) i = rl_i32(parC[e], j % 64);
                            if (i < 0) { tripped = true; trip_code = 3; break; }
#pragma unroll
                            for (int e = 0; e < NK; ++e) if (e == i / 64) jb = rl_i32(parR[e], i % 64);
#pragma unroll
                            for (int e = 0; e < NK; ++e)
                                if (e == n_hops / 64) {
                                    hi[e] = wl_i32(hi[e], i, n_hops % 64);
                                    hj[e] = wl_i32(hj[e], j, n_hops % 64);
                                    hb[e] = wl_i32(hb[e], jb, n_hops % 64);
                                }
#pragma unroll
                            for (int e = 0; e < NK; ++e)
                                if (lane + 64 * e == i) {
#pragma unroll
                                    for (int w = 0; w < NK; ++w) if (w == j / 64) ship[e][w] |= 1ull << (j % 64);
                                }
                            }
                            ++n_hops;
#if defined(EMD_STAT) && EMD_STAT == 3
                            ++n_stat;
#endif
                            if (jb < 0) { src_row = i; break; }              // a source row
                            j = jb;
                        }
                        if (tripped) break;
                        double delta, ra_s = 0.0;
                        {
                            double rb_t = 0.0;
#pragma unroll
                            for (int e = 0; e < NK; ++e) {
                                if (e == target / 64) rb_t = rl_f64(rb[e], target % 64);
                                if (e == src_row / 64) ra_s = rl_f64(ra[e], src_row % 64);
                            }
                            delta = rb_t < ra_s ? rb_t : ra_s;
                        }
                        // bottleneck: all backward-arc flows at once
                        double fb[NK];
                        auto bottleneck = [&]() {
                            double fmin = INF;
#pragma unroll
                            for (int e = 0; e < NK; ++e) {
                                const bool act = lane + 64 * e < n_hops;
                                fb[e] = (act && hb[e] >= 0) ? F[(size_t)hi[e] * K + hb[e]] : INF;
                                fmin = fb[e] < fmin ? fb[e] : fmin;
                            }
                            const double fm = wave_min_f64(fmin);
                            delta = uni_f64(fm < delta ? fm : delta);
                        };
                        if constexpr (LAZY) {
                            if (n_hops > 1) bottleneck();   // (wave-uniform; one hop = the source ships straight to the target)
                        }
                        if (LAZY && (!(ra_s > tol) || !(delta > 0.0))) {
                            // The tree is out of date here (its root ran dry, or a backward arc on the path ran empty, in an
                            // earlier augmentation of this search): nothing moves; arcs the walk marked but that carry no
                            // flow leave the support again, and a new search starts from up-to-date potentials.
#pragma unroll
                            for (int e = 0; e < NK; ++e) {
                                const bool act = lane + 64 * e < n_hops;
                                unsigned long long z = ballot_b(act && F[(size_t)hi[e] * K + hj[e]] == 0.0);
                                while (z) {
                                    const int h = __builtin_ctzll(z);
                                    z &= z - 1ull;
                                    const int i = rl_i32(hi[e], h), jf = rl_i32(hj[e], h);
#pragma unroll
                                    for (int e2 = 0; e2 < NK; ++e2)
                                        if (lane + 64 * e2 == i) {
#pragma unroll
                                            for (int w = 0; w < NK; ++w) if (w == jf / 64) ship[e2][w] &= ~(1ull << (jf % 64));
                                        }
                                }
                            }
                            broke = true;
                            break;
                        }
                        if (n_hops == 1) {
                            if (lane == 0) F[(size_t)src_row * K + target] += delta;
                        } else {
                            if constexpr (!LAZY) bottleneck();
                            // flow values, every hop in its own lane (the arcs of a simple path are distinct entries)
                            unsigned long long emptied[NK];
#pragma unroll
                            for (int e = 0; e < NK; ++e) {
                                const bool act = lane + 64 * e < n_hops;
                                if (act) {
                                    F[(size_t)hi[e] * K + hj[e]] += delta;
                                    if (hb[e] >= 0) F[(size_t)hi[e] * K + hb[e]] = fb[e] - delta;
                                }
                                emptied[e] = ballot_b(act && hb[e] >= 0 && fb[e] == delta);
                                if (emptied[e]) stale = true;          // a tree arc is gone
                            }
                            // backward arcs that ran empty leave the support
#pragma unroll
                            for (int e = 0; e < NK; ++e) {
                                unsigned long long m = emptied[e];
                                while (m) {                                    // wave-uniform, usually no or one arc
                                    const int h = __builtin_ctzll(m);
                                    m &= m - 1ull;
                                    const int i = rl_i32(hi[e], h), jb = rl_i32(hb[e], h);
#pragma unroll
                                    for (int e2 = 0; e2 < NK; ++e2)
                                        if (lane + 64 * e2 == i) {
#pragma unroll
                                            for (int w = 0; w < NK; ++w) if (w == jb / 64) ship[e2][w] &= ~(1ull << (jb % 64));
                                        }
                                }
                            }
                        }
                        EMD_FENCE();
                        __builtin_amdgcn_wave_barrier();
                        bool dry = false;
#pragma unroll
                        for (int e = 0; e < NK; ++e) {
                            if (lane + 64 * e == src_row) { ra[e] -= delta; dry = !(ra[e] > tol); }
                            if (lane + 64 * e == target) rb[e] -= delta;
                        }
                        ++n_aug;
                        demand[et] = ballot_b(rb[et] > 0.0);
                        PROF_END(5);
                        // a dry root, or a target that keeps demand (its path gave out first): the tree is out of date below them
                        if (ballot_b(dry) || ((demand[et] >> (target % 64)) & 1ull)) stale = true;
                        if (!LAZY && stale) { broke = true; break; }        // restart at once: every path found is usable
                    }
                }
                PROF_END(4);
                if (tripped) break;
                if (broke) { dstar = bd; break; }
                PROF_BEGIN(6);
                // columns: backward arcs to the rows that ship to ANY of the tied columns (reduced cost 0): the ballot
                // mask of the tied columns IS a column bit mask, so one AND with the row's support finds them
                double pb[NK];                                               // pu_i - d_i of the rows reached in this step (d_i = bd)
#pragma unroll
                for (int e = 0; e < NK; ++e) {
                    unsigned long long hit = 0ull;
                    int wsel = 0;
#pragma unroll
                    for (int w = 0; w < NK; ++w) {
                        const unsigned long long h = ship[e][w] & tieC[w];
                        if (h) { hit = h; wsel = w; }
                    }
                    tieR[e] = ballot_b(hit != 0ull) & ~reachedm[e];
                    reachedm[e] |= tieR[e];
                    pb[e] = pu[e] - bd;
                    if (__builtin_amdgcn_inverse_ballot_w64(tieR[e])) { parR[e] = __builtin_ctzll(hit) + 64 * wsel; puN[e] = pb[e]; }
                }
                // those rows: forward arcs to every open column
                // (tried: rows tied in one step sharing the per-column tail -- min_i (M_ij - pu_i) first, then one subtract of
                // pv_j, clamp, add and compare per step: 7.27 -> 7.48 ms at c3, the two extra live values cost more than the
                // fp64 operations they save at 64 registers per lane)
                {
#pragma unroll
                for (int e = 0; e < NK; ++e) {
                    unsigned long long m = tieR[e];
                    while (m) {
#if defined(EMD_STAT) && EMD_STAT == 2
                        ++n_stat;
#endif
                        const int l = __builtin_ctzll(m);
                        m = clear_bit(m, l);
                        const int in = l + 64 * e;
                        const double t_i = rl_f64(UL ? pb[e] : pu[e], l);    // (UL: pu_i - d_i; !UL: pu_i)
#pragma unroll
                        for (int e2 = 0; e2 < NK; ++e2) {
                            // (PAD / BF: no bounds test -- a lane beyond K reads a pad column / column K - 1, its own label is closed for good)
                            if (PAD || BF || lane + 64 * e2 < K) {
                                const int idx = BF ? mcol[e2] : lane + 64 * e2;
                                double nd;
                                bool lt;
                                if constexpr (UL) {
                                    nd = Mrd[(size_t)in * MP + idx] - t_i;
                                    // (through the masks: the compiler otherwise turns `open` into a branch around the read)
                                    lt = __builtin_amdgcn_inverse_ballot_w64(ballot_b(nd < dC[e2]) & ~closedm[e2]);
                                } else {
                                    double rc = Mrd[(size_t)in * MP + idx] - t_i - pv[e2];
                                    rc = __builtin_fmax(rc, 0.0);
#ifdef EMD_PERTURB
                                    if (rc < 1e-13) rc = EMD_PERTURB;
#endif
                                    nd = bd + rc;
                                    lt = nd < dC[e2];                                // never true for a scanned column (NEG)
                                }
                                dC[e2] = lt ? nd : dC[e2];          // (selects, not a branch around two moves)
                                parC[e2] = lt ? in : parC[e2];
                            }
                        }
                    }
                }
                }
                PROF_END(6);
            }
            if (tripped) break;
            if (exhausted) {    // numerically exhausted: drop the dust (<= tol-scale mass) of every remaining source
#pragma unroll
                for (int e = 0; e < NK; ++e) ra[e] = 0.0;
                break;
            }
            PROF_BEGIN(9);
            // potentials: rc'(i,j) = rc(i,j) + min(d_i, d*) - min(d_j, d*) >= 0, and 0 on the tree (so on every path used)
#pragma unroll
            for (int e = 0; e < NK; ++e) {
                pu[e] = __builtin_amdgcn_inverse_ballot_w64(reachedm[e]) ? puN[e] : pu[e] - dstar;     // (reached: d_i <= d*)
                if constexpr (UL)    // (the label a column was scanned with: L and pv have not moved since)
                    fC[e] = __builtin_amdgcn_inverse_ballot_w64(closedm[e]) && lane + 64 * e < K ? __builtin_fmax(dC[e] - pv[e], 0.0) : INF;
                pv[e] += __builtin_fmin(fC[e], dstar);
            }
            ++n_search;
            PROF_END(9);
#if defined(EMD_STAT) && EMD_STAT == 5
            ++n_stat;
#endif
        }
        // cost = sum over the support of F_ij * M_ij (lane i walks the bits of row i)
        PROF_BEGIN(7);
        double cost = 0.0;
#pragma unroll
        for (int e = 0; e < NK; ++e) {
            const int idx = lane + 64 * e;
#pragma unroll
            for (int w = 0; w < NK; ++w) {
                unsigned long long m = ship[e][w];
                while (m) {
                    const int j = __builtin_ctzll(m) + 64 * w;
                    m &= m - 1ull;
                    // (one fused multiply-add per arc in either form: the two forms give the same bits)
                    cost = __builtin_fma(F[(size_t)idx * K + j], Mrd[(size_t)idx * MP + j], cost);
                    F[(size_t)idx * K + j] = 0.0;           // the slab goes back to all zeros for the wave's next pair
                }
            }
        }
        cost = uni_f64(wave_sum_f64(cost));
        PROF_END(7); PROF_END(1);
        if (tripped) {          // (never seen; a guard that tripped mid-augmentation may leave entries outside the masks)
            for (int t = lane; t < K * K; t += 64) F[t] = 0.0;
        }
        if (lane == 0) {
            p.emd[q] = tripped ? __builtin_nan("") : cost;
#if defined(EMD_PROF)
            if (p.n_aug) p.n_aug[q] = (int)(prof_acc >> 4);
#elif defined(EMD_STAT)
            if (p.n_aug) p.n_aug[q] = n_stat;
#else
            if (p.n_aug) p.n_aug[q] = tripped ? -(n_aug * 8 + trip_code) : n_aug;
#endif
        }
        EMD_FENCE();      // (the zeros written back above: other lanes write these entries next)
        __builtin_amdgcn_wave_barrier();
    }
}

// mirror the strictly-lower triangle from the upper one (full square grids only)
__global__ void emd_mirror_kernel(double *E, int N) {
    const long total = (long)N * N;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t / N), j = (int)(t % N);
        if (j < i) E[t] = E[(size_t)j * N + i];
    }
}

}  // namespace pilot
