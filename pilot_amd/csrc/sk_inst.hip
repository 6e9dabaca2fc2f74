// Template instantiations of the Sinkhorn kernels, one slice per translation unit (-DSK_PART=0|1: f32 | f64).
#include "sinkhorn_launch.hpp"

namespace pilot {
namespace {

template <class C, int RT, bool SYM, bool TRACK, int TV = 0>
hipError_t stream_one(dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    auto kern = sinkhorn_stream_kernel<C, RT, SYM, TRACK, TV>;
    if (lds > 32 * 1024) {   // beyond the default dynamic-LDS window the limit must be raised explicitly
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVE * WAVES_PER_WG), lds, s, p);
    return hipGetLastError();
}
template <class C, int RT, int TV = 0>
hipError_t stream_rt(bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    if (track) return sym ? stream_one<C, RT, true, true, TV>(grid, lds, s, p) : stream_one<C, RT, false, true, TV>(grid, lds, s, p);
    return sym ? stream_one<C, RT, true, false, TV>(grid, lds, s, p) : stream_one<C, RT, false, false, TV>(grid, lds, s, p);
}
template <class C, int TV = 0>
hipError_t stream_any(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    switch (RT) {
    case 1: if constexpr (TV == 0) return stream_rt<C, 1, TV>(sym, track, grid, lds, s, p); else return hipErrorInvalidValue;
    case 2: return stream_rt<C, 2, TV>(sym, track, grid, lds, s, p);
    case 3: return stream_rt<C, 3, TV>(sym, track, grid, lds, s, p);
    case 4: return stream_rt<C, 4, TV>(sym, track, grid, lds, s, p);
    case 5: return stream_rt<C, 5, TV>(sym, track, grid, lds, s, p);
    case 6: return stream_rt<C, 6, TV>(sym, track, grid, lds, s, p);
    case 7: return stream_rt<C, 7, TV>(sym, track, grid, lds, s, p);
    case 8: return stream_rt<C, 8, TV>(sym, track, grid, lds, s, p);
    default: return hipErrorInvalidValue;
    }
}

// prepare a call: operand images / tables / slot-ordered proportions + the longest-first order keys, then the scatter
template <class C>
hipError_t prep_any(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                    double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                    int *main_queue_head, int mode, int n_blocks, hipStream_t s) {
    using T = typename C::T;
    const int collapse = (mode >> 2) & 1;    // bit 2: no longest-first order (experiment switch)
    const int tiles = n_rows > 0 ? ((N + ORDER_JW - 1) / ORDER_JW) * ((n_rows + ORDER_RI - 1) / ORDER_RI) : 0;
    const int K4 = (K + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)K4 * (ORDER_JW + 1) + (size_t)ORDER_RI * K4) + sizeof(int) * ORDER_NB;
    static bool attr_set[64] = {};    // more than 64 KB of dynamic LDS (K > 112) needs the opt-in; once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sinkhorn_prep_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);      // (K = 256: 140 KB of column tile)
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(sinkhorn_prep_kernel<C>, dim3(tiles + PREP_SETUP_BLOCKS), dim3(256), lds, s, M, K, RT, reg, static_cast<T *>(img), P,
                       static_cast<T *>(Pslot), N, write_tail, stop_thr, floor_ulps, tiles, n_rows, row_begin, row_step, bucket, hist, collapse);
    if (n_rows > 0)
        hipLaunchKernelGGL(order_scatter_kernel, dim3(n_blocks), dim3(256), 0, s, bucket, n_rows * N, hist, hist + ORDER_NB, list, split,
                           main_queue_head, mode & 3);
    return hipGetLastError();
}
}  // namespace

#if SK_PART == 0
hipError_t launch_stream_f32(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    return stream_any<CfgF32x16>(RT, sym, track, grid, lds, s, p);
}
hipError_t launch_prep_f64(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s);
hipError_t launch_prep_s32(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s);
hipError_t launch_prep_h32(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s);
hipError_t launch_prep(int cfg, const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                       double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                       int *main_queue_head, int mode, int n_blocks, hipStream_t s) {
    if (cfg == CFG_F64)
        return launch_prep_f64(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
    if (cfg == CFG_H32)
        return launch_prep_h32(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
    if (cfg == CFG_S32)
        return launch_prep_s32(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
    return prep_any<CfgF32x16>(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
}
size_t form_elems_rt(int cfg, int RT) {
    if (cfg == CFG_H32) return (size_t)form_elems<CfgH32x16>(RT);
    return cfg == CFG_S32 ? (size_t)form_elems<CfgS32x16>(RT) : (cfg == CFG_F64 ? (size_t)form_elems<CfgF64x16>(RT) : (size_t)form_elems<CfgF32x16>(RT));
}
size_t track_img_elems(int cfg, int RT) { return cfg == CFG_H32 ? (size_t)track_img_offset<CfgH32x16>(RT) : 0; }
size_t img_elems(int cfg, int RT) {
    if (cfg == CFG_H32) return (size_t)img_total<CfgH32x16>(RT);
    return cfg == CFG_S32 ? (size_t)img_total<CfgS32x16>(RT) : (cfg == CFG_F64 ? (size_t)img_total<CfgF64x16>(RT) : (size_t)img_total<CfgF32x16>(RT));
}
#define PILOT_TV_DECL(NAME) \
    hipError_t launch_stream_##NAME(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
PILOT_TV_DECL(f32_tv1) PILOT_TV_DECL(f32_tv2) PILOT_TV_DECL(f64_tv1) PILOT_TV_DECL(f64_tv2)
#undef PILOT_TV_DECL
hipError_t launch_stream_tv(int cfg, int tv, int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    if (cfg == CFG_F32) return tv == 1 ? launch_stream_f32_tv1(RT, sym, track, grid, lds, s, p) : launch_stream_f32_tv2(RT, sym, track, grid, lds, s, p);
    return tv == 1 ? launch_stream_f64_tv1(RT, sym, track, grid, lds, s, p) : launch_stream_f64_tv2(RT, sym, track, grid, lds, s, p);
}
#elif SK_PART == 1
hipError_t launch_stream_f64(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    return stream_any<CfgF64x16>(RT, sym, track, grid, lds, s, p);
}
hipError_t launch_solo_track_f64(dim3 grid, hipStream_t s, const GridParams &p) {
    hipLaunchKernelGGL(sinkhorn_solo_track_kernel<CfgF64x16>, grid, dim3(WAVE * WAVES_PER_WG), sizeof(double) * WAVE * WAVES_PER_WG, s, p);
    return hipGetLastError();
}
hipError_t launch_prep_f64(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s) {
    return prep_any<CfgF64x16>(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
}
#elif SK_PART == 7
// split configuration with only register 0 of the last row-tile live (K mod 16 in 1..4): TV = 1 skips the dead registers
hipError_t launch_stream_s32_l1(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    return stream_any<CfgS32x16, 1>(RT, sym, track, grid, lds, s, p);
}
#elif SK_PART == 9
// fp16-split configuration (fast pass only: no tracking variant), dead registers of the last row-tile skipped
hipError_t launch_stream_h32_l1(int RT, bool sym, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    switch (RT) {
    case 2: return sym ? stream_one<CfgH32x16, 2, true, false, 1>(grid, lds, s, p) : stream_one<CfgH32x16, 2, false, false, 1>(grid, lds, s, p);
    case 3: return sym ? stream_one<CfgH32x16, 3, true, false, 1>(grid, lds, s, p) : stream_one<CfgH32x16, 3, false, false, 1>(grid, lds, s, p);
    case 4: return sym ? stream_one<CfgH32x16, 4, true, false, 1>(grid, lds, s, p) : stream_one<CfgH32x16, 4, false, false, 1>(grid, lds, s, p);
    default: return hipErrorInvalidValue;
    }
}
#elif SK_PART == 8
hipError_t launch_stream_h32_l1(int RT, bool sym, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
hipError_t launch_stream_h32(int RT, bool sym, int live1, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    if (live1) return launch_stream_h32_l1(RT, sym, grid, lds, s, p);
    switch (RT) {
    case 1: return sym ? stream_one<CfgH32x16, 1, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 1, false, false>(grid, lds, s, p);
    case 2: return sym ? stream_one<CfgH32x16, 2, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 2, false, false>(grid, lds, s, p);
    case 3: return sym ? stream_one<CfgH32x16, 3, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 3, false, false>(grid, lds, s, p);
    case 4: return sym ? stream_one<CfgH32x16, 4, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 4, false, false>(grid, lds, s, p);
    case 5: return sym ? stream_one<CfgH32x16, 5, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 5, false, false>(grid, lds, s, p);
    case 6: return sym ? stream_one<CfgH32x16, 6, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 6, false, false>(grid, lds, s, p);
    case 7: return sym ? stream_one<CfgH32x16, 7, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 7, false, false>(grid, lds, s, p);
    case 8: return sym ? stream_one<CfgH32x16, 8, true, false>(grid, lds, s, p) : stream_one<CfgH32x16, 8, false, false>(grid, lds, s, p);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_prep_h32(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s) {
    return prep_any<CfgH32x16>(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
}
#elif SK_PART == 6
hipError_t launch_stream_s32_l1(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
hipError_t launch_stream_s32(int RT, bool sym, bool track, int live1, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    if (live1) return launch_stream_s32_l1(RT, sym, track, grid, lds, s, p);
    return stream_any<CfgS32x16>(RT, sym, track, grid, lds, s, p);
}
hipError_t launch_prep_s32(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s) {
    return prep_any<CfgS32x16>(M, K, RT, reg, img, P, Pslot, N, write_tail, stop_thr, floor_ulps, n_rows, row_begin, row_step, bucket, hist, list, split,
                               main_queue_head, mode, n_blocks, s);
}
#else
#if SK_PART == 2
#define PILOT_TV_CFG CfgF32x16
#define PILOT_TV_N 1
#define PILOT_TV_NAME(x) x##f32_tv1
#elif SK_PART == 3
#define PILOT_TV_CFG CfgF32x16
#define PILOT_TV_N 2
#define PILOT_TV_NAME(x) x##f32_tv2
#elif SK_PART == 4
#define PILOT_TV_CFG CfgF64x16
#define PILOT_TV_N 1
#define PILOT_TV_NAME(x) x##f64_tv1
#else
#define PILOT_TV_CFG CfgF64x16
#define PILOT_TV_N 2
#define PILOT_TV_NAME(x) x##f64_tv2
#endif
hipError_t PILOT_TV_NAME(launch_stream_)(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p) {
    return stream_any<PILOT_TV_CFG, PILOT_TV_N>(RT, sym, track, grid, lds, s, p);
}
#endif

}  // namespace pilot
