// Launch entry points of the Sinkhorn kernels.  The template instantiations are spread over several translation
// units (sk_inst.hip compiled with -DSK_PART=0 for f32, 1 for f64, 2..5 for the VALU-tail variants: f32 tv1, f32 tv2, f64 tv1, f64 tv2) so that `make -j` builds them in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "sinkhorn_kernels.hpp"

namespace pilot {

enum { CFG_F32 = 0, CFG_F64 = 1 };   // CfgF32x16, CfgF64x16

// persistent stream kernel (one tile per wave); track: tau-tracking variant
hipError_t launch_stream_f32(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
hipError_t launch_stream_f64(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
// variants with the last row-tile on the VALU (tv = 1: <= 2 live rows, 2: <= 4; see tail_rows); RT >= 2
hipError_t launch_stream_tv(int cfg, int tv, int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
hipError_t launch_coop_tv(int cfg, int tv, int RT, bool sym, int n_wgs, hipStream_t s, const GridParams &p);
// cooperative kernel for the head of the longest-first list (one workgroup of RT waves per tile)
hipError_t launch_coop_f32(int RT, bool sym, int n_wgs, hipStream_t s, const GridParams &p);
hipError_t launch_coop_f64(int RT, bool sym, int n_wgs, hipStream_t s, const GridParams &p);
// helpers
hipError_t launch_value_f32(int RT, dim3 grid, hipStream_t s, const GridParams &p);
hipError_t launch_value_f64(int RT, dim3 grid, hipStream_t s, const GridParams &p);
// one call's preparation: operand images + tables + slot-ordered proportions and the longest-first order keys in one
// launch (sinkhorn_prep_kernel), then the scatter.  mode: bit 0 cooperative head, bit 1 solo duplicates, bit 2 natural order.
hipError_t launch_prep_f32(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s);
hipError_t launch_prep_f64(const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                           double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                           int *main_queue_head, int mode, int n_blocks, hipStream_t s);

}  // namespace pilot
