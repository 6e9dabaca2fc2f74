// Launch entry points of the Sinkhorn kernels.  The template instantiations are spread over several translation
// units (sk_inst.hip compiled with -DSK_PART=0 for f32, 1 for f64, 2..5 for the VALU-tail variants: f32 tv1, f32 tv2,
// f64 tv1, f64 tv2, 6 for the bf16-split configuration) so that `make -j` builds them in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "sinkhorn_kernels.hpp"

namespace pilot {

enum { CFG_F32 = 0, CFG_F64 = 1, CFG_S32 = 2, CFG_H32 = 3 };   // CfgF32x16, CfgF64x16, CfgS32x16 (bf16-split products, f32 values), CfgH32x16 (fp16-split)

// persistent stream kernel (one tile per wave); track: tau-tracking variant
hipError_t launch_stream_f32(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
hipError_t launch_stream_f64(int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
// a short list of pairs (p.list / p.list_len / p.queue_head), one wave per pair, absorptions tracked; symmetric cost, K <= 64
hipError_t launch_solo_track_f64(dim3 grid, hipStream_t s, const GridParams &p);
// live1: K mod 16 in 1..4, the dead registers of the last row-tile are skipped (RT >= 2)
hipError_t launch_stream_s32(int RT, bool sym, bool track, int live1, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
// fp16-split configuration: fast pass only (its tracking pass is the bf16-split kernel on the block at track_img_elems)
hipError_t launch_stream_h32(int RT, bool sym, int live1, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
// variants with the last row-tile on the VALU (tv = 1: <= 2 live rows, 2: <= 4; see tail_rows); RT >= 2
hipError_t launch_stream_tv(int cfg, int tv, int RT, bool sym, bool track, dim3 grid, size_t lds, hipStream_t s, const GridParams &p);
// one call's preparation: operand images + tables + slot-ordered proportions and the longest-first order keys in one
// launch (sinkhorn_prep_kernel), then the scatter.  mode: bit 1 solo duplicates, bit 2 natural order.
hipError_t launch_prep(int cfg, const double *M, int K, int RT, double reg, void *img, const double *P, void *Pslot, int N, int write_tail,
                       double stop_thr, double floor_ulps, int n_rows, int row_begin, int row_step, unsigned char *bucket, int *hist, int *list, int *split,
                       int *main_queue_head, int mode, int n_blocks, hipStream_t s);
// 128 < K <= 256, symmetric cost, fp16-split range (wide_kernels.hpp): the pair-grid kernel (8 waves per 16-pair tile; one
// record of wide_rec_elems() 4-byte words per pair in `rec`) and the kernel that turns the records into costs
hipError_t launch_wide(dim3 grid, hipStream_t s, const GridParams &p, float *rec);
hipError_t launch_wide_value(dim3 grid, hipStream_t s, const GridParams &p, const float *rec);
size_t wide_rec_elems();
// 112 < K <= 128, symmetric cost: the fast pass of the fp16-split configuration with four waves per 16-pair tile (quad_kernels.hpp);
// same GridParams, work list, hand-over and NaN lists as launch_stream_h32, one tile per 256-thread workgroup, costs formed in the kernel
hipError_t launch_quad(dim3 grid, hipStream_t s, const GridParams &p);
bool quad_covers(int K, bool sym);
// elements of T in the operand block of a call (see img_layout in sinkhorn_kernels.hpp)
size_t img_elems(int cfg, int RT);
size_t form_elems_rt(int cfg, int RT);
// offset (in 4-byte elements) of the tracking kernel's operand block inside the call's block; 0: the block itself
size_t track_img_elems(int cfg, int RT);

}  // namespace pilot
