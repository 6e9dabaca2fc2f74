// The 8-waves-per-tile Sinkhorn kernels for 128 < K <= 256 (wide_kernels.hpp), the 4-waves-per-tile kernel for 112 < K <= 128
// (quad_kernels.hpp) and their launch entry points.
#include "sinkhorn_launch.hpp"
#include "wide_kernels.hpp"
#include "quad_kernels.hpp"

namespace pilot {

hipError_t launch_wide(dim3 grid, hipStream_t s, const GridParams &p, float *rec) {
    hipLaunchKernelGGL(sinkhorn_wide_kernel, grid, dim3(WAVE * WIDE_WAVES), 0, s, p, rec);
    return hipGetLastError();
}
hipError_t launch_wide_value(dim3 grid, hipStream_t s, const GridParams &p, const float *rec) {
    hipLaunchKernelGGL(sinkhorn_wide_value_kernel, grid, dim3(WAVE * WAVES_PER_WG), 0, s, p, rec);
    return hipGetLastError();
}
size_t wide_rec_elems() { return (size_t)WIDE_REC; }
hipError_t launch_quad(dim3 grid, hipStream_t s, const GridParams &p) {
    hipLaunchKernelGGL(sinkhorn_quad_kernel<8>, grid, dim3(WAVE * QUAD_WAVES), 0, s, p);
    return hipGetLastError();
}
bool quad_covers(int K, bool sym) { return sym && K >= QUAD_MIN_K && K <= QUAD_MAX_K; }

}  // namespace pilot
