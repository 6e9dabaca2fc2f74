// Microbenchmark: how much independent f32 VALU work fits beside bf16 MFMAs of the two shapes on one SIMD?
// Each loop body is [MFMA, n x v_fma_f32] repeated (program order interleaved, like a software-pipelined kernel); both
// shapes do the same flop per "unit" (one 32x32x16 = two 16x16x32).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = float __attribute__((ext_vector_type(4)));
using f16v = float __attribute__((ext_vector_type(16)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using u4 = unsigned int __attribute__((ext_vector_type(4)));

template <int SHAPE32, int NV>      // NV = VALU instructions per 16x16x32-equivalent (half a 32x32x16)
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f4 acc[4];
    f16v big[2];
    u4 a = u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u}, b = u4{0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    const float c0 = seed * 0.999f, c1 = seed * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if constexpr (SHAPE32) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), big[i], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2 * NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], c0, c1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[i], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], c0, c1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE32, int NV> void run(int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE32, NV><<<256 * wgs_per_cu, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE32, NV><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double units = (double)iters * 8 * 4 * wgs_per_cu;          // 16x16x32-equivalents per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%s + %d v_fma per 16x16x32-equivalent, waves/SIMD=%d: %.1f nominal cycles (2.4 GHz) per 16x16x32-equivalent\n",
           SHAPE32 ? "32x32x16" : "16x16x32", NV, wgs_per_cu, cyc / units);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w); run<0, 1>(w); run<0, 2>(w); run<0, 3>(w); run<0, 4>(w); run<0, 6>(w);
        run<1, 0>(w); run<1, 1>(w); run<1, 2>(w); run<1, 3>(w); run<1, 4>(w); run<1, 6>(w);
    }
    return 0;
}
