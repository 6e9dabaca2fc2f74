// Microbenchmark: v_mfma_f32_16x16x32_bf16 beside f32 VALU work on one SIMD (does VALU co-issue with the bf16 matrix
// pipe, unlike with the f32-input MFMA?).  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_valu mfma_bf16_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = float __attribute__((ext_vector_type(4)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using u4 = unsigned int __attribute__((ext_vector_type(4)));

// NV VALU instructions (and / sub / perm / rcp / mul mix, like an operand split) per group of 4 MFMAs
template <int NV>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f4 acc[4];
    u4 a[4], b;
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = f4{0, 0, 0, 0}; a[i] = u4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; }
    b = u4{0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12; ++rep) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[i]), __builtin_bit_cast(bf8, b), acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float &x = v[j % 8], &y = v[(j + 3) % 8];
                switch (j % 8) {
                case 0: case 4: x = __uint_as_float(__float_as_uint(x) & 0xffff0000u) + y; break;   // and + add (2 instr, counted as 1: see NV scaling)
                case 1: case 5: x = x - y; break;
                case 2: x = __builtin_amdgcn_rcpf(x); break;
                case 3: case 7: x = x * y; break;
                default: x = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(x), __float_as_uint(y), 0x07060302u)); break;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV> void run(int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV><<<256 * wgs_per_cu, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_simd = (double)iters * 12 * wgs_per_cu;          // each WG = 4 waves = 1 wave per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("4 x mfma_16x16x32_bf16 + %2d VALU-ish per group: waves/SIMD=%d  %.1f cycles per group per SIMD (@2.4GHz) = %.1f per MFMA\n",
           NV, wgs_per_cu, cyc / groups_per_simd, cyc / groups_per_simd / 4);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0>(w); run<4>(w); run<8>(w); run<12>(w); run<16>(w); run<24>(w);
    }
    return 0;
}
