// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 for one wave per SIMD (and more), with and without
// interleaved VALU work.  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = float __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f4 acc[NACC];
    float a[NACC], b = seed + threadIdx.x * 1e-3f;
    float v0 = seed, v1 = seed * 2, v2 = seed * 3, v3 = seed * 4;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = f4{0, 0, 0, 0}; a[i] = seed + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[i], 0, 0, 0);
            if (MODE == 1) {   // 2 rcp + 2 mul per NACC MFMAs
                v0 = __builtin_amdgcn_rcpf(v0) * v2; v1 = __builtin_amdgcn_rcpf(v1) * v3;
            } else if (MODE == 2) {   // 4 fma
                v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
            } else if (MODE == 3) {   // 8 independent-ish VALU
                v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
                v0 = __builtin_amdgcn_rcpf(v0) * v2; v1 = __builtin_amdgcn_rcpf(v1) * v3;
            } else if (MODE == 4 || MODE == 6) {   // 4 (8) v_pk_fma_f32
                using f2 = float __attribute__((ext_vector_type(2)));
                f2 p0 = {v0, v1}, p1 = {v2, v3};
                p0 = __builtin_elementwise_fma(p0, p1, p0); p1 = __builtin_elementwise_fma(p1, p0, p1);
                p0 = __builtin_elementwise_fma(p0, p1, p0); p1 = __builtin_elementwise_fma(p1, p0, p1);
                if (MODE == 6) {
                    p0 = __builtin_elementwise_fma(p0, p1, p0); p1 = __builtin_elementwise_fma(p1, p0, p1);
                    p0 = __builtin_elementwise_fma(p0, p1, p0); p1 = __builtin_elementwise_fma(p1, p0, p1);
                }
                v0 = p0[0]; v1 = p0[1]; v2 = p1[0]; v3 = p1[1];
            } else if (MODE == 5) {   // 2 permlane32_swap + 2 permlane16_swap + 4 add
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
                v0 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
                v0 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
                v1 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
                v1 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + v2 + v3;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE> void run(const char *name, int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, MODE><<<256 * wgs_per_cu, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, MODE><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * NACC * wgs_per_cu;          // each WG = 4 waves = 1 wave per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-44s waves/SIMD=%d  %.1f cycles per MFMA per SIMD (@2.4GHz)  -> %.1f TF\n", name, wgs_per_cu, cyc / mfma_per_simd,
           mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

using f16v = float __attribute__((ext_vector_type(16)));
template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k32(float *out, int iters, float seed) {
    f16v acc[NACC];
    float a[NACC], b = seed + threadIdx.x * 1e-3f;
    float v0 = seed, v1 = seed * 2, v2 = seed * 3, v3 = seed * 4;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) acc[i][j] = 0; a[i] = seed + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i], 0, 0, 0);
            if (MODE == 1) { v0 = __builtin_amdgcn_rcpf(v0) * v2; v1 = __builtin_amdgcn_rcpf(v1) * v3; }
            else if (MODE == 2) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
            else if (MODE == 3) {
                v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
                v0 = __builtin_amdgcn_rcpf(v0) * v2; v1 = __builtin_amdgcn_rcpf(v1) * v3;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + v2 + v3;
#pragma unroll
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MODE> void run32(const char *name, int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k32<NACC, MODE><<<256 * wgs_per_cu, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k32<NACC, MODE><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * NACC * wgs_per_cu;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("32x32x2 %-36s waves/SIMD=%d  %.1f cycles per MFMA per SIMD (@2.4GHz)  -> %.1f TF\n", name, wgs_per_cu, cyc / mfma_per_simd,
           mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run32<2, 0>("2 acc, no VALU", w);
        run32<2, 1>("2 acc + 2 rcp + 2 mul per 2 MFMA", w);
        run32<2, 2>("2 acc + 4 fma per 2 MFMA", w);
        run32<2, 3>("2 acc + 8 VALU per 2 MFMA", w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>("1 accumulator (dependent chain), no VALU", w);
        run<2, 0>("2 accumulators, no VALU", w);
        run<4, 0>("4 accumulators, no VALU", w);
        run<7, 0>("7 accumulators, no VALU", w);
        run<4, 1>("4 acc + 2 rcp + 2 mul per 4 MFMA", w);
        run<4, 2>("4 acc + 4 fma per 4 MFMA", w);
        run<4, 3>("4 acc + 4 fma + 2 rcp + 2 mul per 4 MFMA", w);
        run<7, 3>("7 acc + 4 fma + 2 rcp + 2 mul per 7 MFMA", w);
        run<4, 4>("4 acc + 4 pk_fma per 4 MFMA", w);
        run<4, 6>("4 acc + 8 pk_fma per 4 MFMA", w);
        run<4, 5>("4 acc + 4 permlane swap + 4 add per 4 MFMA", w);
    }
    return 0;
}
