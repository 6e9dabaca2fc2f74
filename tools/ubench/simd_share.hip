// How do the waves of one SIMD share it?  Whole-GPU timing (HIP events) of a fixed amount of work PER WAVE with 1, 2, 4 waves
// per SIMD: (a) independent v_mul_f32, (b) v_mfma_f32_16x16x32_f16 only, (c) half the waves (a), half (b), (d) every wave
// alternating blocks of 24 MFMAs and 88 VALU instructions (the shape of one half-update of the fp16-split Sinkhorn kernel).
// build: hipcc -O2 --offload-arch=gfx950 -o simd_share.bin simd_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8_t = _Float16 __attribute__((ext_vector_type(8)));
using f32x4_t = float __attribute__((ext_vector_type(4)));

#define V8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));

__device__ inline void valu_block(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &a6, float &a7, float b) {
    V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL) V8(MUL)      // 88
}
__device__ inline void mfma_block(f32x4_t (&c)[4], f16x8_t x, f16x8_t y) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c[j], 0, 0, 0);    // 24, four independent chains
}

__device__ inline void mfma_block_seq(f32x4_t (&c)[4], f16x8_t x, f16x8_t y) {     // the same 24, chain after chain (6 dependent in a row)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c[j], 0, 0, 0);
}
using f32x16_t = float __attribute__((ext_vector_type(16)));
// the same flop as two mfma_block calls on HALF the instructions: 24 x v_mfma_f32_32x32x16_f16, four independent chains
__device__ inline void mfma_block32(f32x16_t (&c)[4], f16x8_t x, f16x8_t y) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c[j], 0, 0, 0);
}
// mode 6 / 7 (round 5): ONE wave does the work two waves do in mode 3 -- 24 32x32x16 MFMAs (= 48 16x16x32) and 176 VALU per iteration
// (7: the VALU in two halves around the MFMAs)
__global__ void k32(float *out, const float *in, int iters, int mode) {
    float a0 = in[threadIdx.x % 64], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = in[64 + threadIdx.x % 64];
    f16x8_t x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)in[i]; y[i] = (_Float16)in[8 + i]; }
    f32x16_t c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (mode == 7) valu_block(a0, a1, a2, a3, a4, a5, a6, a7, b);
        mfma_block32(c, x, y);
        valu_block(a0, a1, a2, a3, a4, a5, a6, a7, b);
        if (mode != 7) valu_block(a0, a1, a2, a3, a4, a5, a6, a7, b);
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += c[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// mode 0: VALU only; 1: MFMA only; 2: even waves VALU, odd waves (by wave / 4: the SIMD partner) MFMA; 3: alternate in every wave
__global__ void k(float *out, const float *in, int iters, int mode) {
    float a0 = in[threadIdx.x % 64], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = in[64 + threadIdx.x % 64];
    f16x8_t x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)in[i]; y[i] = (_Float16)in[8 + i]; }
    f32x4_t c[4];
    for (int j = 0; j < 4; ++j) c[j] = f32x4_t{0, 0, 0, 0};
    const int wave = threadIdx.x / 64;
    const bool do_valu = mode == 0 || mode == 3 || mode == 5 || (mode == 2 && ((wave / 4) & 1) == 0);
    const bool do_mfma = mode == 1 || mode == 3 || (mode == 2 && ((wave / 4) & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (mode >= 4) mfma_block_seq(c, x, y);
        else if (do_mfma) mfma_block(c, x, y);
        if (do_valu) valu_block(a0, a1, a2, a3, a4, a5, a6, a7, b);
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *d_in, *d_out;
    hipMalloc(&d_in, 4096); hipMalloc(&d_out, 4 * 1024 * 1024 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0f + 1e-3f * i;
    hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[] = {"VALU only (88 v_mul per block)", "MFMA only (24 16x16x32 f16 per block)", "half the waves VALU, their SIMD partners MFMA", "every wave: 24 MFMA then 88 VALU", "MFMA only, 4 chains of 6 dependent in sequence", "every wave: 24 MFMA (sequential chains) then 88 VALU"};
    printf("ns per block-iteration of one wave (whole GPU busy, 256 workgroups); waves per SIMD = threads / 256\n");
    for (int mode = 0; mode < 6; ++mode)
        for (int threads = 256; threads <= 1024; threads *= 2) {
            if (mode == 2 && threads == 256) continue;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d_out, d_in, iters, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-50s %d waves/SIMD: %8.1f ns per iteration per wave  (%.1f ns per SIMD per wave-iteration)\n", names[mode], threads / 256, ms * 1e6 / iters, ms * 1e6 / iters / (threads / 256));
        }
    printf("\nround 5: the work of TWO mode-3 waves (48 16x16x32 MFMAs + 176 VALU) in ONE wave with 24 32x32x16 MFMAs\n");
    for (int mode = 6; mode <= 7; ++mode)
        for (int threads = 256; threads <= 512; threads *= 2) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k32, dim3(256), dim3(threads), 0, 0, d_out, d_in, iters, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-50s %d waves/SIMD: %8.1f ns per iteration per wave  (%.1f ns per SIMD per 48-MFMA16-equivalent + 176 VALU)\n",
                   mode == 6 ? "24 MFMA 32x32x16 then 176 VALU" : "88 VALU, 24 MFMA 32x32x16, 88 VALU", threads / 256, ms * 1e6 / iters, ms * 1e6 / iters / (threads / 256));
        }
    return 0;
}
