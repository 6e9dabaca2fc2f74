// Does v_mfma_f32_16x16x16_{f16,bf16} on the LOW four k-slots of the 16x16x32 operand registers give what the 16x16x32 instruction gives
// when the upper four slots are zero?  (The 16-wide tail k-block of odd row-tile counts, sinkhorn_kernels.hpp.)  Also in a chain behind
// a 16x16x32 on the same accumulator.   hipcc --offload-arch=gfx950 -O2 -o mfma_tail16 mfma_tail16.hip && ./mfma_tail16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using s16x4 = short __attribute__((ext_vector_type(4)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));

__global__ void k(const u32x4 *A, const u32x4 *B, const u32x4 *A2, const u32x4 *B2, f32x4 *out) {
    const int l = threadIdx.x;
    u32x4 a = A[l], b = B[l], a2 = A2[l], b2 = B2[l];
    u32x4 az = a, bz = b;  az[2] = az[3] = 0u; bz[2] = bz[3] = 0u;          // upper four k-slots zero
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // f16
    f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, az), __builtin_bit_cast(f16x8, bz), z, 0, 0, 0);
    const u32x2 alo = {a[0], a[1]}, blo = {b[0], b[1]};
    f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, alo), __builtin_bit_cast(f16x4, blo), z, 0, 0, 0);
    // chain: full x32 on (a2, b2), then the tail on the same accumulator
    f32x4 c32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, b2), z, 0, 0, 0);
    c32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, az), __builtin_bit_cast(f16x8, bz), c32, 0, 0, 0);
    f32x4 c16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, b2), z, 0, 0, 0);
    c16 = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, alo), __builtin_bit_cast(f16x4, blo), c16, 0, 0, 0);
    // bf16 (same bits reinterpreted: any finite patterns do)
    f32x4 q32 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, az), __builtin_bit_cast(bf16x8, bz), z, 0, 0, 0);
    f32x4 q16 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, alo), __builtin_bit_cast(s16x4, blo), z, 0, 0, 0);
    out[l] = r32; out[64 + l] = r16; out[128 + l] = c32; out[192 + l] = c16; out[256 + l] = q32; out[320 + l] = q16;
}

int main() {
    const int n = 64 * 4;
    unsigned int hA[n], hB[n], hA2[n], hB2[n];
    srand(1);
    auto h16 = [](float x) { _Float16 h = (_Float16)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return (unsigned int)u; };
    for (int i = 0; i < n; ++i) {
        auto r = []() { return (rand() % 2001 - 1000) / 500.0f; };
        hA[i] = h16(r()) | (h16(r()) << 16); hB[i] = h16(r()) | (h16(r()) << 16);
        hA2[i] = h16(r()) | (h16(r()) << 16); hB2[i] = h16(r()) | (h16(r()) << 16);
    }
    u32x4 *dA, *dB, *dA2, *dB2; f32x4 *dO;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dA2, sizeof(hA)); hipMalloc(&dB2, sizeof(hB)); hipMalloc(&dO, 6 * 64 * sizeof(f32x4));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dA2, hA2, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB2, hB2, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dB2, dO);
    float o[6 * 256];
    hipMemcpy(o, dO, sizeof(o), hipMemcpyDeviceToHost);
    const char *names[3] = {"f16  x16 vs x32 with zero upper slots", "f16  chain x32 -> x16 vs x32 -> x32", "bf16 x16 vs x32 with zero upper slots"};
    for (int t = 0; t < 3; ++t) {
        double md = 0, mx = 0; int nd = 0;
        for (int i = 0; i < 256; ++i) { const double d = fabs((double)o[(2 * t) * 256 + i] - o[(2 * t + 1) * 256 + i]); md = d > md ? d : md; nd += d != 0; mx = fmax(mx, fabs(o[(2 * t) * 256 + i])); }
        printf("%s: max |diff| %.3e (max |value| %.3f), %d of 256 outputs differ\n", names[t], md, mx, nd);
    }
    return 0;
}
