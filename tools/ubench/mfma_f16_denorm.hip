// Does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs?  (The fp16-split Sinkhorn kernel relies on low pieces that can be
// subnormal.)  And: v_fma_mix_f32 residual + v_cvt_pk_f16_f32 rounding mode check.
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_f16_denorm.bin mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using f16x8_t = _Float16 __attribute__((ext_vector_type(8)));
using f32x4_t = float __attribute__((ext_vector_type(4)));
using f16x2_t = _Float16 __attribute__((ext_vector_type(2)));
using f32x2_t = float __attribute__((ext_vector_type(2)));

__global__ void k(float a_val, float b_val, float *out) {
    f16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // element k=0 of lane group 0 only: A[row][0] = a_val for all rows, B[0][col] = b_val
    if (threadIdx.x < 16) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f32x4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void k2(const float *x, float *o) {
    float x0 = x[0], x1 = x[1];
    f32x2_t v = {x0, x1};
    unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
    o[0] = (float)hv[0]; o[1] = (float)hv[1]; o[2] = r0; o[3] = r1;
}
int main() {
    float *d; hipMalloc(&d, 64);
    float h[4];
    const float cases[][2] = {{1.f, 1.f}, {3.0517578125e-05f /*2^-15 subnormal*/, 1024.f}, {1024.f, 3.0517578125e-05f},
                              {5.9604644775390625e-08f /*2^-24 min subnormal*/, 32768.f}, {32768.f, 5.9604644775390625e-08f},
                              {6.103515625e-05f /*2^-14 min normal*/, 1.f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%.10g b=%.10g  mfma=%.10g  expected=%.10g %s\n", c[0], c[1], h[0], c[0] * c[1], h[0] == c[0] * c[1] ? "ok" : "FLUSHED/DIFFERENT");
    }
    float x[2] = {1.00048828125f + 0.000244140625f * 0.5f /*tie*/, 3.14159274f}, *dx;
    hipMalloc(&dx, 8); hipMemcpy(dx, x, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k2, dim3(1), dim3(1), 0, 0, dx, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("x0=%.10g -> h=%.10g r=%.10g (x-h=%.10g)\nx1=%.10g -> h=%.10g r=%.10g (x-h=%.10g)\n", x[0], h[0], h[2], x[0] - h[0], x[1], h[1], h[3], x[1] - h[1]);
    return 0;
}
