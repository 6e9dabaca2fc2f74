// Issue cost of the vector instructions the Sinkhorn kernels' element-wise code is made of, on gfx950: cycles per instruction
// of one wave (8 independent chains, or one dependent chain), with 1 and 2 waves per SIMD.
// build: hipcc -O2 --offload-arch=gfx950 -o valu_rates.bin valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, INDEP, DEP)                                                                        \
    __global__ void k_##NAME(float *out, const float *in, int iters, int dep, long long *cyc) {         \
        float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b = in[64 + threadIdx.x], c = in[128 + threadIdx.x];                                      \
        const unsigned long long smask = __ballot(b > 1.03f);                                           \
        unsigned long long p64[8] = {1, 2, 3, 4, 5, 6, 7, 8}, pb64 = threadIdx.x;                      \
        const long long t0 = wall_clock64();                                                            \
        if (!dep) { for (int i = 0; i < iters; ++i) { INDEP INDEP INDEP INDEP } }                       \
        else { for (int i = 0; i < iters; ++i) { DEP DEP DEP DEP } }                                    \
        const long long t1 = wall_clock64();                                                            \
        out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(p64[0] + p64[1] + p64[2] + p64[3] + p64[4] + p64[5] + p64[6] + p64[7]); \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                      \
    }
#define A(i) a##i
#define P(i) p64[i]
#define I_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A(i)) : "v"(b));
#define D_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
#define I_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
#define I_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(A(i)));
#define D_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));
#define I_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(A(i)));
#define D_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a0));
#define I_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(A(i)) : "v"(b));
#define D_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
#define I_CVT(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(A(i)));
#define D_CVT(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a0));
#define I_MIX(i) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(A(i)) : "v"(b));
#define D_MIX(i) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(b));
#define I_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0" : "+v"(a0) : "v"(b));
#define I_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %0, %1, 0" : "+v"(a0) : "v"(b));
#define I_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
#define I_PKMAX3(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_PKMAX3(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
#define I_PKMAX(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(A(i)) : "v"(b));
#define D_PKMAX(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a0) : "v"(b));
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A(i)) : "v"(b));
#define D_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b));
#define I_CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(A(i)) : "v"(b), "s"(smask));
#define D_CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "s"(smask));
#define I_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(A(i)) : "v"(b), "v"(c));
#define D_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
#define I_ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(P(i)) : "v"(pb64));
#define D_ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(P(0)) : "v"(pb64));
#define I_PERM32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(A(i)), "+v"(b));
#define D_PERM32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b));

KERNEL(mul, REP8(I_MUL), REP8(D_MUL))
KERNEL(fma, REP8(I_FMA), REP8(D_FMA))
KERNEL(rcp, REP8(I_RCP), REP8(D_RCP))
KERNEL(exp, REP8(I_EXP), REP8(D_EXP))
KERNEL(cvt_pk_f16, REP8(I_CVTPK), REP8(D_CVTPK))
KERNEL(cvt_f32_f16, REP8(I_CVT), REP8(D_CVT))
KERNEL(fma_mix_f32, REP8(I_MIX), REP8(D_MIX))
KERNEL(fma_mixlo_f16, REP8(I_MIXLO), REP8(D_MIXLO))
KERNEL(fma_mixhi_f16, REP8(I_MIXHI), REP8(D_MIXHI))
KERNEL(max3_f32, REP8(I_MAX3), REP8(D_MAX3))
KERNEL(pk_maximum3_f16, REP8(I_PKMAX3), REP8(D_PKMAX3))
KERNEL(pk_max_f16, REP8(I_PKMAX), REP8(D_PKMAX))
KERNEL(cndmask, REP8(I_CNDMASK), REP8(D_CNDMASK))
KERNEL(permlane32_swap, REP8(I_PERM32), REP8(D_PERM32))
KERNEL(cndmask_sgpr, REP8(I_CNDS), REP8(D_CNDS))
KERNEL(and_or, REP8(I_ANDOR), REP8(D_ANDOR))
KERNEL(lshl_add_u64, REP8(I_ADD64), REP8(D_ADD64))

// packed f32 on register pairs, and mixes of a transcendental with plain instructions (does v_rcp_f32 leave issue slots?)
#define KERNEL2(NAME, BODY)                                                                             \
    __global__ void k2_##NAME(float *out, const float *in, int iters, int dep, long long *cyc) {        \
        using f2 = float __attribute__((ext_vector_type(2)));                                           \
        f2 p0 = {in[threadIdx.x], in[threadIdx.x + 1]}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f;    \
        const f2 pb = {in[64 + threadIdx.x], in[65 + threadIdx.x]};                                     \
        float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b = in[64 + threadIdx.x];                                                                 \
        const long long t0 = wall_clock64();                                                            \
        for (int i = 0; i < iters; ++i) { BODY BODY BODY BODY }                                         \
        const long long t1 = wall_clock64();                                                            \
        out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];      \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                      \
    }
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pb));
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(pb));
#define RCPX(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define MULX(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MIXX(x) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(b));
KERNEL2(pk_mul, PKMUL(p0) PKMUL(p1) PKMUL(p2) PKMUL(p3) PKMUL(p0) PKMUL(p1) PKMUL(p2) PKMUL(p3))                       // 8 per body
KERNEL2(pk_fma, PKFMA(p0) PKFMA(p1) PKFMA(p2) PKFMA(p3) PKFMA(p0) PKFMA(p1) PKFMA(p2) PKFMA(p3))
KERNEL2(rcp1_mul1, RCPX(a0) MULX(a4) RCPX(a1) MULX(a5) RCPX(a2) MULX(a6) RCPX(a3) MULX(a7))                          // 4 + 4
KERNEL2(rcp1_mul3, RCPX(a0) MULX(a4) MULX(a5) MULX(a6) RCPX(a1) MULX(a7) MULX(a4) MULX(a5))                          // 2 + 6
KERNEL2(rcp4_then_mul4, RCPX(a0) RCPX(a1) RCPX(a2) RCPX(a3) MULX(a4) MULX(a5) MULX(a6) MULX(a7))                      // 4 + 4, blocked
KERNEL2(rcp1_mix1, RCPX(a0) MIXX(a4) RCPX(a1) MIXX(a5) RCPX(a2) MIXX(a6) RCPX(a3) MIXX(a7))

int main() {
    float *d_in, *d_out; long long *d_c;
    hipMalloc(&d_in, 4096); hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_c, 8);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0f + 1e-3f * i;
    hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    // reference: the clock of wall_clock64 is 100 MHz; shader clock from the mul kernel is printed as the unit
    struct { const char *name; void (*fn)(float *, const float *, int, int, long long *); } ks[] = {
        {"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_rcp_f32", k_rcp}, {"v_exp_f32", k_exp}, {"v_cvt_pk_f16_f32", k_cvt_pk_f16},
        {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_fma_mix_f32", k_fma_mix_f32}, {"v_fma_mixlo_f16", k_fma_mixlo_f16},
        {"v_fma_mixhi_f16", k_fma_mixhi_f16}, {"v_max3_f32", k_max3_f32}, {"v_pk_maximum3_f16", k_pk_maximum3_f16},
        {"v_pk_max_f16", k_pk_max_f16}, {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cndmask_b32 (sgpr pair)", k_cndmask_sgpr}, {"v_and_or_b32", k_and_or},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_permlane32_swap", k_permlane32_swap}};
    double unit = 0;
    printf("%-22s %10s %10s %10s   (ns per instruction of one wave; independent x8 / dependent chain; 1 and 2 waves per SIMD)\n", "instruction", "indep w1", "dep w1", "indep w2");
    for (auto &k : ks) {
        double r[3];
        for (int m = 0; m < 3; ++m) {
            const int threads = m == 2 ? 512 : 256, dep = m == 1;
            long long c;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d_out, d_in, iters, dep, d_c);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
            r[m] = c * 10.0 / (double(iters) * 32);       // ns per instruction
        }
        if (!strcmp(k.name, "v_mul_f32")) unit = r[0] / 4.0;
        printf("%-22s %10.2f %10.2f %10.2f   = %.1f / %.1f / %.1f cycles (v_mul_f32 independent := 4)\n", k.name, r[0], r[1], r[2], r[0] / unit, r[1] / unit, r[2] / unit);
    }
    struct { const char *name; void (*fn)(float *, const float *, int, int, long long *); const char *what; } k2s[] = {
        {"v_pk_mul_f32", k2_pk_mul, "8 packed"}, {"v_pk_fma_f32", k2_pk_fma, "8 packed"}, {"rcp,mul alternating", k2_rcp1_mul1, "4 rcp + 4 mul"},
        {"rcp,mul,mul,mul", k2_rcp1_mul3, "2 rcp + 6 mul"}, {"4 rcp then 4 mul", k2_rcp4_then_mul4, "4 rcp + 4 mul"}, {"rcp,fma_mix alternating", k2_rcp1_mix1, "4 rcp + 4 fma_mix"}};
    printf("\nbodies of 8 instructions: cycles per BODY of one wave, 1 and 2 waves per SIMD (v_mul_f32 := 4 cycles)\n");
    for (auto &k : k2s) {
        double r[2];
        for (int m = 0; m < 2; ++m) {
            long long c;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k.fn, dim3(1), dim3(m ? 512 : 256), 0, 0, d_out, d_in, iters, 0, d_c); hipDeviceSynchronize(); }
            hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
            r[m] = c * 10.0 / (double(iters) * 4);        // ns per body
        }
        printf("%-26s (%-18s) %8.1f %8.1f cycles per body\n", k.name, k.what, r[0] / unit, r[1] / unit);
    }
    return 0;
}
