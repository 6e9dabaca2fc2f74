// Microbenchmark: do a bf16-MFMA wave and a VALU wave on the SAME SIMD overlap?  512-thread workgroups (waves w and w + 4
// share a SIMD), one workgroup per CU.  mode bit 0: waves 0-3 run an MFMA loop; bit 1: waves 4-7 run a VALU loop; bit 2: waves
// 4-7 at s_setprio 1 / bit 3: waves 0-3 at s_setprio 1; PHASED: every wave alternates blocks of 48 MFMAs and 128 VALU, waves
// 4-7 starting with the other block.  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_two_waves mfma_valu_two_waves.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = float __attribute__((ext_vector_type(4)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using u4 = unsigned int __attribute__((ext_vector_type(4)));

__device__ inline void mfma_block(f4 (&acc)[4], const u4 &a, const u4 &b) {
#pragma unroll
    for (int i = 0; i < 48; ++i)
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[i & 3], 0, 0, 0);
}
__device__ inline void valu_block(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 128; ++i) v[i & 7] = fmaf(v[i & 7], v[(i + 3) & 7], 0.5f);
}

__global__ void __launch_bounds__(512) k(float *out, int iters, int mode) {
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    u4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + i * 1e-3f + threadIdx.x * 1e-6f;
    const bool second = threadIdx.x >= 256;
    if ((mode & 4) && second) __builtin_amdgcn_s_setprio(1);
    if ((mode & 8) && !second) __builtin_amdgcn_s_setprio(1);
    if (mode & 16) {            // phased: both kinds of work in every wave, the halves in anti-phase
        if (second) valu_block(v);
        for (int it = 0; it < iters; ++it) {
            mfma_block(acc, a, b);
            __builtin_amdgcn_sched_barrier(0);
            valu_block(v);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode & 32) {     // phased, priority raised for the MFMA block
        if (second) valu_block(v);
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_setprio(1);
            mfma_block(acc, a, b);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            valu_block(v);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode & 64) {     // phased, priority raised for the VALU block
        if (second) valu_block(v);
        for (int it = 0; it < iters; ++it) {
            mfma_block(acc, a, b);
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            valu_block(v);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        if (!second && (mode & 1)) for (int it = 0; it < iters; ++it) { mfma_block(acc, a, b); __builtin_amdgcn_sched_barrier(0); }
        if (second && (mode & 2)) for (int it = 0; it < iters; ++it) { valu_block(v); __builtin_amdgcn_sched_barrier(0); }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void run(const char *name, int mode) {
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, 10, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %.3f ms  = %.0f nominal cycles (2.4 GHz) per loop iteration (48 MFMA and/or 128 VALU)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(out);
}

int main() {
    run("waves 0-3: 48 MFMA per iteration, waves 4-7 idle", 1);
    run("waves 4-7: 128 VALU per iteration, waves 0-3 idle", 2);
    run("both (separate waves on one SIMD)", 3);
    run("both, VALU waves at s_setprio 1", 3 | 4);
    run("both, MFMA waves at s_setprio 1", 3 | 8);
    run("every wave MFMA block then VALU block, halves in anti-phase", 16);
    run("  ... MFMA block at s_setprio 1", 32);
    run("  ... VALU block at s_setprio 1", 64);
    return 0;
}
