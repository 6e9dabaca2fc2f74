// Microbenchmark: lane-per-pair matvec on the VALU.  Every lane owns one K-vector in registers; the K x K matrix is
// streamed row by row through SGPRs (s_load from the scalar cache) and multiplied with v_pk_fma_f32; the raw dot
// products go to a per-wave LDS buffer and come back as the next source vector (x <- 1 / (M x), like a Sinkhorn half
// update).  Question: does this reach the 157 TF f32 vector peak that v_mfma_f32 cannot reach once VALU work is mixed in?
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_matvec valu_matvec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float *cptr;

template <int KP, int WAVES, int MODE>   // KP: padded vector length (multiple of 4)
__global__ void __launch_bounds__(64 * WAVES) k(const float *M, float *out, int rows, int iters) {
    extern __shared__ float lds[];                        // WAVES x KP x 64
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    float *buf = lds + wave * KP * 64;
    f2 x[KP / 2];
#pragma unroll
    for (int k = 0; k < KP / 2; ++k) x[k] = f2{1.0f + 0.001f * lane + k, 1.0f + 0.002f * lane + k};
    cptr Mc = (cptr)M;
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rows; ++r) {
            cptr row = Mc + (MODE == 1 ? 0 : (size_t)r * KP);   // MODE 1: loop-invariant row -> loads hoisted, pure VALU rate
            f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KP / 2; k += 2) {
                const f2 m0 = {row[2 * k], row[2 * k + 1]}, m1 = {row[2 * k + 2], row[2 * k + 3]};
                acc0 = __builtin_elementwise_fma(m0, x[k], acc0);
                acc1 = __builtin_elementwise_fma(m1, x[k + 1], acc1);
            }
            acc0 += acc1;
            buf[(r >> 2) * 256 + lane * 4 + (r & 3)] = acc0[0] + acc0[1];
        }
        // back into registers: x <- 1 / dot
#pragma unroll
        for (int k = 0; k < KP / 4; ++k) {
            const f4 d = *reinterpret_cast<const f4 *>(buf + k * 256 + lane * 4);
            x[2 * k] = f2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            x[2 * k + 1] = f2{__builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KP / 2; ++k) s += x[k][0] + x[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KP, int WAVES, int MODE = 0> void run(int wgs_per_cu, int rows) {
    float *M, *out;
    hipMalloc(&M, KP * KP * sizeof(float));
    float *h = (float *)malloc(KP * KP * sizeof(float));
    for (int i = 0; i < KP * KP; ++i) h[i] = 0.5f + (i % 7) * 0.1f;
    hipMemcpy(M, h, KP * KP * sizeof(float), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    const int iters = 400;
    const size_t ldsz = sizeof(float) * WAVES * KP * 64;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<KP, WAVES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KP, WAVES, MODE><<<256 * wgs_per_cu, 64 * WAVES, ldsz>>>(M, out, rows, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KP, WAVES, MODE><<<256 * wgs_per_cu, 64 * WAVES, ldsz>>>(M, out, rows, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = 256.0 * wgs_per_cu * WAVES;
    const double flops = waves * 64 * iters * (double)rows * rows * 2;          // useful flops (rows x rows)
    const double cyc_per_row = ms * 1e-3 * 2.4e9 / (iters * (double)rows) / (wgs_per_cu * WAVES / 4.0);
    printf("MODE=%d KP=%d rows=%d waves/WG=%d WG/CU=%d (%.1f waves/SIMD): %.3f ms  %.1f TF useful  %.1f SIMD-cycles per row per wave (%d pk_fma)\n",
           MODE, KP, rows, WAVES, wgs_per_cu, wgs_per_cu * WAVES / 4.0, ms, flops / (ms * 1e-3) / 1e12, cyc_per_row, KP / 2);
    hipFree(M); hipFree(out); free(h);
}

int main() {
    run<52, 4>(1, 50);
    run<52, 4>(2, 50);
    run<52, 4>(3, 50);
    run<52, 4, 1>(1, 50);
    run<52, 4, 1>(2, 50);
    run<52, 4, 1>(3, 50);
    run<32, 4>(4, 30);
    run<32, 4, 1>(1, 30);
    run<32, 4, 1>(4, 30);
    return 0;
}
