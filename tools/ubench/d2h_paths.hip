// Device -> host result paths for a 600 x 600 fp64 matrix (2.88 MB): hipMemcpyAsync into pinned memory (SDMA), a copy kernel storing
// into the same pinned memory (zero-copy over PCIe), and the host memcpy out of the pinned block into pageable memory.
// build: hipcc -O3 --offload-arch=gfx950 -o d2h_paths.bin d2h_paths.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t bytes : {(size_t)360000 * 8, (size_t)4000000 * 8}) {
        void *d, *pin;
        hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
        hipHostMalloc(&pin, bytes, hipHostMallocDefault);
        char *dst = (char *)malloc(bytes);
        memset(dst, 0, bytes);
        for (int rep = 0; rep < 3; ++rep) {
            double t = now();
            hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, nullptr); hipStreamSynchronize(nullptr);
            double t1 = now();
            for (int blocks : {64, 256, 1024}) {
                double a = now();
                hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint4 *)pin, (const uint4 *)d, bytes / 16);
                hipStreamSynchronize(nullptr);
                printf("  copy kernel %4d blocks: %.0f us (%.1f GB/s)\n", blocks, now() - a, bytes / (now() - a) / 1e3);
            }
            double t2 = now();
            memcpy(dst, pin, bytes);
            double t3 = now();
            char *fresh = (char *)malloc(bytes);
            double t4 = now();
            memcpy(fresh, pin, bytes);
            double t5 = now();
            free(fresh);
            printf("%zu bytes: hipMemcpyAsync + sync %.0f us (%.1f GB/s) | memcpy pinned -> touched %.0f us (%.1f GB/s), -> fresh malloc %.0f us\n", bytes, t1 - t,
                   bytes / (t1 - t) / 1e3, t3 - t2, bytes / (t3 - t2) / 1e3, t5 - t4);
        }
        hipFree(d); hipHostFree(pin); free(dst);
    }
    return 0;
}
