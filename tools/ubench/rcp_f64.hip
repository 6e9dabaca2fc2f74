// f64 arithmetic of the Jensen-Shannon branch of cost_matrix_kernel against the host's, on the rows tools/fuzz_prepass.py found
// (scipy multiplies by the reciprocal of a row's sum; a sum that rounds below zero gives NaN).  build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *X, int D, double *out) {
    const double *u = X, *v = X + D;
    double su = 0.0, sv = 0.0;
    for (int d = 0; d < D; ++d) { su += u[d]; sv += v[d]; }
    const double ru = 1.0 / su, rv = 1.0 / sv;
    double js = 0.0;
    for (int d = 0; d < D; ++d) {
        const double p = u[d] * ru, q = v[d] * rv, m = (p + q) / 2.0;
        if (p > 0.0) js += p * log(p / m);
        if (q > 0.0) js += q * log(q / m);
        out[2] = p; out[3] = q; out[4] = m; out[5] = log(p / m); out[6] = log(q / m);
    }
    out[0] = js; out[1] = sqrt(js / 2.0);
}
int main() {
    double h[2] = {0.042934618384080525, 0.15078800451679664};
    double *d, *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 8 * 8); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, 1, o);
    double r[8]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("gpu: js=%.17g out=%.17g p=%.17g q=%.17g m=%.17g log(p/m)=%.17g log(q/m)=%.17g\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
    return 0;
}
