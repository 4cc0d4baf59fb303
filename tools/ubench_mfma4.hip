// Throughput of the fp64 4x4x4 matrix instruction: CH independent accumulation chains per wave, W waves
// per SIMD (workgroups of 4 W waves, one per CU).  Prints clocks per instruction and SIMD at 2.4 GHz.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma4.hip -o /tmp/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH> __global__ void k(double *out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double c[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) c[j] = j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 12 / CH; ++q)
#pragma unroll
            for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j], 0, 0, 0);
    }
    double r = 0;
#pragma unroll
    for (int j = 0; j < CH; ++j) r += c[j];
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int CH> void run(double *out, int W) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256 * W), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("chains %d, waves per SIMD %d: %.1f clocks per instruction and SIMD\n", CH, W, best * 1e-3 * 2.4e9 / (iters * 12.0 * W));
}
int main() {
    double *out;
    hipMalloc(&out, 65536);
    for (int W = 1; W <= 4; W *= 2) {
        run<1>(out, W);
        run<2>(out, W);
        run<3>(out, W);
        run<6>(out, W);
    }
    return 0;
}
