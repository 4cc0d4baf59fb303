// ALU micro-benchmark (development aid): fp64 FMA dependent-chain latency and issue rate on
// gfx950 at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ILP>
__global__ __launch_bounds__(64) void fma_chain(double *out, int iters, double a, double b) {
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP> void run(double *out, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fma_chain<ILP>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0000001, 0.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((fma_chain<ILP>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0000001, 0.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fmas_per_wave = (double)iters * 16 * ILP;
    const double ns_per_fma_per_simd = ms * 1e6 / (fmas_per_wave * waves_per_simd);
    printf("ILP=%d waves/SIMD=%d: %.3f ms, %.2f ns per wave-FMA per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", ILP, waves_per_simd, ms,
           ns_per_fma_per_simd, ns_per_fma_per_simd * 2.4, fmas_per_wave * blocks * 64 * 2 / (ms * 1e-3) / 1e12);
}

int main() {
    double *out;
    CHECK(hipMalloc(&out, 1024 * 8 * 64 * 8));
    for (int w : {1, 2, 4}) {
        run<1>(out, w);
        run<2>(out, w);
        run<4>(out, w);
        run<8>(out, w);
    }
    return 0;
}
