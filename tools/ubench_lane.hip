// Cross-lane micro-benchmark (development aid): latency of a dependent chain on gfx950, one
// wave per SIMD, for the building blocks of a several-lanes-per-problem Riccati step:
//   fma                 v = fma(v, a, b)
//   fma + quad dpp      v = fma(quad_broadcast(v, 0), a, b)       (two v_mov_b32 dpp per fp64)
//   fma + bpermute      v = fma(shfl(v, src), a, b)               (two ds_bpermute_b32 per fp64)
//   mfma 4x4x4          D = mfma_f64_4x4x4(D', b, c)              (dependent through the A operand)
//   division            v = 1 / v
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double quad_bcast0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); // quad_perm:[0,0,0,0]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int MODE> __global__ __launch_bounds__(64) void chain(double *out, int iters, double a, double b) {
    double v = threadIdx.x * 1e-3 + 1.0;
    const int src = (threadIdx.x ^ 5) << 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) v = __builtin_fma(v, a, b);
            if (MODE == 1) v = __builtin_fma(quad_bcast0(v), a, b);
            if (MODE == 2) {
                int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
                int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
                v = __builtin_fma(__hiloint2double(hi, lo), a, b);
            }
            if (MODE == 3) v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, a, b, 0, 0, 0);
            if (MODE == 4) v = 1.0 / v + a;
            if (MODE == 5) v = __builtin_amdgcn_rcp(v) + a;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v;
}

template <int MODE> void run(double *out, const char *name) {
    const int blocks = 1024, iters = 2048;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((chain<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)iters * 16);
    printf("%-16s %.2f ns per link (%.1f cycles at 2.4 GHz)\n", name, ns, ns * 2.4);
}

int main() {
    double *out;
    CHECK(hipMalloc(&out, 1024 * 64 * 8));
    run<0>(out, "fma");
    run<1>(out, "fma+quad dpp");
    run<2>(out, "fma+bpermute");
    run<3>(out, "mfma 4x4x4");
    run<4>(out, "div+add");
    run<5>(out, "rcp+add");
    return 0;
}
