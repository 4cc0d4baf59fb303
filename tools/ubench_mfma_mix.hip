// What issues beside an fp64 matrix instruction of ANOTHER wave of the SIMD?  Workgroups of 8 waves (two
// per SIMD), one per CU.  Waves 0..3 run a loop of independent v_mfma_f64_16x16x4; waves 4..7 run a loop of
// (a) s_add/s_xor scalar instructions, (b) ds_read_b64, (c) nothing.  Each loop is also timed alone.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_mix.hip -o /tmp/ubench_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double *out, int iters, int mat_on, int other) {
    __shared__ double sh[4096];
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) sh[i] = i;
    __syncthreads();
    double r = 0;
    if (w < 4) {
        if (!mat_on) return;
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (other == 1) { // scalar: 64 dependent-free scalar instructions per trip
        unsigned x0 = iters, x1 = 1, x2 = 2, x3 = 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                asm volatile("s_add_u32 %0, %0, 3\n s_xor_b32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_xor_b32 %3, %3, 9" : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3) : : "scc");
        }
        r = x0 + x1 + x2 + x3;
    } else if (other == 2) { // LDS: 16 independent 8-byte reads per trip
        const double *p = sh + (threadIdx.x & 63);
        double acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(p) ), "n"(j * 512));
                acc[j] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += acc[j];
    }
    if (r == 12345.678) out[threadIdx.x] = r;
}
int main() {
    double *out;
    hipMalloc(&out, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    const char *name[3] = {"nothing", "64 scalar instructions", "16 ds_read_b64"};
    for (int other = 0; other < 3; ++other)
        for (int mat = 0; mat < 2; ++mat) {
            if (!mat && !other) continue;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mat, other);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("matrix loop %s, other waves: %-24s %.1f clocks per trip (4 matrix instructions = 256)\n", mat ? "on, " : "off,", name[other], best * 1e-3 * 2.4e9 / iters);
        }
    return 0;
}
