// Do the fp64 matrix instruction and fp64 vector instructions of DIFFERENT waves of a SIMD overlap?
// Workgroups of 8 waves, one workgroup per CU: waves 0..3 and 4..7 land on SIMDs 0..3.  Mode 0: every
// wave issues independent v_mfma_f64_16x16x4; mode 1: every wave issues independent v_fma_f64; mode 2:
// waves 0..3 the matrix loop, waves 4..7 the vector loop; mode 3/4: only waves 0..3 work (matrix / vector).
// Modes 5/6: a loop of 32-bit integer vector instructions alone (waves 0..3) / beside the matrix loop;
// modes 7/8: the fp32 matrix instruction alone / beside the fp64 vector loop; 9: fp32 matrix beside fp32 vector FMAs.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <bool F32> __global__ __launch_bounds__(512) void k(double *out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    const bool mat = mode == 0 || (mode == 2 && w < 4) || (mode == 3 && w < 4);
    const bool vec = mode == 1 || (mode == 2 && w >= 4) || (mode == 4 && w < 4) || (mode == 8 && w >= 4);
    const bool ivec = (mode == 5 && w < 4) || (mode == 6 && w >= 4);
    const bool mat32 = (mode >= 7 && mode <= 9 && w < 4);
    const bool vec32 = mode == 9 && w >= 4;
    const bool mat6 = mode == 6 && w < 4;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double r = 0;
    if (mat || mat6) {
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (vec) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = a + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) // 64 fp64 FMAs = the issue time of one matrix instruction
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], b, a);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += x[j];
    }
    else if (ivec) {
        unsigned x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = threadIdx.x + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = x[j] * 3u + (x[j] >> 7); // v_mul_lo / v_lshrrev / v_add or a fused form
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += x[j];
    } else if (mat32) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float fa = (float)a, fb = (float)b;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (vec32) {
        float x[16];
        const float fa = (float)a, fb = (float)b;
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = fa + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], fb, fa);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += x[j];
    }
    if (r == 12345.678) out[threadIdx.x] = r;
}
int main() {
    double *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 10; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, out, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        // per wave: iters * 4 matrix instructions (64 clocks each) or iters * 64 FMAs (4 clocks each)
        printf("mode %d: %.3f ms  (%.1f clocks per group of 4 mfma / 64 fma at 2.4 GHz)\n", mode, best, best * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
