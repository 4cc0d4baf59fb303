// Micro-benchmark of the memory access shapes used by the solver kernels (development aid).
// One wave per 64-problem tile walks T time steps; per step it reads R rows of 512 B and writes
// W rows of 512 B, from/to either several separate arrays or one merged record array.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// separate arrays: A0 (4 rows/step), A1 (4 rows/step), A2 (2 rows/step) read; W0 (8 rows/step), W1 (2 rows/step) written
template <int NT> __device__ __forceinline__ void st(double *p, double v) {
    if (NT & 1) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int NT> __device__ __forceinline__ double ld(const double *p) {
    return (NT & 2) ? __builtin_nontemporal_load(p) : *p;
}
// NT bit 0: nontemporal stores, bit 1: nontemporal loads; DO_READ false: write-only
template <int PF, bool REVERSE, bool DO_WRITE, int FLOPS, int NT = 0, bool DO_READ = true>
__global__ __launch_bounds__(64) void pattern_separate(const double *A0, const double *A1, const double *A2, double *W0, double *W1, int T) {
    const int tile = blockIdx.x, ln = threadIdx.x;
    const double *a0 = A0 + (size_t)tile * T * 4 * 64 + ln;
    const double *a1 = A1 + (size_t)tile * T * 4 * 64 + ln;
    const double *a2 = A2 + (size_t)tile * T * 2 * 64 + ln;
    double *w0 = W0 + (size_t)tile * T * 8 * 64 + ln;
    double *w1 = W1 + (size_t)tile * T * 2 * 64 + ln;
    double ring[PF][10];
    auto load = [&](int t, double (&r)[10]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = DO_READ ? ld<NT>(a0 + (t * 4 + i) * 64) : 1.0; r[4 + i] = DO_READ ? ld<NT>(a1 + (t * 4 + i) * 64) : 2.0; }
#pragma unroll
        for (int i = 0; i < 2; ++i) r[8 + i] = DO_READ ? ld<NT>(a2 + (t * 2 + i) * 64) : 3.0;
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) { int t = REVERSE ? T - 1 - s : s; if (s < T) load(t, ring[s]); }
    double acc = 0;
    for (int tb = 0; tb < T; tb += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            int step = tb + s;
            if (step >= T) break;
            int t = REVERSE ? T - 1 - step : step;
            double cur[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) cur[i] = ring[s][i];
            int nstep = step + PF;
            if (nstep < T) load(REVERSE ? T - 1 - nstep : nstep, ring[s]);
            double v = acc;
#pragma unroll
            for (int i = 0; i < 10; ++i) v += cur[i];
#pragma unroll
            for (int f = 0; f < FLOPS; ++f) v = __builtin_fma(v, 1.0000001, 0.5);   // dependent chain
            acc = v;
            if (DO_WRITE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) st<NT>(w0 + (t * 8 + i) * 64, v + i);
#pragma unroll
                for (int i = 0; i < 2; ++i) st<NT>(w1 + (t * 2 + i) * 64, v - i);
            }
        }
    }
    if (acc == 12345.678) W1[0] = acc;
}

// merged: one read array with 10 rows/step, one write array with 10 rows/step
template <int PF, bool REVERSE, bool DO_WRITE, int FLOPS>
__global__ __launch_bounds__(64) void pattern_merged(const double *A, double *W, int T) {
    const int tile = blockIdx.x, ln = threadIdx.x;
    const double *a = A + (size_t)tile * T * 10 * 64 + ln;
    double *w = W + (size_t)tile * T * 10 * 64 + ln;
    double ring[PF][10];
    auto load = [&](int t, double (&r)[10]) {
#pragma unroll
        for (int i = 0; i < 10; ++i) r[i] = a[(t * 10 + i) * 64];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) { int t = REVERSE ? T - 1 - s : s; if (s < T) load(t, ring[s]); }
    double acc = 0;
    for (int tb = 0; tb < T; tb += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            int step = tb + s;
            if (step >= T) break;
            int t = REVERSE ? T - 1 - step : step;
            double cur[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) cur[i] = ring[s][i];
            int nstep = step + PF;
            if (nstep < T) load(REVERSE ? T - 1 - nstep : nstep, ring[s]);
            double v = acc;
#pragma unroll
            for (int i = 0; i < 10; ++i) v += cur[i];
#pragma unroll
            for (int f = 0; f < FLOPS; ++f) v = __builtin_fma(v, 1.0000001, 0.5);
            acc = v;
            if (DO_WRITE) {
#pragma unroll
                for (int i = 0; i < 10; ++i) w[(t * 10 + i) * 64] = v + i;
            }
        }
    }
    if (acc == 12345.678) W[0] = acc;
}

__global__ void copy4(const double4 *a, double4 *b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}

template <class F> float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        f();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int tiles = 1024, T = 199;
    const size_t rows = (size_t)tiles * T * 64;
    double *A0, *A1, *A2, *W0, *W1, *AM, *WM;
    CHECK(hipMalloc(&A0, rows * 4 * 8));
    CHECK(hipMalloc(&A1, rows * 4 * 8));
    CHECK(hipMalloc(&A2, rows * 2 * 8));
    CHECK(hipMalloc(&W0, rows * 8 * 8));
    CHECK(hipMalloc(&W1, rows * 2 * 8));
    CHECK(hipMalloc(&AM, rows * 10 * 8));
    CHECK(hipMalloc(&WM, rows * 10 * 8));
    CHECK(hipMemset(A0, 0, rows * 4 * 8));
    CHECK(hipMemset(A1, 0, rows * 4 * 8));
    CHECK(hipMemset(A2, 0, rows * 2 * 8));
    CHECK(hipMemset(AM, 0, rows * 10 * 8));
    const double rd = rows * 10.0 * 8, wr = rows * 10.0 * 8;
    auto report = [&](const char *name, float ms, bool w) {
        printf("%-46s %8.3f ms  %7.1f GB/s\n", name, ms, (rd + (w ? wr : 0)) / ms / 1e6);
    };
#define RUN_SEP(PF, REV, WR, FL) report("separate PF=" #PF " rev=" #REV " write=" #WR " flops=" #FL, \
        time_ms([&] { hipLaunchKernelGGL((pattern_separate<PF, REV, WR, FL>), dim3(tiles), dim3(64), 0, 0, A0, A1, A2, W0, W1, T); }), WR)
#define RUN_MRG(PF, REV, WR, FL) report("merged   PF=" #PF " rev=" #REV " write=" #WR " flops=" #FL, \
        time_ms([&] { hipLaunchKernelGGL((pattern_merged<PF, REV, WR, FL>), dim3(tiles), dim3(64), 0, 0, AM, WM, T); }), WR)
#define RUN_NT(PF, NT, RD) { float ms_ = time_ms([&] { hipLaunchKernelGGL((pattern_separate<PF, true, true, 0, NT, RD>), dim3(tiles), dim3(64), 0, 0, A0, A1, A2, W0, W1, T); }); \
        printf("separate PF=" #PF " nt=" #NT " read=" #RD " %8.3f ms  %7.1f GB/s\n", ms_, ((RD ? rd : 0) + wr) / ms_ / 1e6); }
    RUN_NT(3, 1, true);
    RUN_NT(3, 2, true);
    RUN_NT(3, 3, true);
    RUN_NT(3, 0, false);
    RUN_NT(3, 1, false);
    RUN_SEP(1, true, true, 0);
    RUN_SEP(3, true, true, 0);
    RUN_SEP(8, true, true, 0);
    RUN_SEP(3, false, true, 0);
    RUN_SEP(3, true, false, 0);
    RUN_SEP(8, true, false, 0);
    RUN_MRG(3, true, true, 0);
    RUN_MRG(8, true, true, 0);
    RUN_MRG(3, false, true, 0);
    RUN_MRG(8, true, false, 0);
    RUN_SEP(3, true, true, 64);
    RUN_SEP(3, true, true, 128);
    RUN_SEP(3, true, true, 256);
    RUN_SEP(1, true, true, 256);
    RUN_SEP(3, true, false, 256);
    {
        size_t n = rows * 10 * 8 / 32;
        float ms = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, (const double4 *)AM, (double4 *)WM, n); });
        report("linear copy double4, grid-stride", ms, true);
    }
    return 0;
}
