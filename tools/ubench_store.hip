// Micro-benchmark (development aid): how the gain rows of a backward pass should be laid out for the
// stores.  One wave per 64-problem tile walks T steps backwards; per step it reads 10 rows of 512 B
// (three arrays) and writes 8 values per lane (the 2 x 4 gain matrix) in one of these shapes:
//   0: eight 512-byte rows, 8 bytes per lane each (plain tiled layout)
//   1: two rows of 32 bytes per lane, written as two 16-byte stores per row (lane stride 32 bytes:
//      every store instruction covers half of each of 16 lines)
//   2: four rows of 16 bytes per lane (every store instruction is 1 KiB contiguous)
//   3: as 1, but the lanes of a pair exchange halves first so that every store instruction is 1 KiB
//      contiguous (positions 0..31 of a row hold the even lanes' problems, 32..63 the odd lanes')
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o tools/ubench_store && tools/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double swap_pair(double v) { // value of the neighbouring lane (lane ^ 1)
    int lo = __builtin_amdgcn_mov_dpp((int)__double2loint(v), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    int hi = __builtin_amdgcn_mov_dpp((int)__double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int SHAPE, int PF, bool DO_READ>
__global__ __launch_bounds__(64) void pattern(const double *A0, const double *A1, const double *A2, double *W0, int T) {
    const int tile = blockIdx.x, ln = threadIdx.x;
    const double *a0 = A0 + (size_t)tile * T * 4 * 64 + ln;
    const double *a1 = A1 + (size_t)tile * T * 4 * 64 + ln;
    const double *a2 = A2 + (size_t)tile * T * 2 * 64 + ln;
    double *w0 = W0 + (size_t)tile * T * 8 * 64;
    double ring[PF][10];
    auto load = [&](int t, double (&r)[10]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = DO_READ ? a0[(t * 4 + i) * 64] : 1.0; r[4 + i] = DO_READ ? a1[(t * 4 + i) * 64] : 2.0; }
#pragma unroll
        for (int i = 0; i < 2; ++i) r[8 + i] = DO_READ ? a2[(t * 2 + i) * 64] : 3.0;
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) if (s < T) load(T - 1 - s, ring[s]);
    double acc = 0;
    for (int tb = 0; tb < T; tb += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            int step = tb + s;
            if (step >= T) break;
            int t = T - 1 - step;
            double cur[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) cur[i] = ring[s][i];
            if (step + PF < T) load(T - 1 - (step + PF), ring[s]);
            double v = acc;
#pragma unroll
            for (int i = 0; i < 10; ++i) v += cur[i];
            acc = v;
            double k[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) k[i] = v + i;
            double *w = w0 + (size_t)t * 8 * 64;
            if (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i * 64 + ln] = k[i];
            } else if (SHAPE == 1) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    v2d *p = (v2d *)(w + r * 256 + ln * 4);
                    p[0] = v2d{k[4 * r], k[4 * r + 1]};
                    p[1] = v2d{k[4 * r + 2], k[4 * r + 3]};
                }
            } else if (SHAPE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) *(v2d *)(w + r * 128 + ln * 2) = v2d{k[2 * r], k[2 * r + 1]};
            } else {
                const bool odd = ln & 1;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double a0_ = k[4 * r], a1_ = k[4 * r + 1], b0 = k[4 * r + 2], b1 = k[4 * r + 3];
                    // what this lane gives away: the even lane its second half, the odd lane its first half
                    const double y0 = swap_pair(odd ? a0_ : b0), y1 = swap_pair(odd ? a1_ : b1);
                    // first store: positions 0..31 = the even lanes' problems, lane l writes half l % 2 of position l / 2
                    v2d *p = (v2d *)(w + r * 256) + ln;
                    p[0] = odd ? v2d{y0, y1} : v2d{a0_, a1_};
                    p[64] = odd ? v2d{b0, b1} : v2d{y0, y1};
                }
            }
        }
    }
    if (acc == 12345.678) W0[0] = acc;
}

template <class F> float time_ms(F f, int reps = 7) {
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
    double *A0, *A1, *A2, *W0;
    CHECK(hipMalloc(&A0, rows * 4 * 8));
    CHECK(hipMalloc(&A1, rows * 4 * 8));
    CHECK(hipMalloc(&A2, rows * 2 * 8));
    CHECK(hipMalloc(&W0, rows * 8 * 8));
    CHECK(hipMemset(A0, 0, rows * 4 * 8));
    CHECK(hipMemset(A1, 0, rows * 4 * 8));
    CHECK(hipMemset(A2, 0, rows * 2 * 8));
    const double rd = rows * 10.0 * 8, wr = rows * 8.0 * 8;
#define RUN(SHAPE, PF, RD) { float ms = time_ms([&] { hipLaunchKernelGGL((pattern<SHAPE, PF, RD>), dim3(tiles), dim3(64), 0, 0, A0, A1, A2, W0, T); }); \
        printf("shape " #SHAPE " PF=" #PF " read=" #RD "  %8.3f ms  %7.1f GB/s\n", ms, ((RD ? rd : 0) + wr) / ms / 1e6); }
    RUN(0, 3, true); RUN(1, 3, true); RUN(2, 3, true); RUN(3, 3, true);
    RUN(0, 3, false); RUN(1, 3, false); RUN(2, 3, false); RUN(3, 3, false);
    return 0;
}
