// Does the distance between the tiles of an array matter to HBM?  1024 waves, one per tile, each streams
// T steps of R rows of 512 bytes (lane = 8 bytes) from (or to) its tile; the tiles lie `stride` bytes apart:
// T R 512 (what the solver's arrays have) plus a pad.  Prints TB/s per pad.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_stride.hip -o /tmp/ubench_stride
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, bool STORE> __global__ __launch_bounds__(64) void k(double *buf, size_t stride_elems, int T) {
    double *base = buf + (size_t)blockIdx.x * stride_elems + threadIdx.x;
    double acc = 0;
    constexpr int PF = 4;
    double ring[PF][R];
    if (!STORE) {
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int i = 0; i < R; ++i) ring[s][i] = base[(size_t)(s * R + i) * 64];
    }
    for (int tb = 0; tb < T; tb += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = tb + s;
            if (t >= T) break;
            if (STORE) {
#pragma unroll
                for (int i = 0; i < R; ++i) base[(size_t)(t * R + i) * 64] = 1.0 * t + i;
            } else {
                double v = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) v += ring[s][i];
                acc += v;
                const int tn = t + PF < T ? t + PF : T - 1;
#pragma unroll
                for (int i = 0; i < R; ++i) ring[s][i] = base[(size_t)(tn * R + i) * 64];
            }
        }
    }
    if (!STORE && acc == 12345.678) buf[0] = acc;
}
template <int R, bool STORE> void run(double *buf, int T) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t tile = (size_t)T * R * 64;
    for (size_t pad : {0ul, 64ul, 192ul, 512ul, 576ul, 1088ul, 4160ul, 16448ul}) { // elements: 0, 512 B, 1.5 KiB, 4 KiB, 4.5 KiB, 8.5 KiB, 32.5 KiB, 128.5 KiB
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<R, STORE>), dim3(1024), dim3(64), 0, 0, buf, tile + pad, T);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%s R=%2d rows/step, tile %zu B + pad %6zu B: %.3f ms, %.2f TB/s\n", STORE ? "store" : "load ", R, tile * 8, pad * 8, best,
               1024.0 * tile * 8 / (best * 1e-3) / 1e12);
    }
}
int main() {
    double *buf;
    const int T = 199;
    hipMalloc(&buf, 1024ull * ((size_t)T * 16 * 64 + 20000) * 8);
    hipMemset(buf, 0, 1024ull * ((size_t)T * 16 * 64 + 20000) * 8);
    run<16, false>(buf, T);
    run<8, false>(buf, T);
    run<8, true>(buf, T);
    run<4, true>(buf, T);
    return 0;
}
