// Do stores (loads) that skip some lanes of a row cost less HBM time?  1024 waves, each streams ROWS rows of
// 64 lanes x PIECE bytes (PIECE = 32: the interleaved gain rows; 8: plain rows); a fraction of the lanes
// (chosen per wave by a hash, fixed over the rows) is sent to a small per-lane slot instead.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_partial.hip -o /tmp/ubench_partial
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int PIECE, bool STORE> __global__ __launch_bounds__(64) void k(char *buf, char *slots, int rows, unsigned thresh) {
    const int lane = threadIdx.x, wave = blockIdx.x;
    unsigned h = (wave * 64 + lane) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const bool skip = (h & 1023u) < thresh;
    char *base = buf + (size_t)wave * rows * 64 * PIECE + lane * PIECE;
    char *slot = slots + ((size_t)wave * 64 + lane) * PIECE;
    double acc = 0;
    for (int r = 0; r < rows; ++r) {
        char *p = skip ? slot : base + (size_t)r * 64 * PIECE;
        if constexpr (STORE) {
            if constexpr (PIECE == 32) { ((double4 *)p)[0] = double4{1.0 * r, 2.0, 3.0, 4.0}; }
            else *(double *)p = 1.0 * r;
        } else {
            if constexpr (PIECE == 32) {
                typedef double v4d __attribute__((ext_vector_type(4)));
                v4d v = __builtin_nontemporal_load((const v4d *)p);
                acc += v[0] + v[3];
            } else acc += __builtin_nontemporal_load((const double *)p);
        }
    }
    if (!STORE && acc == 12345.678) *(double *)slot = acc;
}
template <int PIECE, bool STORE> void run(char *buf, char *slots, size_t bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rows = (int)(bytes / (1024ull * 64 * PIECE));
    for (unsigned f : {0u, 154u, 307u, 512u, 1024u}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<PIECE, STORE>), dim3(1024), dim3(64), 0, 0, buf, slots, rows, f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%s piece %2d B, %4.0f %% of lanes skipped: %.3f ms (%.2f TB/s of the full stream)\n", STORE ? "store" : "load ", PIECE, f / 10.24, best,
               bytes / (best * 1e-3) / 1e12);
    }
}
int main() {
    const size_t bytes = 2ull << 30;
    char *buf, *slots;
    hipMalloc(&buf, bytes);
    hipMalloc(&slots, 1024 * 64 * 32);
    hipMemset(buf, 0, bytes);
    run<32, true>(buf, slots, bytes);
    run<8, true>(buf, slots, bytes);
    run<32, false>(buf, slots, bytes);
    run<8, false>(buf, slots, bytes);
    return 0;
}
