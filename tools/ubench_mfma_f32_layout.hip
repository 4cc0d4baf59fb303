// Prints which element of D = A B each lane and register of v_mfma_f32_16x16x4 holds (D[i][j] = 100 i + j).
// Build: hipcc -O2 --offload-arch=gfx950 ubench_mfma_f32_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D) {
    int l = threadIdx.x;
    float a = A[(l % 16) * 4 + l / 16];
    float b = B[(l / 16) * 16 + l % 16];
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    float hA[64] = {0}, hB[64] = {0}, hD[256];
    for (int i = 0; i < 16; ++i) { hA[i * 4 + 0] = i; hA[i * 4 + 1] = 1; }
    for (int j = 0; j < 16; ++j) { hB[0 * 16 + j] = 100; hB[1 * 16 + j] = j; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 7) printf("lane %2d: %5g %5g %5g %5g\n", l, hD[l*4], hD[l*4+1], hD[l*4+2], hD[l*4+3]);
    return 0;
}
