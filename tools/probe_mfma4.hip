// Which lanes feed which lanes in v_mfma_f64_4x4x4 (four independent 4x4x4 products)?
// out[0][s][l] = D(lane l) with A = [lane == s], B = 1;  out[1][s][l] with A = 1, B = [lane == s].
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out) {
    const int l = threadIdx.x;
    for (int s = 0; s < 64; ++s) {
        out[(0 * 64 + s) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == s ? 1.0 : 0.0, 1.0, 0.0, 0, 0, 0);
        out[(1 * 64 + s) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, l == s ? 1.0 : 0.0, 0.0, 0, 0, 0);
    }
}
int main() {
    double *d, h[2 * 64 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("%s operand: source lane -> lanes of D it reaches\n", w ? "B" : "A");
        for (int s = 0; s < 64; ++s) {
            printf("  %2d ->", s);
            for (int l = 0; l < 64; ++l)
                if (h[(w * 64 + s) * 64 + l] != 0.0) printf(" %d", l);
            printf("\n");
        }
    }
    return 0;
}
