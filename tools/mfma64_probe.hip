// Developer probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (one wave; prints which candidate layout matches A * B).
// hipcc --offload-arch=gfx950 -O2 -o tools/mfma64_probe tools/mfma64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {   // A[16][4], B[4][16] row-major
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * 4 + kk], B[kk * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + (i * 37 % 101) * 0.25; hB[i] = 2 + (i * 53 % 97) * 0.5; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int q = 0; q < 4; ++q) s += hA[i * 4 + q] * hB[q * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = hD[l * 4 + r];
        if (v != ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) ok1 = 0;      // rows 4 (l / 16) + r   (the f32 16x16x4 layout)
        if (v != ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) ok2 = 0;      // rows l / 16 + 4 r
        if (v != ref[(l & 15) * 16 + 4 * (l >> 4) + r]) ok3 = 0;        // transposed
    }
    printf("mfma_f64_16x16x4: A[i][k] lane 16k+i, B[k][j] lane 16k+j; D rows 4(l/16)+r: %d, rows l/16+4r: %d, transposed: %d\n", ok1, ok2, ok3);
    return 0;
}
