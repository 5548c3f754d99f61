// Can one wave overlap v_mfma_f32_16x16x4_f32 with dependent VALU chains? Per iteration: 64 dependent v_fma_f32 and 8 independent-accumulator
// MFMAs, (a) VALU only, (b) MFMA only, (c) both, MFMAs grouped in front, (d) both, one MFMA per 8 VALU instructions. 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench5 tools/ubench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define N 4096
template <int MODE> __global__ void __launch_bounds__(512) k(float* out, float c) {
    float a = threadIdx.x * 0.001f, b = c;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float ma = a, mb = c;
    for (int it = 0; it < N; ++it) {
        if (MODE == 0 || MODE == 2) {
            if (MODE == 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[m & 3], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[m & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(ma), "v"(mb));
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <typename K> void run(const char* name, K kern, float* d, int threads) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1e-9f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1e-9f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD %d  %7.3f ms  %7.1f ns per iteration (64 fma + 8 mfma)\n", name, threads / 256, ms, ms * 1e6 / N);
}
int main() { float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int th : {256, 512}) {
        run("valu chain only", k<0>, d, th); run("mfma only", k<1>, d, th); run("mfma block, then valu chain", k<2>, d, th); run("1 mfma per 8 valu", k<3>, d, th);
    }
    return 0; }
