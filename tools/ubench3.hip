// Is VALU issue limited by instruction fetch for long straight-line code? Same instruction mix, body size varied.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNROLL, int VOP3>
__global__ void k_body(float* out, float c, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = c * i + threadIdx.x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (VOP3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int w, int unroll, float* d) {
    const int blocks = 256 * w, threads = 256;
    const int total_per_wave = 1 << 18;                       // VALU instructions per wave
    const int iters = total_per_wave / (8 * unroll);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0e-9f, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0e-9f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s body %6d B  w/SIMD=%d  %7.3f ms  %6.2f ns/inst/SIMD\n", name, unroll * 8 * (name[0] == 'f' && name[1] == 'm' && name[2] == 'a' ? 8 : 4), w, ms,
           ms * 1e6 / ((double)total_per_wave * w));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 3}) {
        run("fmac u1", k_body<1, 0>, w, 1, d); run("fmac u64", k_body<64, 0>, w, 64, d); run("fmac u512", k_body<512, 0>, w, 512, d); run("fmac u2048", k_body<2048, 0>, w, 2048, d);
        run("fma3 u1", k_body<1, 1>, w, 1, d); run("fma3 u64", k_body<64, 1>, w, 64, d); run("fma3 u512", k_body<512, 1>, w, 512, d); run("fma3 u2048", k_body<2048, 1>, w, 2048, d);
    }
    return 0;
}
