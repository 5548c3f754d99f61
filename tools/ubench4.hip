// Dependent-chain latencies (1 wave per SIMD) of the instruction kinds used by the lane scan.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N 32768
#define CHAIN(NAME, DECL, STMT) __global__ void NAME(float* out, float c) { float a = threadIdx.x * 0.001f, b = c; f2 p = {a, b}, q = {c, c}; f2 sq = {c, 2 * c}; DECL; \
    for (int it = 0; it < N / 8; ++it) { _Pragma("unroll") for (int u = 0; u < 8; ++u) { STMT; } } out[blockIdx.x * blockDim.x + threadIdx.x] = a + p.x + p.y; }
CHAIN(c_fma, , asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b)))
CHAIN(c_fma_s, , asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "s"(c)))
CHAIN(c_pk_v, , asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(q)))
CHAIN(c_pk_s, , asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "s"(sq)))
CHAIN(c_pk_opsel, , asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[1,0,1]" : "+v"(p) : "s"(sq)))
CHAIN(c_dpp_fma, float t, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(a)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(t), "v"(b)))
CHAIN(c_bcast_fma, float t, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(a)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(t), "v"(b)))
CHAIN(c_waveshr_fma, float t, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(t), "v"(b)))
CHAIN(c_readlane_fma, float s, asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(a)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "s"(s), "v"(b)))
template <typename K> void run(const char* name, K kern, int per, float* d, int w = 1) {
    hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), 0, 0, d, 1e-9f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), 0, 0, d, 1e-9f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s w=%d %7.3f ms  %6.2f ns per chain step (%d instr)\n", name, w, ms, ms * 1e6 / N, per);
}
int main() { float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 3}) { run("fma", c_fma, 1, d, w); run("fma_sgpr", c_fma_s, 1, d, w); run("pk_fma_v", c_pk_v, 1, d, w); run("pk_fma_s", c_pk_s, 1, d, w); run("pk_fma_opsel", c_pk_opsel, 1, d, w);
    run("dpp+fma", c_dpp_fma, 2, d, w); run("bcast15+fma", c_bcast_fma, 2, d, w); run("wave_shr+fma", c_waveshr_fma, 2, d, w); run("readlane+fma", c_readlane_fma, 2, d, w); }
    return 0; }
