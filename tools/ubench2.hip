// Second round: which operand kinds / encodings issue at full rate on gfx950? Reports in-kernel
// shader cycles (s_memtime) per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 2048
#define BODY8(STMT) _Pragma("unroll") for (int i = 0; i < 8; ++i) { STMT; }
#define KERNEL(NAME, DECL, STMT)                                                       \
    __global__ void NAME(float* out, long long* cyc, float c) {                        \
        float a[8], b[8]; f2 p[8]; f2 q = {c, c + 1}; f2 c2 = {c, 2 * c};              \
        for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = c * i + threadIdx.x; p[i] = f2{a[i], b[i]}; } \
        DECL;                                                                          \
        long long t0 = clock64();                                                      \
        for (int it = 0; it < ITER; ++it) { BODY8(STMT) }                              \
        long long t1 = clock64();                                                      \
        float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i] + p[i].x + p[i].y;   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + q.x + c2.x;                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                       \
    }
KERNEL(k_fmac_vv, , asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])))
KERNEL(k_fmac_sv, , asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i])))
KERNEL(k_mul_sv, , asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "s"(c), "v"(b[i])))
KERNEL(k_fma_vvv, , asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])))
KERNEL(k_fma_svv, , asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(c), "v"(b[i])))
KERNEL(k_pkfma_vvv, , asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q), "v"(p[(i + 1) & 7])))
KERNEL(k_pkfma_svv, , asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(c2), "v"(p[(i + 1) & 7])))
KERNEL(k_pkmul_vv, , asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(q), "v"(p[(i + 1) & 7])))
KERNEL(k_movdpp, , asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(a[i]) : "v"(b[i])))
KERNEL(k_fmacdpp, , asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])))
KERNEL(k_readlane, float sr, asm volatile("v_readlane_b32 %0, %1, 15" : "=s"(sr) : "v"(b[i])); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sr), "v"(b[i])))
KERNEL(k_exp, , asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i])))
KERNEL(k_log, , asm volatile("v_log_f32 %0, %1" : "=v"(a[i]) : "v"(b[i])))
KERNEL(k_rcp, , asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i])))
KERNEL(k_cndmask, , asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])))
KERNEL(k_swap32, , asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i])))

template <typename K>
void run(const char* name, K kern, int w, int insts_per_stmt, float* d, long long* dc) {
    const int blocks = 256 * w, threads = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, dc, 1.0e-9f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, dc, 1.0e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double inst_wave = 8.0 * ITER * insts_per_stmt;
    printf("%-12s w/SIMD=%d  %7.3f ms  wave0 %9lld cyc  -> %6.2f cyc/inst/SIMD (clock64), %6.2f (wall@2.4GHz)\n", name, w, ms, cyc,
           (double)cyc / (inst_wave * w), ms * 1e-3 * 2.4e9 / (inst_wave * w));
}
#define RUN(K, N) run(#K, K, w, N, d, dc)
int main() {
    float* d; long long* dc; hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); hipMalloc(&dc, 8);
    for (int w : {1, 2, 4}) {
        RUN(k_fmac_vv, 1); RUN(k_fmac_sv, 1); RUN(k_mul_sv, 1); RUN(k_fma_vvv, 1); RUN(k_fma_svv, 1); RUN(k_pkfma_vvv, 1); RUN(k_pkfma_svv, 1);
        RUN(k_pkmul_vv, 1); RUN(k_movdpp, 1); RUN(k_fmacdpp, 1); RUN(k_readlane, 2); RUN(k_exp, 1); RUN(k_log, 1); RUN(k_rcp, 1); RUN(k_cndmask, 1); RUN(k_swap32, 1);
    }
    return 0;
}
