// Micro-benchmarks of the VALU / cross-lane primitives the scan kernels are built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define ITER 4096
__global__ void k_fma(float* out, float c) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(c));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fmac_vv(float* out, float c) {
    float a[8]; float b = c + threadIdx.x;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma(float* out, float c) {
    f2 a[8]; f2 b = {c, c + 1};
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
    }
    f2 s = {0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_pkfma_s(float* out, float c) {   // SGPR-pair coefficient, op_sel broadcast of low half
    f2 a[8]; f2 c2 = {c, c * 2};
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(c2));
    }
    f2 s = {0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_dpp(float* out, float c) {      // v_mov_dpp row_shr:1 + fma (one scan level per stream)
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xf, 0xf, true));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(t));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fmac_dpp(float* out, float c) {  // fused: v_fmac_f32_dpp
    float a[8]; float cc = c;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(cc));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_bperm(float* out, float c) {     // ds_bpermute dependent chain (latency)
    float a = threadIdx.x * 0.001f;
    int idx = ((threadIdx.x + 1) & 63) * 4;
    for (int it = 0; it < ITER; ++it) {
        a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, a))) + c;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_dppchain(float* out, float c) {  // dependent chain: dpp mov + fma (latency per scan level)
    float a = threadIdx.x * 0.001f;
    for (int it = 0; it < ITER; ++it) {
        float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xf, 0xf, true));
        a = fmaf(t, c, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_fmachain(float* out, float c) {  // dependent fma chain (latency)
    float a = threadIdx.x * 0.001f;
    for (int it = 0; it < ITER * 8; ++it) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "s"(c));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_readlane(float* out, float c) {  // v_readlane -> SGPR -> VALU use
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[(i + 3) & 7]), 15));
            a[i] = fmaf(s, c, a[i]);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd, double ops_per_thread, float* d) {
    const int blocks = 256 * waves_per_simd, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double total = ops_per_thread * blocks * threads;
    // cycles per wave-instruction per SIMD at 2.4 GHz
    const double inst_per_simd = ops_per_thread * waves_per_simd;  // each SIMD runs waves_per_simd waves
    printf("%-12s w/SIMD=%d  %8.3f ms  %7.2f Tops/s  %6.2f cyc/inst/SIMD(@2.4GHz)\n", name, waves_per_simd, ms, total / ms / 1e9,
           ms * 1e-3 * 2.4e9 / inst_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        run("fma_s", k_fma, w, 8.0 * ITER, d);
        run("fmac_vv", k_fmac_vv, w, 8.0 * ITER, d);
        run("pk_fma_v", k_pkfma, w, 8.0 * ITER, d);
        run("pk_fma_s", k_pkfma_s, w, 8.0 * ITER, d);
        run("dpp+fmac", k_dpp, w, 16.0 * ITER, d);
        run("fmac_dpp", k_fmac_dpp, w, 8.0 * ITER, d);
        run("readlane+fma", k_readlane, w, 16.0 * ITER, d);
    }
    run("bperm_chain", k_bperm, 1, 1.0 * ITER, d);
    run("dpp_chain", k_dppchain, 1, 2.0 * ITER, d);
    run("fma_chain", k_fmachain, 1, 8.0 * ITER, d);
    return 0;
}
