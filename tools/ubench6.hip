// What one wave alone on a SIMD can issue, and what the clock is while it does: per kernel, s_memtime (shader clock) and wall_clock64
// (100 MHz) around a loop of 1024 x 32 instructions: dependent v_xor, independent v_xor, wave-shift DPP moves, v_bfi, and an LDS
// round trip (ds_write_b128 -> ds_read_b128 of the same wave). 80 workgroups of 64 (as the random stream's small case) and 1024 of 64.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench6 tools/ubench6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define IT 1024
template <int MODE> __global__ void __launch_bounds__(64) k(unsigned* out, unsigned long long* clk, unsigned c) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    unsigned a = threadIdx.x, b = c, d = c + 1, e = c + 2, f = c + 3;
    u4 v = {a, b, d, e};
    const unsigned addr = 16 * threadIdx.x;
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < IT; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(b));
                                          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f) : "v"(b)); }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(d));
                                          asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(e));
                                          asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(e) : "v"(f));
                                          asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(a)); }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "s"(c), "v"(b)); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(d) : "s"(c), "v"(b));
                                          asm volatile("v_bfe_i32 %0, %0, 0, 1" : "+v"(e)); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(f) : "v"(b), "s"(c)); }
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { asm volatile("ds_write_b128 %1, %0\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(addr) : "memory"); }
        } else if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(addr) : "memory"); }
        } else if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr) : "memory"); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a ^ d ^ e ^ f ^ v.x ^ v.y ^ v.z ^ v.w;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <typename K> void run(const char* name, K kern, unsigned* d, unsigned long long* clk, int wgs) {
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), 0, 0, d, clk, 7u); hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), 0, 0, d, clk, 7u); hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ns = h[1] * 10.0, per = ns / (IT * 32.0);
    printf("%-40s wgs %5d  s_memtime %9llu  wall %8.0f ns  -> %6.2f ns per instruction, memtime ticks per instruction %6.2f, memtime MHz %7.1f\n", name, wgs, h[0], ns, per,
           h[0] / (IT * 32.0), h[0] / ns * 1e3);
}
int main() { unsigned* d; unsigned long long* clk; hipMalloc(&d, 1024 * 64 * 4); hipMalloc(&clk, 16);
    for (int wgs : {80, 1024}) {
        run("dependent v_xor", k<0>, d, clk, wgs); run("4 independent v_xor chains", k<1>, d, clk, wgs); run("wave_shr dpp moves", k<2>, d, clk, wgs);
        run("bfi/bfe/bitop3 mix", k<3>, d, clk, wgs); run("ds_write_b128 -> ds_read_b128 -> wait", k<4>, d, clk, wgs); run("ds_read_b128 -> wait", k<5>, d, clk, wgs);
        run("ds_read_b32 -> wait", k<6>, d, clk, wgs);
    }
    return 0; }
