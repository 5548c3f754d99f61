// Issue cost, for one wave alone on its SIMD, of LDS instructions that nothing waits for: 32 in a row, distinct addresses, one
// s_waitcnt at the end of each group; ds_write_b32 / b64 / b128, ds_read_b32 / b64 / b128, all 64 lanes and 56 lanes; v_readlane.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench7 tools/ubench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define IT 1024
template <int MODE> __global__ void __launch_bounds__(64) k(unsigned* out, unsigned long long* clk, unsigned c, int lanes) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    unsigned a = threadIdx.x + c;
    u4 v = {a, a + 1, a + 2, a + 3};
    u2 v2 = {a, a + 1};
    unsigned addr = 16 * threadIdx.x;
    unsigned s = 0;
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
    __syncthreads();
    if ((int)threadIdx.x >= lanes) { out[blockIdx.x * 64 + threadIdx.x] = 0; return; }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (MODE == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(a), "n"(u * 1024) : "memory");
            if (MODE == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v2), "n"(u * 1024) : "memory");
            if (MODE == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(u * 1024) : "memory");
            if (MODE == 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(u * 1024) : "memory");
            if (MODE == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v2) : "v"(addr), "n"(u * 1024) : "memory");
            if (MODE == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 1024) : "memory");
            if (MODE == 6) { unsigned r; asm volatile("v_readlane_b32 %0, %1, 55" : "=s"(r) : "v"(a)); s ^= r; }
            if (MODE == 7) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(addr), "v"(v2), "v"(v2), "n"((u * 1024 / 8) & 255), "n"((u * 1024 / 8 + 1) & 255) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a ^ v.x ^ v.y ^ v.z ^ v.w ^ v2.x ^ v2.y ^ s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <typename K> void run(const char* name, K kern, unsigned* d, unsigned long long* clk, int lanes) {
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(80), dim3(64), 49152, 0, d, clk, 7u, lanes); (void)hipDeviceSynchronize(); }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-16s lanes %2d  %6.2f cycles per instruction (%6.2f ns)\n", name, lanes, h[0] / (IT * 32.0), h[1] * 10.0 / (IT * 32.0));
}
int main() { unsigned* d; unsigned long long* clk; (void)hipMalloc(&d, 80 * 64 * 4); (void)hipMalloc(&clk, 16);
    for (int lanes : {64, 56, 32}) {
        run("ds_write_b32", k<0>, d, clk, lanes); run("ds_write_b64", k<1>, d, clk, lanes); run("ds_write_b128", k<2>, d, clk, lanes); run("ds_write2_b64", k<7>, d, clk, lanes);
        run("ds_read_b32", k<3>, d, clk, lanes); run("ds_read_b64", k<4>, d, clk, lanes); run("ds_read_b128", k<5>, d, clk, lanes); run("v_readlane_b32", k<6>, d, clk, lanes);
    }
    return 0; }
