// Stereo utilities: stereo_widener, stereo_panner, stereo_bus - forward and backward, fused elementwise kernels with in-kernel
// control-gradient reduction. Replaces dasp_pytorch/functional.py:580-605 (widener), :608-636 (panner), :32-62 (bus) and their
// autograd graphs. All HBM-bound streaming kernels: one pass over the inputs, one over the outputs, nothing else moves.
//
// A workgroup handles one segment of ST_SEG samples of one "line" (a batch item for the widener, a (batch, track) row for the
// panner, a (batch, channel) row for the bus); backward kernels write one partial sum per (line, segment[, track]) and a finalize
// kernel reduces them in fp64 and applies the chain rule of the control.
#include "common.hpp"

namespace dasp {

constexpr int ST_THREADS = 256;
constexpr int ST_SEG = 4096;     // samples per workgroup: 4 float4 per thread

// STREAM: non-temporal hint, used by the backward kernels (several read streams + a write stream; elementwise.hip: 0.160 -> 0.132 ms)
template <bool STREAM = false> __device__ __forceinline__ f4 ld4(const float* p, long i, long n, bool vec) {
    if (vec) return STREAM ? ld_stream(reinterpret_cast<const f4*>(p + i)) : *reinterpret_cast<const f4*>(p + i);
    return f4{i < n ? p[i] : 0.f, i + 1 < n ? p[i + 1] : 0.f, i + 2 < n ? p[i + 2] : 0.f, i + 3 < n ? p[i + 3] : 0.f};
}
template <bool STREAM = false> __device__ __forceinline__ void st4(float* p, long i, long n, bool vec, f4 v) {
    if (vec) { if (STREAM) st_stream(reinterpret_cast<f4*>(p + i), v); else *reinterpret_cast<f4*>(p + i) = v; return; }
    if (i < n) p[i] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
}
__device__ __forceinline__ float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// workgroup sum of `v`, written by thread 0 to *out
__device__ __forceinline__ void block_sum_to(float v, float* out) {
    __shared__ float red[ST_THREADS / 64];
    const float w = wave_sum(v);
    __syncthreads();                     // red may still be read from a previous call
    if (lane_id() == 0) red[wave_id()] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ST_THREADS / 64; ++i) s += red[i];
        *out = s;
    }
}

// ---- widener: left = L + k R, right = k L + R, k = 1 - 2 width  (mid/side scaling of functional.py:592-603 multiplied out) -------
// x, y (B, 2, N); width (B). Backward (x = gy here, out = gx): the same map (it is symmetric); d/dwidth = -2 sum (R gL + L gR).
template <bool BWD>
__global__ void __launch_bounds__(ST_THREADS)
widener_kernel(const float* __restrict__ x, const float* __restrict__ width, const float* __restrict__ gy, float* __restrict__ out,
               float* __restrict__ partials, long N, int nseg, int vec) {
    const int b = blockIdx.x / nseg, seg = blockIdx.x % nseg;
    const float k = 1.f - 2.f * width[b];
    const float* xl = x + (size_t)b * 2 * N;
    const float* xr = xl + N;
    const float* src_l = BWD ? gy + (size_t)b * 2 * N : xl;
    const float* src_r = src_l + N;
    float* ol = out + (size_t)b * 2 * N;
    float* orr = ol + N;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < ST_SEG / (4 * ST_THREADS); ++j) {
        const long i = (long)seg * ST_SEG + (long)(j * ST_THREADS + threadIdx.x) * 4;
        if (i >= N) break;
        const f4 l = ld4<BWD>(src_l, i, N, vec), r = ld4<BWD>(src_r, i, N, vec);
        st4<BWD>(ol, i, N, vec, l + k * r);
        st4<BWD>(orr, i, N, vec, k * l + r);
        if (BWD) acc += dot4(ld4<true>(xr, i, N, vec), l) + dot4(ld4<true>(xl, i, N, vec), r);     // R gL + L gR
    }
    if (BWD) block_sum_to(-2.f * acc, partials + (size_t)b * nseg + seg);
}

// ---- panner: y[b, 0, t, :] = lg x[b, t, :], y[b, 1, t, :] = rg x[b, t, :]  (functional.py:621-634) -------------------------------
// theta = pan pi / 2, lg = sqrt((pi/2 - theta) (2/pi) cos theta), rg = sqrt(theta (2/pi) sin theta). x (B, T, N), pan (B * T), y (B, 2, T, N).
__device__ __forceinline__ void pan_gains(float pan, float& lg, float& rg) {
    const float hp = 1.57079632679489661923f, theta = pan * hp;
    lg = sqrtf((hp - theta) * (2.f / 3.14159265358979323846f) * cosf(theta));
    rg = sqrtf(theta * (2.f / 3.14159265358979323846f) * sinf(theta));
}
template <bool BWD>
__global__ void __launch_bounds__(ST_THREADS)
panner_kernel(const float* __restrict__ x, const float* __restrict__ pan, const float* __restrict__ gy, float* __restrict__ out,
              float* __restrict__ partials, int T, long N, int nseg, int vec) {
    const int row = blockIdx.x / nseg, seg = blockIdx.x % nseg;      // row = b * T + t
    const int b = row / T, t = row % T;
    float lg, rg;
    pan_gains(pan[row], lg, rg);
    const float* xr = x + (size_t)row * N;
    const size_t y0 = ((size_t)(b * 2) * T + t) * N, y1 = ((size_t)(b * 2 + 1) * T + t) * N;
    float al = 0.f, ar = 0.f;
#pragma unroll
    for (int j = 0; j < ST_SEG / (4 * ST_THREADS); ++j) {
        const long i = (long)seg * ST_SEG + (long)(j * ST_THREADS + threadIdx.x) * 4;
        if (i >= N) break;
        const f4 xv = ld4(xr, i, N, vec);
        if (!BWD) {
            st4(out + y0, i, N, vec, lg * xv);
            st4(out + y1, i, N, vec, rg * xv);
        } else {
            const f4 g0 = ld4<true>(gy + y0, i, N, vec), g1 = ld4<true>(gy + y1, i, N, vec);
            st4<true>(out + (size_t)row * N, i, N, vec, lg * g0 + rg * g1);
            al += dot4(xv, g0); ar += dot4(xv, g1);
        }
    }
    if (BWD) {
        block_sum_to(al, partials + ((size_t)row * nseg + seg) * 2);
        block_sum_to(ar, partials + ((size_t)row * nseg + seg) * 2 + 1);
    }
}

// ---- bus: y[b, c, :] = sum_t 10^(send_db[b, t] / 20) x[b, c, t, :]  (functional.py:49-59) ------------------------------------------
// x (B, 2, T, N), send_db (B * T), y (B, 2, N). Backward: gx[b, c, t, :] = s[b, t] gy[b, c, :]; partial[(b, c), seg, t] = sum gy x.
constexpr int ST_TMAX = 64;
template <bool BWD>
__global__ void __launch_bounds__(ST_THREADS)
bus_kernel(const float* __restrict__ x, const float* __restrict__ send_db, const float* __restrict__ gy, float* __restrict__ out,
           float* __restrict__ partials, int T, long N, int nseg, int vec) {
    __shared__ float s_lin[ST_TMAX];
    const int row = blockIdx.x / nseg, seg = blockIdx.x % nseg;      // row = b * 2 + c
    const int b = row >> 1;
    for (int t = threadIdx.x; t < T; t += ST_THREADS) s_lin[t] = exp10f(send_db[(size_t)b * T + t] * 0.05f);
    __syncthreads();
    const float* xr = x + (size_t)row * T * N;
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < ST_SEG / (4 * ST_THREADS); ++j) {
            const long i = (long)seg * ST_SEG + (long)(j * ST_THREADS + threadIdx.x) * 4;
            if (i >= N) break;
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t) acc += s_lin[t] * ld4(xr + (size_t)t * N, i, N, vec);
            st4(out + (size_t)row * N, i, N, vec, acc);
        }
    } else {
        f4 g[ST_SEG / (4 * ST_THREADS)];
#pragma unroll
        for (int j = 0; j < ST_SEG / (4 * ST_THREADS); ++j) {
            const long i = (long)seg * ST_SEG + (long)(j * ST_THREADS + threadIdx.x) * 4;
            g[j] = i < N ? ld4(gy + (size_t)row * N, i, N, vec) : f4{0.f, 0.f, 0.f, 0.f};
        }
        for (int t = 0; t < T; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < ST_SEG / (4 * ST_THREADS); ++j) {
                const long i = (long)seg * ST_SEG + (long)(j * ST_THREADS + threadIdx.x) * 4;
                if (i >= N) break;
                acc += dot4(ld4<true>(xr + (size_t)t * N, i, N, vec), g[j]);
                st4<true>(out + ((size_t)row * T + t) * N, i, N, vec, s_lin[t] * g[j]);
            }
            block_sum_to(acc, partials + ((size_t)row * nseg + seg) * T + t);
        }
    }
}

// mode 0 widener: gctl[b] = sum_seg partials[b][seg]
// mode 1 panner:  gctl[row] = d lg/d pan * sum partial_l + d rg/d pan * sum partial_r   (fp64; zero where the gain is zero: pan = 0 or 1)
// mode 2 bus:     gctl[b * T + t] = ln10/20 s[b,t] * sum_{c, seg} partials[(b, c), seg, t]
__global__ void stereo_finalize_kernel(const float* __restrict__ partials, const float* __restrict__ ctl, float* __restrict__ gctl, int n,
                                       int nseg, int T, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) {
        double s = 0.0;
        for (int k = 0; k < nseg; ++k) s += (double)partials[(size_t)i * nseg + k];
        gctl[i] = (float)s;
    } else if (mode == 1) {
        double sl = 0.0, sr = 0.0;
        for (int k = 0; k < nseg; ++k) { sl += (double)partials[((size_t)i * nseg + k) * 2]; sr += (double)partials[((size_t)i * nseg + k) * 2 + 1]; }
        const double hp = 1.57079632679489661923, th = (double)ctl[i] * hp, c = cos(th), sn = sin(th);
        const double ul = (hp - th) * (2.0 / 3.14159265358979323846) * c, ur = th * (2.0 / 3.14159265358979323846) * sn;   // lg^2, rg^2
        const double dul = (2.0 / 3.14159265358979323846) * (-c - (hp - th) * sn), dur = (2.0 / 3.14159265358979323846) * (sn + th * c);   // d/dtheta
        const double dl = ul > 0.0 ? 0.5 * dul / sqrt(ul) * hp : 0.0, dr = ur > 0.0 ? 0.5 * dur / sqrt(ur) * hp : 0.0;
        gctl[i] = (float)(dl * sl + dr * sr);
    } else {
        const int b = i / T, t = i % T;
        double s = 0.0;
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < nseg; ++k) s += (double)partials[(((size_t)(b * 2 + c)) * nseg + k) * T + t];
        gctl[i] = (float)(s * 0.11512925464970228 * pow(10.0, (double)ctl[i] * 0.05));
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
inline int st_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
inline bool st_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int st_nseg(long N) { return (int)((N + ST_SEG - 1) / ST_SEG); }
}  // namespace

extern "C" {

long dasp_stereo_partial_floats(int op, long B, int T, long N) {      // op 0 widener, 1 panner, 2 bus
    const long ns = st_nseg(N);
    return op == 0 ? B * ns : op == 1 ? B * T * ns * 2 : B * 2 * ns * T;
}

int dasp_widener_forward(const float* x, const float* width, float* y, int B, long N, void* stream) {
    if (!x || !width || !y || B <= 0 || N <= 0) return DASP_ERR_ARG;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(y);
    if ((long)B * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(widener_kernel<false>, dim3((unsigned)((long)B * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, width,
                       (const float*)nullptr, y, (float*)nullptr, N, nseg, vec);
    return st_check();
}
int dasp_widener_backward(const float* x, const float* width, const float* gy, float* gx, float* gwidth, float* partials, int B, long N,
                          void* stream) {
    if (!x || !width || !gy || !gx || !gwidth || !partials || B <= 0 || N <= 0) return DASP_ERR_ARG;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(gy) && st_al16(gx);
    if ((long)B * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(widener_kernel<true>, dim3((unsigned)((long)B * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, width, gy, gx,
                       partials, N, nseg, vec);
    hipLaunchKernelGGL(stereo_finalize_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const float*)partials, width, gwidth, B,
                       nseg, 1, 0);
    return st_check();
}

int dasp_panner_forward(const float* x, const float* pan, float* y, int B, int T, long N, void* stream) {
    if (!x || !pan || !y || B <= 0 || T <= 0 || N <= 0) return DASP_ERR_ARG;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(y);
    if ((long)B * T * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(panner_kernel<false>, dim3((unsigned)((long)B * T * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, pan,
                       (const float*)nullptr, y, (float*)nullptr, T, N, nseg, vec);
    return st_check();
}
int dasp_panner_backward(const float* x, const float* pan, const float* gy, float* gx, float* gpan, float* partials, int B, int T, long N,
                         void* stream) {
    if (!x || !pan || !gy || !gx || !gpan || !partials || B <= 0 || T <= 0 || N <= 0) return DASP_ERR_ARG;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(gy) && st_al16(gx);
    if ((long)B * T * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(panner_kernel<true>, dim3((unsigned)((long)B * T * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, pan, gy, gx,
                       partials, T, N, nseg, vec);
    hipLaunchKernelGGL(stereo_finalize_kernel, dim3((B * T + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const float*)partials, pan, gpan,
                       B * T, nseg, T, 1);
    return st_check();
}

int dasp_bus_forward(const float* x, const float* send_db, float* y, int B, int T, long N, void* stream) {
    if (!x || !send_db || !y || B <= 0 || T <= 0 || N <= 0) return DASP_ERR_ARG;
    if (T > ST_TMAX) return DASP_ERR_UNSUPPORTED;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(y);
    if ((long)B * 2 * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bus_kernel<false>, dim3((unsigned)((long)B * 2 * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, send_db,
                       (const float*)nullptr, y, (float*)nullptr, T, N, nseg, vec);
    return st_check();
}
int dasp_bus_backward(const float* x, const float* send_db, const float* gy, float* gx, float* gsend, float* partials, int B, int T, long N,
                      void* stream) {
    if (!x || !send_db || !gy || !gx || !gsend || !partials || B <= 0 || T <= 0 || N <= 0) return DASP_ERR_ARG;
    if (T > ST_TMAX) return DASP_ERR_UNSUPPORTED;
    const int nseg = st_nseg(N), vec = (N % 4 == 0) && st_al16(x) && st_al16(gy) && st_al16(gx);
    if ((long)B * 2 * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bus_kernel<true>, dim3((unsigned)((long)B * 2 * nseg)), dim3(ST_THREADS), 0, (hipStream_t)stream, x, send_db, gy, gx,
                       partials, T, N, nseg, vec);
    hipLaunchKernelGGL(stereo_finalize_kernel, dim3((B * T + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const float*)partials, send_db,
                       gsend, B * T, nseg, T, 2);
    return st_check();
}

}  // extern "C"
