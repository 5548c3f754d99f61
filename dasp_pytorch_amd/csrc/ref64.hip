// Double-precision path of the recurrences and elementwise effects (gfx950).
//
// The reference follows the dtype of its input (`.type_as(x)`, dasp_pytorch/signal.py:113,119, functional.py:211): float64 in, float64
// arithmetic. The fp32 kernels of this library (sosfilt.hip, dynamics.hip, elementwise.hip) are built around fp32 MFMA / DPP / LDS tiles
// and have no double instantiation; float64 input used to be rounded to fp32 silently. These kernels are the genuine fp64 path instead:
// the same recurrences evaluated plainly - one thread per row (cascade) or per batch item (dynamics), one sequential sweep per section -
// which is all a float64 user needs (validation, torch.autograd.gradcheck, small reference runs: fp64 VALU is ample there) and makes
// every line checkable against the formulas it restates. They are not tuned; large float64 batches run at a small fraction of the fp32
// kernels' rate.
//
//   cascade      dasp_pytorch/signal.py:136-166 (sosfilt_via_fsm) as a direct-form II recursion; backward = the transposed recursion run
//                backwards in time + the coefficient correlations with the saved all-pole signals
//   dynamics     dasp_pytorch/functional.py:275-399 (compressor; mode 1 = the expander of dynamics.hip)
//   gain / distortion  functional.py:10-29, :65-78
#include "common.hpp"

namespace dasp {

// ---- cascade -------------------------------------------------------------------------------------------------------------------------
// c5: (Bs, S, 5) normalised coefficients b0 b1 b2 a1 a2. One thread per row (b, c); x, y (rows, N); wsave (rows, S, N) receives the
// all-pole signal w_k[n] of every section (the backward pass correlates with it) or is null.
__global__ void sos64_fwd_kernel(const double* __restrict__ c5, int bcast, const double* __restrict__ x, double* __restrict__ y,
                                 double* __restrict__ wsave, int rows, int C, long N, int S) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const double* c = c5 + (size_t)(bcast ? 0 : row / C) * S * 5;
    const double* u = x + (size_t)row * N;
    double* o = y + (size_t)row * N;
    for (int k = 0; k < S; ++k) {
        const double b0 = c[k * 5], b1 = c[k * 5 + 1], b2 = c[k * 5 + 2], a1 = c[k * 5 + 3], a2 = c[k * 5 + 4];
        double* ws = wsave ? wsave + ((size_t)row * S + k) * N : nullptr;
        double w1 = 0.0, w2 = 0.0;
        for (long n = 0; n < N; ++n) {
            const double w = u[n] - a1 * w1 - a2 * w2;
            o[n] = b0 * w + b1 * w1 + b2 * w2;
            if (ws) ws[n] = w;
            w2 = w1; w1 = w;
        }
        u = o;          // the next section filters this one's output in place
    }
}

// gx (rows, N) = adjoint cascade of gy; gc5 (Bs, S, 5) += this row's coefficient gradients (double atomics; zeroed by the caller):
//   dL/db_i = sum_n g_k[n] w_k[n-i],  dL/da_i = -sum_n o_k[n] w_k[n-i]   with g_k / o_k the adjoint input / output of section k
__global__ void sos64_bwd_kernel(const double* __restrict__ c5, int bcast, const double* __restrict__ gy, const double* __restrict__ wsave,
                                 double* __restrict__ gx, double* __restrict__ gc5, int rows, int C, long N, int S) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int item = bcast ? 0 : row / C;
    const double* c = c5 + (size_t)item * S * 5;
    const double* g = gy + (size_t)row * N;
    double* o = gx + (size_t)row * N;
    for (int k = S - 1; k >= 0; --k) {
        const double b0 = c[k * 5], b1 = c[k * 5 + 1], b2 = c[k * 5 + 2], a1 = c[k * 5 + 3], a2 = c[k * 5 + 4];
        const double* ws = wsave ? wsave + ((size_t)row * S + k) * N : nullptr;
        double z1 = 0.0, z2 = 0.0, gb0 = 0.0, gb1 = 0.0, gb2 = 0.0, ga1 = 0.0, ga2 = 0.0;
        for (long n = N - 1; n >= 0; --n) {
            const double gn = g[n];
            const double on = b0 * gn + z1;                 // transposed direct form II, time reversed
            z1 = b1 * gn - a1 * on + z2;
            z2 = b2 * gn - a2 * on;
            if (ws) {
                const double w0 = ws[n], wm1 = n >= 1 ? ws[n - 1] : 0.0, wm2 = n >= 2 ? ws[n - 2] : 0.0;
                gb0 += gn * w0; gb1 += gn * wm1; gb2 += gn * wm2;
                ga1 -= on * wm1; ga2 -= on * wm2;
            }
            o[n] = on;
        }
        if (ws && gc5) {
            double* a = gc5 + ((size_t)item * S + k) * 5;
            atomicAdd(a + 0, gb0); atomicAdd(a + 1, gb1); atomicAdd(a + 2, gb2); atomicAdd(a + 3, ga1); atomicAdd(a + 4, ga2);
        }
        g = o;
    }
}

// sos (Bs, S, 6) rows [b0 b1 b2 a0 a1 a2] -> c5 (Bs, S, 5) normalised by a0
__global__ void sos64_normalize_kernel(const double* __restrict__ sos, double* __restrict__ c5, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* s = sos + (size_t)i * 6;
    const double a0 = s[3];
    double* c = c5 + (size_t)i * 5;
    c[0] = s[0] / a0; c[1] = s[1] / a0; c[2] = s[2] / a0; c[3] = s[4] / a0; c[4] = s[5] / a0;
}
// gradient w.r.t. the normalised coefficients -> mode 0: w.r.t. sos as given (a0 from scale invariance, as finalize_section in sosfilt.hip);
// mode 1: w.r.t. (gain_db, cutoff_freq, q_factor) through jac (n, 15) of dasp_biquad_design, out (n, 3)
__global__ void sos64_grads_kernel(const double* __restrict__ c5, const double* __restrict__ sos, const double* __restrict__ gc5,
                                   const double* __restrict__ jac, int n, int mode, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* g = gc5 + (size_t)i * 5;
    if (mode == 0) {
        const double a0 = sos[(size_t)i * 6 + 3];
        const double* c = c5 + (size_t)i * 5;
        double dot = 0.0;
        for (int j = 0; j < 5; ++j) dot += g[j] * c[j];
        double* o = out + (size_t)i * 6;
        o[0] = g[0] / a0; o[1] = g[1] / a0; o[2] = g[2] / a0; o[3] = -dot / a0; o[4] = g[3] / a0; o[5] = g[4] / a0;
    } else {
        for (int d = 0; d < 3; ++d) {
            double v = 0.0;
            for (int j = 0; j < 5; ++j) v += g[j] * jac[(size_t)i * 15 + j * 3 + d];
            out[(size_t)i * 3 + d] = v;
        }
    }
}

// ---- dynamics -------------------------------------------------------------------------------------------------------------------------
struct Dyn64 { double thr, ratio, knee, makeup, eps, alpha; };
// static gain computer (functional.py:350-369; expander as in dynamics.hip): g_c and, if D, its partial derivatives
template <int MODE, bool D>
__device__ __forceinline__ double gain_computer64(double x_db, const Dyn64& it, double& d_x, double& d_t, double& d_r, double& d_w) {
    const double half = 0.5 * it.knee, lo = it.thr - half, hi = it.thr + half;
    const bool in_knee = (x_db >= lo) && (x_db <= hi) && (it.knee > 0.0);
    double g = 0.0;
    if (D) { d_x = d_t = d_r = d_w = 0.0; }
    if (MODE == 0) {
        const double ir = 1.0 / it.ratio, sl = ir - 1.0;
        if (x_db > hi) {
            g = (x_db - it.thr) * sl;
            if (D) { d_x = sl; d_t = -sl; d_r = -(x_db - it.thr) * ir * ir; }
        } else if (in_knee) {
            const double q = x_db - lo, iw = 1.0 / it.knee, h = 0.5 * q * q * iw;
            g = sl * h;
            if (D) { d_x = sl * q * iw; d_t = -d_x; d_r = -h * ir * ir; d_w = sl * (0.5 * q * iw - h * iw); }
        }
    } else {
        const double sl = 1.0 - it.ratio;
        if (x_db < lo) {
            g = -(x_db - it.thr) * sl;
            if (D) { d_x = -sl; d_t = sl; d_r = x_db - it.thr; }
        } else if (in_knee) {
            const double q = x_db - hi, iw = 1.0 / it.knee, h = 0.5 * q * q * iw;
            g = sl * h;
            if (D) { d_x = sl * q * iw; d_t = -d_x; d_r = -h; d_w = sl * (-0.5 * q * iw - h * iw); }
        }
    }
    return g;
}
__device__ __forceinline__ Dyn64 load_dyn64(const double* ctl, int b, double sample_rate, double eps) {
    const double* c = ctl + (size_t)b * 5;      // threshold_db, ratio, attack_ms, knee_db, makeup_gain_db
    Dyn64 it;
    it.thr = c[0]; it.ratio = c[1]; it.knee = c[3]; it.makeup = c[4]; it.eps = eps;
    it.alpha = exp(-2.1972245773362196 / (sample_rate * (c[2] / 1e3)));          // functional.py:339-342, ln 9
    return it;
}
constexpr double K_DB = 8.685889638065037, K_LN10_20 = 0.11512925464970228;       // 20 / ln 10, ln 10 / 20

// one thread per batch item. gsave (B, N): the smoothed gain g[n] in dB (backward). y[c][n] = x[c][n - look] lin[n] (:383-394)
template <int MODE>
__global__ void dyn64_fwd_kernel(const double* __restrict__ x, const double* __restrict__ ctl, double* __restrict__ y, double* __restrict__ gsave,
                                 int B, int C, long N, int look, double sample_rate, double eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Dyn64 it = load_dyn64(ctl, b, sample_rate, eps);
    const double* xb = x + (size_t)b * C * N;
    double* yb = y + (size_t)b * C * N;
    double g = 0.0, d0, d1, d2, d3;
    for (long n = 0; n < N; ++n) {
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += xb[(size_t)c * N + n];                  // :328
        const double x_db = K_DB * log(fmax(fabs(s), it.eps));                   // :347
        const double gc = gain_computer64<MODE, false>(x_db, it, d0, d1, d2, d3);
        g = it.alpha * g + (1.0 - it.alpha) * gc;                                // :372-380 as a recursion
        if (gsave) gsave[(size_t)b * N + n] = g;
        const double lin = exp((g + it.makeup) * K_LN10_20);                     // :388-391
        for (int c = 0; c < C; ++c) yb[(size_t)c * N + n] = (n - look >= 0 ? xb[(size_t)c * N + n - look] : 0.0) * lin;
    }
}
// gctl (B, 5): dL/d threshold_db, ratio, attack_ms, knee_db, makeup_gain_db
template <int MODE>
__global__ void dyn64_bwd_kernel(const double* __restrict__ x, const double* __restrict__ ctl, const double* __restrict__ gy,
                                 const double* __restrict__ gsave, double* __restrict__ gx, double* __restrict__ gctl, int B, int C, long N,
                                 int look, double sample_rate, double eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Dyn64 it = load_dyn64(ctl, b, sample_rate, eps);
    const double* xb = x + (size_t)b * C * N;
    const double* gb = gy + (size_t)b * C * N;
    double* gxb = gx + (size_t)b * C * N;
    const double* gs = gsave + (size_t)b * N;
    const double beta = 1.0 - it.alpha;
    double r = 0.0, acc_t = 0.0, acc_r = 0.0, acc_a = 0.0, acc_w = 0.0, acc_m = 0.0;
    // the signal path first: gx[c][m] = gy[c][m + look] lin[m + look]
    for (long m = 0; m < N; ++m) {
        const double lin = m + look < N ? exp((gs[m + look] + it.makeup) * K_LN10_20) : 0.0;
        for (int c = 0; c < C; ++c) gxb[(size_t)c * N + m] = m + look < N ? gb[(size_t)c * N + m + look] * lin : 0.0;
    }
    // the side chain, backwards in time: q = dL/d(g + makeup), r = adjoint one-pole, then the gain computer's derivatives
    for (long n = N - 1; n >= 0; --n) {
        const double lin = exp((gs[n] + it.makeup) * K_LN10_20);
        double q = 0.0, s = 0.0;
        for (int c = 0; c < C; ++c) {
            q += gb[(size_t)c * N + n] * (n - look >= 0 ? xb[(size_t)c * N + n - look] : 0.0);
            s += xb[(size_t)c * N + n];
        }
        q *= K_LN10_20 * lin;
        acc_m += q;
        r = q + it.alpha * r;                                                    // r[n] = q[n] + alpha r[n + 1]
        const double mag = fabs(s), x_db = K_DB * log(fmax(mag, it.eps));
        double d_x, d_t, d_r, d_w;
        const double gc = gain_computer64<MODE, true>(x_db, it, d_x, d_t, d_r, d_w);
        acc_a += r * ((n >= 1 ? gs[n - 1] : 0.0) - gc);                          // dL/dalpha
        const double p = beta * r;                                               // dL/dg_c[n]
        acc_t += p * d_t; acc_r += p * d_r; acc_w += p * d_w;
        const double gside = mag >= it.eps ? p * d_x * K_DB * (s < 0.0 ? -1.0 : 1.0) / mag : 0.0;
        for (int c = 0; c < C; ++c) gxb[(size_t)c * N + n] += gside;
    }
    const double nat = sample_rate * (ctl[(size_t)b * 5 + 2] / 1e3);
    double* o = gctl + (size_t)b * 5;
    o[0] = acc_t; o[1] = acc_r;
    o[2] = acc_a * it.alpha * 2.1972245773362196 / (nat * nat) * (sample_rate / 1e3);      // d alpha / d attack_ms
    o[3] = acc_w; o[4] = acc_m;
}

// ---- gain / distortion ------------------------------------------------------------------------------------------------------------------
// OP 0: y = x 10^(ctl/20), ctl per batch item (functional.py:10-29); OP 1: y = tanh(x 10^(ctl/20)), ctl per row (:65-78).
// One workgroup per row; backward: gx, and gctl[item or row] += sum gy dy/dctl (double atomics; zeroed by the caller).
template <int OP, bool BWD>
__global__ void ew64_kernel(const double* __restrict__ x, const double* __restrict__ ctl, const double* __restrict__ gy, double* __restrict__ out,
                            double* __restrict__ gctl, int C, long N) {
    __shared__ double red[256];
    const int row = blockIdx.x;
    const int ci = OP == 0 ? row / C : row;
    const double lin = exp(ctl[ci] * K_LN10_20);
    const double* xr = x + (size_t)row * N;
    double acc = 0.0;
    for (long n = threadIdx.x; n < N; n += blockDim.x) {
        const double v = xr[n] * lin;
        if (!BWD) {
            out[(size_t)row * N + n] = OP == 0 ? v : tanh(v);
        } else {
            const double t = OP == 0 ? 1.0 : 1.0 - tanh(v) * tanh(v);
            const double g = gy[(size_t)row * N + n] * t;
            out[(size_t)row * N + n] = g * lin;
            acc += g * v * K_LN10_20;
        }
    }
    if (BWD) {
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = blockDim.x / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(gctl + ci, red[0]);
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
inline int r64_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
}  // namespace

extern "C" {

/* sos (Bs, S, 6) fp64 rows [b0 b1 b2 a0 a1 a2] -> c5 (Bs, S, 5) = [b0 b1 b2 a1 a2] / a0 */
int dasp_sos64_normalize(const double* sos, int Bs, int S, double* c5, void* stream) {
    if (!sos || !c5 || Bs <= 0 || S <= 0) return DASP_ERR_ARG;
    const int n = Bs * S;
    hipLaunchKernelGGL(sos64_normalize_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, sos, c5, n);
    return r64_check();
}
/* y = cascade(x) in fp64. c5 (Bs, S, 5), Bs = B or 1; x, y (B, C, N); wsave: (B*C, S, N) doubles kept for the backward pass, or NULL. */
int dasp_sos64_forward(const double* c5, int Bs, const double* x, double* y, double* wsave, int B, int C, long N, int S, void* stream) {
    if (!c5 || !x || !y || B <= 0 || C <= 0 || N <= 0 || S <= 0 || (Bs != 1 && Bs != B)) return DASP_ERR_ARG;
    const int rows = B * C;
    hipLaunchKernelGGL(sos64_fwd_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, c5, Bs == 1 && B != 1, x, y, wsave, rows, C, N, S);
    return r64_check();
}
/* gx = adjoint cascade(gy); gc5 (Bs, S, 5) receives the gradient w.r.t. the normalised coefficients (it is zeroed here), or NULL
 * (then wsave may be NULL too). */
int dasp_sos64_backward(const double* c5, int Bs, const double* gy, const double* wsave, double* gx, double* gc5, int B, int C, long N,
                        int S, void* stream) {
    if (!c5 || !gy || !gx || B <= 0 || C <= 0 || N <= 0 || S <= 0 || (Bs != 1 && Bs != B) || (gc5 && !wsave)) return DASP_ERR_ARG;
    const int rows = B * C;
    if (gc5) {
        const hipError_t e = zero_async(gc5, sizeof(double) * (size_t)Bs * S * 5, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(sos64_bwd_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, c5, Bs == 1 && B != 1, gy, gc5 ? wsave : nullptr, gx,
                       gc5, rows, C, N, S);
    return r64_check();
}
/* mode 0: out (Bs, S, 6) = gradient w.r.t. sos (needs sos); mode 1: out (Bs, S, 3) = gradient w.r.t. (gain_db, cutoff_freq, q_factor)
 * through jac (Bs, S, 15), the Jacobians dasp_biquad_design returned for the sections. */
int dasp_sos64_grads(const double* c5, const double* sos, const double* gc5, const double* jac, int Bs, int S, int mode, double* out,
                     void* stream) {
    if (!c5 || !gc5 || !out || Bs <= 0 || S <= 0 || (mode == 0 && !sos) || (mode == 1 && !jac) || mode < 0 || mode > 1) return DASP_ERR_ARG;
    const int n = Bs * S;
    hipLaunchKernelGGL(sos64_grads_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, c5, sos, gc5, jac, n, mode, out);
    return r64_check();
}

/* compressor (mode 0) / expander (mode 1) in fp64: x, y (B, C, N); ctl (B, 5) as for dasp_dynamics_forward; gsave (B, N) kept for backward or NULL */
int dasp_dynamics64_forward(int mode, const double* x, const double* ctl, double* y, double* gsave, int B, int C, long N, double sample_rate,
                            double eps, int lookahead, void* stream) {
    if (!x || !ctl || !y || B <= 0 || C <= 0 || N <= 0 || lookahead < 0 || (mode != 0 && mode != 1)) return DASP_ERR_ARG;
    if (mode == 0) hipLaunchKernelGGL(dyn64_fwd_kernel<0>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, ctl, y, gsave, B, C, N, lookahead, sample_rate, eps);
    else hipLaunchKernelGGL(dyn64_fwd_kernel<1>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, ctl, y, gsave, B, C, N, lookahead, sample_rate, eps);
    return r64_check();
}
int dasp_dynamics64_backward(int mode, const double* x, const double* ctl, const double* gy, const double* gsave, double* gx, double* gctl,
                             int B, int C, long N, double sample_rate, double eps, int lookahead, void* stream) {
    if (!x || !ctl || !gy || !gsave || !gx || !gctl || B <= 0 || C <= 0 || N <= 0 || lookahead < 0 || (mode != 0 && mode != 1)) return DASP_ERR_ARG;
    if (mode == 0) hipLaunchKernelGGL(dyn64_bwd_kernel<0>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, ctl, gy, gsave, gx, gctl, B, C, N, lookahead, sample_rate, eps);
    else hipLaunchKernelGGL(dyn64_bwd_kernel<1>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, ctl, gy, gsave, gx, gctl, B, C, N, lookahead, sample_rate, eps);
    return r64_check();
}

/* op 0: gain (ctl: B values), op 1: distortion (ctl: B*C values), fp64. backward: gctl is zeroed here. */
int dasp_ew64_forward(int op, const double* x, const double* ctl, double* y, int B, int C, long N, void* stream) {
    if (!x || !ctl || !y || B <= 0 || C <= 0 || N <= 0 || (op != 0 && op != 1)) return DASP_ERR_ARG;
    if (op == 0) hipLaunchKernelGGL((ew64_kernel<0, false>), dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, ctl, (const double*)nullptr, y, (double*)nullptr, C, N);
    else hipLaunchKernelGGL((ew64_kernel<1, false>), dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, ctl, (const double*)nullptr, y, (double*)nullptr, C, N);
    return r64_check();
}
int dasp_ew64_backward(int op, const double* x, const double* ctl, const double* gy, double* gx, double* gctl, int B, int C, long N, void* stream) {
    if (!x || !ctl || !gy || !gx || !gctl || B <= 0 || C <= 0 || N <= 0 || (op != 0 && op != 1)) return DASP_ERR_ARG;
    const hipError_t e = zero_async(gctl, sizeof(double) * (size_t)(op == 0 ? B : B * C), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (op == 0) hipLaunchKernelGGL((ew64_kernel<0, true>), dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, ctl, gy, gx, gctl, C, N);
    else hipLaunchKernelGGL((ew64_kernel<1, true>), dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, ctl, gy, gx, gctl, C, N);
    return r64_check();
}

}  // extern "C"
