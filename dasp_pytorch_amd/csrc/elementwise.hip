// gain and distortion: fused elementwise forward / backward with in-kernel parameter-gradient
// reduction. Replaces dasp_pytorch/functional.py:10-29 (gain) and :65-78 (distortion) and their
// autograd graphs.  HBM-bound: forward 8 B per channel-sample (read x, write y), backward 12 B
// (read x, read gy, write gx; the derivative is recomputed from x, y is never re-read).
//
// Layout: x, y, gy, gx (B, C, N) fp32 contiguous; a row = one (b, c) signal. Control values are
// indexed  row / cdiv  (gain: cdiv = C, one gain per batch item repeated over channels,
// functional.py:26-28;  distortion: cdiv = 1, one drive per (b, c) row, functional.py:78).
// Grid = rows x segments per row (1-D); a segment is SEG consecutive samples handled by one 256-thread
// workgroup with float4 accesses. Backward writes one partial sum per (row, segment); a finalize
// kernel reduces them in fp64 (deterministic, no float atomics).
#include "common.hpp"

namespace dasp {

constexpr int EW_THREADS = 256;
constexpr int EW_SEG = 8192;                 // samples per workgroup: 8 float4 per thread
constexpr float LN10_OVER_20 = 0.11512925464970228f;

enum { EW_GAIN = 0, EW_DIST = 1 };

// d tanh(u) / du = sech^2(u) = 4 e / (1 + e)^2, e = exp(-2 |u|): keeps its relative accuracy where tanh saturates (1 - tanh^2 loses
// all digits there to cancellation); the kernels are HBM-bound, the exponential is free
__device__ __forceinline__ float sech2(float u) {
    const float e = expf(-2.f * fabsf(u)), d = 1.f + e;
    return 4.f * e / (d * d);
}

template <int OP>
__device__ __forceinline__ float ew_fwd(float x, float lin) {
    return OP == EW_GAIN ? x * lin : tanhf(x * lin);
}
// returns d(loss)/dx, accumulates d(loss)/d(lin)*lin into acc (chain rule to dB applied in finalize)
template <int OP>
__device__ __forceinline__ float ew_bwd(float x, float g, float lin, float& acc) {
    if (OP == EW_GAIN) {
        acc = fmaf(g, x, acc);
        return g * lin;
    } else {
        const float t = g * sech2(x * lin);
        acc = fmaf(t, x, acc);
        return t * lin;
    }
}

template <int OP, bool BWD>
__global__ void __launch_bounds__(EW_THREADS)
ew_kernel(const float* __restrict__ x, const float* __restrict__ ctl_db, const float* __restrict__ gy,
          float* __restrict__ out, float* __restrict__ partials, int cdiv, long N, int nseg, int vec) {
    const int row = blockIdx.x / nseg, seg = blockIdx.x % nseg, tid = threadIdx.x;   // 1-D grid: no 65535-row limit
    const float lin = exp10f(ctl_db[row / cdiv] * 0.05f);
    const long base = (long)row * N, s0 = (long)seg * EW_SEG;
    const long s1 = s0 + EW_SEG < N ? s0 + EW_SEG : N;
    float acc = 0.f;
    if (vec && s1 - s0 == EW_SEG) {
#pragma unroll
        for (int j = 0; j < EW_SEG / (4 * EW_THREADS); ++j) {
            const long i = base + s0 + (long)(j * EW_THREADS + tid) * 4;
            // backward: two read streams + one write stream -> streaming hints (0.160 -> 0.132 ms); forward is as fast without them
            const f4 xv = BWD ? ld_stream(reinterpret_cast<const f4*>(x + i)) : *reinterpret_cast<const f4*>(x + i);
            f4 o;
            if (BWD) {
                const f4 g = ld_stream(reinterpret_cast<const f4*>(gy + i));
                o.x = ew_bwd<OP>(xv.x, g.x, lin, acc); o.y = ew_bwd<OP>(xv.y, g.y, lin, acc);
                o.z = ew_bwd<OP>(xv.z, g.z, lin, acc); o.w = ew_bwd<OP>(xv.w, g.w, lin, acc);
            } else {
                o.x = ew_fwd<OP>(xv.x, lin); o.y = ew_fwd<OP>(xv.y, lin); o.z = ew_fwd<OP>(xv.z, lin); o.w = ew_fwd<OP>(xv.w, lin);
            }
            if (BWD) st_stream(reinterpret_cast<f4*>(out + i), o); else *reinterpret_cast<f4*>(out + i) = o;
        }
    } else {
        for (long n = s0 + tid; n < s1; n += EW_THREADS)
            out[base + n] = BWD ? ew_bwd<OP>(x[base + n], gy[base + n], lin, acc) : ew_fwd<OP>(x[base + n], lin);
    }
    if (BWD) {
        __shared__ float red[EW_THREADS / 64];
        const float w = wave_sum(acc);
        if (lane_id() == 0) red[wave_id()] = w;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < EW_THREADS / 64; ++i) s += red[i];
            partials[(size_t)row * nseg + seg] = s * lin;
        }
    }
}

// gctl[i] = ln10/20 * sum over the cdiv rows of control i and their segments of partials
__global__ void ew_finalize_kernel(const float* __restrict__ partials, float* __restrict__ gctl, int nctl, int cdiv, int nseg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nctl) return;
    double s = 0.0;
    const float* p = partials + (size_t)i * cdiv * nseg;
    for (int k = 0; k < cdiv * nseg; ++k) s += (double)p[k];
    gctl[i] = (float)(s * (double)LN10_OVER_20);
}

// distortion with one drive value PER SAMPLE (the reference's drive_db.view(bs, chs, -1) with bs * chs * seq_len values, functional.py:78):
// plain elementwise, n = B * C * N; forward 12 B per sample, backward 20 B (read x, drive, gy; write gx, gdrive)
template <bool BWD>
__global__ void __launch_bounds__(EW_THREADS)
dist_sample_kernel(const float* __restrict__ x, const float* __restrict__ drive_db, const float* __restrict__ gy, float* __restrict__ out,
                   float* __restrict__ gdrive, long n, int vec) {
    const long t = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    auto one = [&](float xv, float dv, float gv, float& o, float& gd) {
        const float lin = exp10f(dv * 0.05f);
        if (BWD) {
            const float tt = gv * sech2(xv * lin) * lin;
            o = tt; gd = tt * xv * LN10_OVER_20;
        } else {
            o = tanhf(xv * lin);
        }
    };
    if (vec) {
        const long i = t * 4;
        if (i >= n) return;
        const f4 xv = *reinterpret_cast<const f4*>(x + i), dv = *reinterpret_cast<const f4*>(drive_db + i);
        f4 gv = f4{0.f, 0.f, 0.f, 0.f};
        if (BWD) gv = *reinterpret_cast<const f4*>(gy + i);
        float o[4], gd[4];
        one(xv.x, dv.x, gv.x, o[0], gd[0]); one(xv.y, dv.y, gv.y, o[1], gd[1]); one(xv.z, dv.z, gv.z, o[2], gd[2]); one(xv.w, dv.w, gv.w, o[3], gd[3]);
        *reinterpret_cast<f4*>(out + i) = f4{o[0], o[1], o[2], o[3]};
        if (BWD) *reinterpret_cast<f4*>(gdrive + i) = f4{gd[0], gd[1], gd[2], gd[3]};
    } else if (t < n) {
        float o, gd;
        one(x[t], drive_db[t], BWD ? gy[t] : 0.f, o, gd);
        out[t] = o;
        if (BWD) gdrive[t] = gd;
    }
}

// Controls of the reference's effect chain (examples/style_transfer.py:150-154: EQ -> compressor -> reverb -> gain) from the normalised
// parameter tensors of Processor.process_normalized (modules.py:25-51) in one launch per direction: de-normalisation lo + span * p of the
// compressor's (B, 6), the reverb's (B, 25) and the gain's (B, 1) parameters, written in the layouts the kernels take - ctl (B, 5)
// [threshold, ratio, attack, knee, make-up + gain] (the chain's final gain folded into the make-up gain; release_ms is unused,
// functional.py:340), band gains (B, 12), band decays (B, 12), mix (B) - and the adjoint map back to the three parameter tensors.
// As separate torch ops this is ~30 launches of 3-4 us per training step.
struct ChainAffine { float lo[32], span[32]; };        // [0, 6) compressor, [6, 31) reverb, [31] gain

// flag (may be null): bit i of *flag is set when column i of the 32 (compressor 0 - 5, reverb 6 - 30, gain 31) holds a value outside [0, 1] -
// the reference's ValueError (modules.py:83-84), raised by the host after it has read the word back; NaN passes, as it does there
__device__ __forceinline__ void unit_range_flag(unsigned* flag, float p, int col) {
    if (flag && (p < 0.f || p > 1.f)) atomicOr(flag, 1u << col);
}
__global__ void chain_controls_kernel(const float* __restrict__ pc, const float* __restrict__ pr, const float* __restrict__ pg, ChainAffine a,
                                      float* __restrict__ ctl, float* __restrict__ gains, float* __restrict__ decays, float* __restrict__ mix,
                                      unsigned* __restrict__ flag, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = t / 30, j = t % 30;
    if (b >= B) return;
    if (j < 5) {
        const int c = j < 3 ? j : j + 1;
        const float p = pc[b * 6 + c];
        unit_range_flag(flag, p, c);
        if (j == 2) unit_range_flag(flag, pc[b * 6 + 3], 3);      // release_ms: checked like the others although nothing reads it
        float v = fmaf(a.span[c], p, a.lo[c]);
        if (j == 4) {
            unit_range_flag(flag, pg[b], 31);
            v += fmaf(a.span[31], pg[b], a.lo[31]);
        }
        ctl[b * 5 + j] = v;
    } else {
        const int c = j - 5;                              // reverb column 0 .. 24
        const float p = pr[b * 25 + c];
        unit_range_flag(flag, p, 6 + c);
        const float v = fmaf(a.span[6 + c], p, a.lo[6 + c]);
        if (c < 12) gains[b * 12 + c] = v; else if (c < 24) decays[b * 12 + c - 12] = v; else mix[b] = v;
    }
}
__global__ void chain_controls_backward_kernel(const float* __restrict__ gctl, const float* __restrict__ ggain, const float* __restrict__ gdecay,
                                               const float* __restrict__ gmix, ChainAffine a, float* __restrict__ gpc, float* __restrict__ gpr,
                                               float* __restrict__ gpg, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = t / 32, j = t % 32;
    if (b >= B) return;
    if (j < 6) {
        const int k = j < 3 ? j : j - 1;                  // row of ctl; column 3 (release_ms) has no path to the output
        gpc[b * 6 + j] = j == 3 ? 0.f : gctl[b * 5 + k] * a.span[j];
    } else if (j < 31) {
        const int c = j - 6;
        const float g = c < 12 ? ggain[b * 12 + c] : (c < 24 ? gdecay[b * 12 + c - 12] : gmix[b]);
        gpr[b * 25 + c] = g * a.span[j];
    } else {
        gpg[b] = gctl[b * 5 + 4] * a.span[31];
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
inline int ew_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
inline bool ew_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int ew_nseg(long N) { return (int)((N + EW_SEG - 1) / EW_SEG); }

template <int OP>
int ew_forward(const float* x, const float* ctl, float* y, int B, int C, long N, void* stream) {
    if (!x || !ctl || !y || B <= 0 || C <= 0 || N <= 0) return DASP_ERR_ARG;
    const long rows = (long)B * C;
    const int nseg = ew_nseg(N), vec = (N % 4 == 0) && ew_al16(x) && ew_al16(y);
    if (rows * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((ew_kernel<OP, false>), dim3((unsigned)(rows * nseg)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, ctl, nullptr, y,
                       nullptr, OP == EW_GAIN ? C : 1, N, nseg, vec);
    return ew_check();
}
template <int OP>
int ew_backward(const float* x, const float* ctl, const float* gy, float* gx, float* gctl, float* partials, int B, int C, long N,
                void* stream) {
    if (!x || !ctl || !gy || !gx || !gctl || !partials || B <= 0 || C <= 0 || N <= 0) return DASP_ERR_ARG;
    const long rows = (long)B * C;
    const int nseg = ew_nseg(N), vec = (N % 4 == 0) && ew_al16(x) && ew_al16(gy) && ew_al16(gx);
    if (rows * nseg > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int cdiv = OP == EW_GAIN ? C : 1, nctl = (int)(rows / cdiv);
    hipLaunchKernelGGL((ew_kernel<OP, true>), dim3((unsigned)(rows * nseg)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, ctl, gy, gx, partials,
                       cdiv, N, nseg, vec);
    int st = ew_check();
    if (st != DASP_OK) return st;
    hipLaunchKernelGGL(ew_finalize_kernel, dim3((nctl + 127) / 128), dim3(128), 0, (hipStream_t)stream, partials, gctl, nctl, cdiv, nseg);
    return ew_check();
}
}  // namespace

extern "C" {
long dasp_ew_partial_floats(long rows, long N) { return rows * ew_nseg(N); }
int dasp_gain_forward(const float* x, const float* gain_db, float* y, int B, int C, long N, void* stream) {
    return ew_forward<EW_GAIN>(x, gain_db, y, B, C, N, stream);
}
int dasp_gain_backward(const float* x, const float* gain_db, const float* gy, float* gx, float* ggain, float* partials, int B, int C,
                       long N, void* stream) {
    return ew_backward<EW_GAIN>(x, gain_db, gy, gx, ggain, partials, B, C, N, stream);
}
int dasp_distortion_forward(const float* x, const float* drive_db, float* y, int B, int C, long N, void* stream) {
    return ew_forward<EW_DIST>(x, drive_db, y, B, C, N, stream);
}
int dasp_distortion_backward(const float* x, const float* drive_db, const float* gy, float* gx, float* gdrive, float* partials, int B,
                             int C, long N, void* stream) {
    return ew_backward<EW_DIST>(x, drive_db, gy, gx, gdrive, partials, B, C, N, stream);
}
int dasp_distortion_sample_forward(const float* x, const float* drive_db, float* y, long n, void* stream) {
    if (!x || !drive_db || !y || n <= 0) return DASP_ERR_ARG;
    const int vec = (n % 4 == 0) && ew_al16(x) && ew_al16(drive_db) && ew_al16(y);
    const long threads = vec ? n / 4 : n, blocks = (threads + EW_THREADS - 1) / EW_THREADS;
    if (blocks > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dist_sample_kernel<false>, dim3((unsigned)blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, drive_db, nullptr, y, nullptr, n, vec);
    return ew_check();
}
int dasp_distortion_sample_backward(const float* x, const float* drive_db, const float* gy, float* gx, float* gdrive, long n, void* stream) {
    if (!x || !drive_db || !gy || !gx || !gdrive || n <= 0) return DASP_ERR_ARG;
    const int vec = (n % 4 == 0) && ew_al16(x) && ew_al16(drive_db) && ew_al16(gy) && ew_al16(gx) && ew_al16(gdrive);
    const long threads = vec ? n / 4 : n, blocks = (threads + EW_THREADS - 1) / EW_THREADS;
    if (blocks > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dist_sample_kernel<true>, dim3((unsigned)blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, x, drive_db, gy, gx, gdrive, n, vec);
    return ew_check();
}
int dasp_chain_controls(const float* comp_pn, const float* reverb_pn, const float* gain_pn, const float* lo, const float* span, float* ctl,
                        float* gains, float* decays, float* mix, unsigned* flag, int B, void* stream) {
    if (!comp_pn || !reverb_pn || !gain_pn || !lo || !span || !ctl || !gains || !decays || !mix || B <= 0) return DASP_ERR_ARG;
    ChainAffine a;
    for (int i = 0; i < 32; ++i) { a.lo[i] = lo[i]; a.span[i] = span[i]; }
    hipLaunchKernelGGL(chain_controls_kernel, dim3((B * 30 + 255) / 256), dim3(256), 0, (hipStream_t)stream, comp_pn, reverb_pn, gain_pn, a, ctl, gains,
                       decays, mix, flag, B);
    return ew_check();
}
int dasp_chain_controls_backward(const float* gctl, const float* ggain, const float* gdecay, const float* gmix, const float* span,
                                 float* gcomp_pn, float* greverb_pn, float* ggain_pn, int B, void* stream) {
    if (!gctl || !ggain || !gdecay || !gmix || !span || !gcomp_pn || !greverb_pn || !ggain_pn || B <= 0) return DASP_ERR_ARG;
    ChainAffine a;
    for (int i = 0; i < 32; ++i) { a.lo[i] = 0.f; a.span[i] = span[i]; }
    hipLaunchKernelGGL(chain_controls_backward_kernel, dim3((B * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, gctl, ggain, gdecay, gmix, a,
                       gcomp_pn, greverb_pn, ggain_pn, B);
    return ew_check();
}
}  // extern "C"
