// Noise-shaped reverberation: shaped-noise impulse response + long convolution, forward and adjoint,
// as batched real FFTs (hipFFT/rocFFT) with fused spectral-multiply, overlap-add and reduction kernels.
//
// Replaces dasp_pytorch.functional.noise_shaped_reverberation (dasp_pytorch/functional.py:406-577):
//   wn_filt = grouped 1023-tap FIR of white noise, 12 bands          (:548-558, direct conv1d)
//   ir      = mean_band(wn_filt * exp(-(10 decay + 1) t) * gain)     (:561-567)
//   y_wet   = causal convolution of x with ir, truncated to N        (:570-572, direct conv1d, 65536 taps)
//   y       = (1 - mix) x + mix y_wet                                (:575)
// The reference's direct convolutions are 99.6 % of its time. Here both are
// frequency-domain products. The filter bank (short filters, 2B*12 independent noise rows) is ONE fused kernel:
// overlap-save windows of 4096 samples, the left/right rows of an item packed as one complex signal, forward FFT,
// product with the band's spectrum and inverse FFT inside the workgroup (fft_lds.hpp), envelope / gain / band mean
// applied to the result in registers - the filtered noise never exists in HBM (the backward pass re-runs the same
// kernel with the impulse-response gradient as a weight instead of saving it). The 65536-tap convolution is
// overlap-add over blocks of Lb samples with n1 = 2 Lb point FFTs (hipFFT/rocFFT).
//
// The FFT library is bound at run time (dasp_fft_init dlopens the libhipfft the host process
// already uses, so a process never holds two copies); plans are cached per (length, batch).
#include "common.hpp"
#include "fft_lds.hpp"
#include <dlfcn.h>
#include <hipfft/hipfft.h>
#include <map>
#include <mutex>

namespace dasp {

constexpr int RV_BANDS_MAX = 16;
typedef float2 cpx;

__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// dst[row][i] = i < len(row) ? scale(row) * src[off(row) + i] : 0     rows of n1 floats
//   mode 0: plain rows           off = row * src_stride, len = src_len
//   mode 1: signal blocks        row = (sig * nblk + k): off = sig * N + k * Lb, len = min(Lb, N - k Lb); scale = sc[sig / 2] or 1
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int n1, int mode, long src_stride,
                                int src_len, long N, int Lb, int nblk, const float* __restrict__ sc) {
    const long row = blockIdx.y;
    long off; int len; float s = 1.f;
    if (mode == 0) { off = row * src_stride; len = src_len; }
    else {
        const long sig = row / nblk; const int k = (int)(row % nblk);
        off = sig * N + (long)k * Lb;
        const long rem = N - (long)k * Lb;
        len = rem < Lb ? (int)(rem > 0 ? rem : 0) : Lb;
        if (sc) s = sc[sig / 2];
    }
    float* d = dst + row * (long)n1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) d[i] = i < len ? s * src[off + i] : 0.f;
}

// out[row][f] = A[row][f] * (conj?) Bs[(row / div) % mod][f]
__global__ void cmul_rows_kernel(const cpx* __restrict__ A, const cpx* __restrict__ Bs, cpx* __restrict__ out, int nfreq, int div, int mod,
                                 int conjB) {
    const long row = blockIdx.y;
    const cpx* a = A + row * (long)nfreq;
    const cpx* b = Bs + ((row / div) % mod) * (long)nfreq;
    cpx* o = out + row * (long)nfreq;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nfreq; f += gridDim.x * blockDim.x) o[f] = conjB ? cmulc(a[f], b[f]) : cmul(a[f], b[f]);
}

// ---- fused filter bank ---------------------------------------------------------------------------------------------
// spec layout (complex): [0, 4096) forward twiddles exp(-2 pi i e / 4096); then nb rows of 4096: conj(FFT(filter_band)) / 4096.
__global__ void fb_twiddle_kernel(f2* __restrict__ spec) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < FFT_N) {
        double sn, cs;
        sincospi(2.0 * (double)e / (double)FFT_N, &sn, &cs);
        spec[e] = f2{(float)cs, (float)-sn};
    }
}
__global__ __launch_bounds__(FFT_T) void fb_spectrum_kernel(const float* __restrict__ filters, f2* __restrict__ spec, int taps) {
    __shared__ f2 lds[FFT_LDS];
    const int j = threadIdx.x, band = blockIdx.x;
    float r[8], i[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int idx = j + 512 * q; r[q] = idx < taps ? filters[(long)band * taps + idx] : 0.f; i[q] = 0.f; }
    const FftTw tw = fft_twiddles(j, spec);
    fft4096<-1>(r, i, j, tw, lds);
    constexpr float inv = 1.f / (float)FFT_N;
#pragma unroll
    for (int q = 0; q < 8; ++q) spec[FFT_N + (long)band * FFT_N + j + 512 * q] = f2{r[q] * inv, -i[q] * inv};
}

// One workgroup = one (batch item b, window w): output samples n = w V + idx, idx < V = 512 VQ <= 4096 - (taps - 1).
//   z_band[idx] = noise[b,0,band][n0 + idx] + i noise[b,1,band][n0 + idx]          (functional.py:548; both rows share the band filter)
//   o_band     = IFFT(FFT(z_band) conj(F_band))  -> valid cross-correlations for idx < V   (:551-558)
//   MODE 0:  ir[b,c][n] = 1/nb sum_band gain env_band(t_n) o_band                   (:561-567)
//   MODE 1:  part[(b, w), band] = (sum_n gir o env / nb,  sum_n gir o env gain (-10 t_n) / nb),  gir = (p[n] + q[n + Lb]) * pq_scale
template <int MODE>
__global__ __launch_bounds__(FFT_T, 4) void fb_fused_kernel(const float* __restrict__ noise, const f2* __restrict__ spec, const float* __restrict__ gains,
                                                         const float* __restrict__ decays, float* __restrict__ ir_pad, const float* __restrict__ p,
                                                         const float* __restrict__ qq, float* __restrict__ part, int nb, int L, int taps, int n1,
                                                         int Lb, int VQ, float pq_scale) {
    __shared__ f2 lds[FFT_LDS];
    __shared__ float red[FFT_T / 64][RV_BANDS_MAX][2];
    const int j = threadIdx.x, w = blockIdx.x, b = blockIdx.y;
    const int V = VQ * 512, n0 = w * V, row_len = L + taps - 1;
    const float tstep = L > 1 ? 1.f / (float)(L - 1) : 0.f, inv_nb = 1.f / (float)nb;
    const FftTw tw = fft_twiddles(j, spec);
    const float t0 = (float)(n0 + j) * tstep, t512 = 512.f * tstep;      // t_n = n / (L - 1), torch.linspace(0, 1, L)
    float accr[8], acci[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = n0 + j + 512 * q;
        accr[q] = 0.f; acci[q] = 0.f;
        if (MODE == 1 && q < VQ && n < L) {          // weights: gradient w.r.t. the two impulse responses of this item
            accr[q] = (p[(long)(2 * b) * n1 + n] + qq[(long)(2 * b) * n1 + n + Lb]) * pq_scale;
            acci[q] = (p[(long)(2 * b + 1) * n1 + n] + qq[(long)(2 * b + 1) * n1 + n + Lb]) * pq_scale;
        }
    }
    for (int band = 0; band < nb; ++band) {
        float r[8], i[8];
        {
            const float* rl = noise + ((long)(2 * b) * nb + band) * row_len;
            const float* rr = noise + ((long)(2 * b + 1) * nb + band) * row_len;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = n0 + j + 512 * q;
                r[q] = idx < row_len ? rl[idx] : 0.f;
                i[q] = idx < row_len ? rr[idx] : 0.f;
            }
        }
        fft4096<-1>(r, i, j, tw, lds);
        const f2* F = spec + FFT_N + (long)band * FFT_N;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f2 f = F[j + 512 * q];
            const float t = r[q] * f.x - i[q] * f.y;
            i[q] = r[q] * f.y + i[q] * f.x;
            r[q] = t;
        }
        fft4096<1>(r, i, j, tw, lds);
        const float g = gains[b * nb + band], d = 10.f * decays[b * nb + band] + 1.f;
        if (MODE == 0) {
            const float gs = g * inv_nb;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float e = __expf(-d * fmaf((float)q, t512, t0)) * gs;
                accr[q] = fmaf(e, r[q], accr[q]);
                acci[q] = fmaf(e, i[q], acci[q]);
            }
        } else {
            float sg = 0.f, sd = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float tq = fmaf((float)q, t512, t0);
                const float e = (accr[q] * r[q] + acci[q] * i[q]) * __expf(-d * tq) * inv_nb;     // weights are zero outside the valid range
                sg += e;
                sd = fmaf(e, -10.f * tq * g, sd);
            }
            sg = wave_sum(sg); sd = wave_sum(sd);
            if (lane_id() == 0) { red[wave_id()][band][0] = sg; red[wave_id()][band][1] = sd; }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = n0 + j + 512 * q;
            if (q < VQ && n < L) { ir_pad[(long)(2 * b) * n1 + n] = accr[q]; ir_pad[(long)(2 * b + 1) * n1 + n] = acci[q]; }
        }
    } else {
        __syncthreads();
        if (j < nb * 2) {
            const int band = j >> 1, k = j & 1;
            float a = 0.f;
            for (int v = 0; v < FFT_T / 64; ++v) a += red[v][band][k];
            part[(((long)b * gridDim.x + w) * nb + band) * 2 + k] = a;
        }
    }
}

// zero the tail [L, n1) of every impulse-response row (the head is written by fb_fused_kernel<0>)
__global__ void zero_tail_kernel(float* __restrict__ ir_pad, int L, int n1) {
    float* d = ir_pad + (long)blockIdx.y * n1;
    for (int n = L + blockIdx.x * blockDim.x + threadIdx.x; n < n1; n += gridDim.x * blockDim.x) d[n] = 0.f;
}

// y[b,c,n] = (1 - mix) x + mix * (z[k][r] + z[k-1][r + Lb]) / n1      n = k Lb + r
__global__ void ola_mix_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ mix, float* __restrict__ y,
                               long N, int Lb, int nblk, int n1) {
    const long sig = blockIdx.y;
    const float m = mix[sig >> 1], inv = 1.f / (float)n1;
    for (long n = blockIdx.x * (long)blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
        const int k = (int)(n / Lb), r = (int)(n - (long)k * Lb);
        float wet = z[(sig * nblk + k) * (long)n1 + r];
        if (k > 0) wet += z[(sig * nblk + k - 1) * (long)n1 + r + Lb];
        const float xv = x[sig * N + n];
        y[sig * N + n] = fmaf(m, wet * inv - xv, xv);
    }
}

// gx[b,c,n] = (1 - mix) gy + (c_k[r] + c_{k+1}[r + Lb]) / n1 ;  also partial sums of gy * (y_wet - x) for d/dmix
__global__ void bwd_combine_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ cc,
                                   const float* __restrict__ mix, float* __restrict__ gx, float* __restrict__ mix_part, long N, int Lb, int nblk,
                                   int n1) {
    const long sig = blockIdx.y;
    const float m = mix[sig >> 1], inv = 1.f / (float)n1;
    float acc = 0.f;
    for (long n = blockIdx.x * (long)blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
        const int k = (int)(n / Lb), r = (int)(n - (long)k * Lb);
        float wet = z[(sig * nblk + k) * (long)n1 + r];
        if (k > 0) wet += z[(sig * nblk + k - 1) * (long)n1 + r + Lb];
        float c = cc[(sig * nblk + k) * (long)n1 + r];
        if (k + 1 < nblk) c += cc[(sig * nblk + k + 1) * (long)n1 + r + Lb];
        const float g = gy[sig * N + n], xv = x[sig * N + n];
        gx[sig * N + n] = fmaf(1.f - m, g, c * inv);     // cc already carries the factor mix (it is the adjoint of mix * gy)
        acc = fmaf(g, wet * inv - xv, acc);
    }
    __shared__ float red[4];
    const float w = wave_sum(acc);
    if (lane_id() == 0) red[wave_id()] = w;
    __syncthreads();
    if (threadIdx.x == 0) mix_part[sig * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// P[sig][f] = sum_k G[k][f] conj(X[k][f]) ;  Q[sig][f] = sum_{k>=1} G[k][f] conj(X[k-1][f])
__global__ void ir_grad_spec_kernel(const cpx* __restrict__ G, const cpx* __restrict__ X, cpx* __restrict__ P, cpx* __restrict__ Q, int nfreq, int nblk) {
    const long sig = blockIdx.y;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nfreq; f += gridDim.x * blockDim.x) {
        cpx p = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        for (int k = 0; k < nblk; ++k) {
            const cpx g = G[(sig * nblk + k) * (long)nfreq + f];
            const cpx a = cmulc(g, X[(sig * nblk + k) * (long)nfreq + f]);
            p.x += a.x; p.y += a.y;
            if (k > 0) { const cpx c = cmulc(g, X[(sig * nblk + k - 1) * (long)nfreq + f]); q.x += c.x; q.y += c.y; }
        }
        P[sig * (long)nfreq + f] = p; Q[sig * (long)nfreq + f] = q;
    }
}

// ggain, gdecay (B, nb) and gmix (B) from the per-block partial sums, in fp64
__global__ void reverb_finalize_kernel(const float* __restrict__ part, const float* __restrict__ mix_part, float* __restrict__ ggain,
                                       float* __restrict__ gdecay, float* __restrict__ gmix, int B, int nb, int chunks, int mix_chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * nb) {
        const int b = i / nb, k = i % nb;
        double a = 0.0, c = 0.0;
        for (int j = 0; j < chunks; ++j) {
            const float* p = part + (((long)b * chunks + j) * nb + k) * 2;
            a += (double)p[0]; c += (double)p[1];
        }
        ggain[i] = (float)a; gdecay[i] = (float)c;
    }
    if (i < B) {
        double m = 0.0;
        for (int j = 0; j < 2 * mix_chunks; ++j) m += (double)mix_part[(long)i * 2 * mix_chunks + j];
        gmix[i] = (float)m;
    }
}

// ------------------------------------------------------------------------------------------------
// run-time binding of hipFFT
struct FftApi {
    void* handle = nullptr;
    hipfftResult (*Plan1d)(hipfftHandle*, int, hipfftType, int) = nullptr;
    hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
    hipfftResult (*ExecR2C)(hipfftHandle, hipfftReal*, hipfftComplex*) = nullptr;
    hipfftResult (*ExecC2R)(hipfftHandle, hipfftComplex*, hipfftReal*) = nullptr;
    std::map<std::tuple<int, int, int>, hipfftHandle> plans;
    std::mutex mu;
};
static FftApi g_fft;

static int fft_plan(int n, int type, int batch, hipfftHandle* out) {
    std::lock_guard<std::mutex> lk(g_fft.mu);
    if (!g_fft.handle) return DASP_ERR_UNSUPPORTED;
    auto key = std::make_tuple(n, type, batch);
    auto it = g_fft.plans.find(key);
    if (it == g_fft.plans.end()) {
        hipfftHandle h;
        if (g_fft.Plan1d(&h, n, (hipfftType)type, batch) != HIPFFT_SUCCESS) return DASP_ERR_UNSUPPORTED;
        it = g_fft.plans.emplace(key, h).first;
    }
    *out = it->second;
    return DASP_OK;
}
static int fft_r2c(int n, long batch, float* in, cpx* out, hipStream_t st) {
    hipfftHandle h;
    int s = fft_plan(n, HIPFFT_R2C, (int)batch, &h);
    if (s) return s;
    if (g_fft.SetStream(h, st) != HIPFFT_SUCCESS) return DASP_ERR_UNSUPPORTED;
    return g_fft.ExecR2C(h, in, reinterpret_cast<hipfftComplex*>(out)) == HIPFFT_SUCCESS ? DASP_OK : DASP_ERR_UNSUPPORTED;
}
static int fft_c2r(int n, long batch, cpx* in, float* out, hipStream_t st) {
    hipfftHandle h;
    int s = fft_plan(n, HIPFFT_C2R, (int)batch, &h);
    if (s) return s;
    if (g_fft.SetStream(h, st) != HIPFFT_SUCCESS) return DASP_ERR_UNSUPPORTED;
    return g_fft.ExecC2R(h, reinterpret_cast<hipfftComplex*>(in), out) == HIPFFT_SUCCESS ? DASP_OK : DASP_ERR_UNSUPPORTED;
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
inline int rv_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
inline long next_pow2(long v) { long p = 1; while (p < v) p <<= 1; return p; }
struct RvDims { int Lb, n1, nfreq, nblk, VQ, nwin; long R; };
inline RvDims rv_dims(int B, long N, int L, int taps) {
    RvDims d;
    d.Lb = (int)next_pow2(L > taps ? L : taps);
    d.n1 = 2 * d.Lb;
    d.nfreq = d.n1 / 2 + 1;
    d.nblk = (int)((N + d.Lb - 1) / d.Lb);
    d.R = 2L * B;
    d.VQ = (FFT_N - (taps - 1)) / 512;          // valid outputs per filter-bank window = 512 VQ (0: filters too long for the window)
    if (d.VQ < 0) d.VQ = 0;
    d.nwin = d.VQ ? (L + 512 * d.VQ - 1) / (512 * d.VQ) : 0;
    return d;
}
constexpr int RV_T = 256;
inline dim3 rv_grid(long n, long rows, int cap = 64) {
    long bx = (n + RV_T - 1) / RV_T;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)rows);
}
#define RV_TRY(expr) do { int s_ = (expr); if (s_ != DASP_OK) return s_; } while (0)
}  // namespace

extern "C" {

int dasp_fft_init(const char* libhipfft_path) {
    std::lock_guard<std::mutex> lk(g_fft.mu);
    if (g_fft.handle) return DASP_OK;
    void* h = dlopen(libhipfft_path ? libhipfft_path : "libhipfft.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return DASP_ERR_UNSUPPORTED;
    g_fft.Plan1d = reinterpret_cast<decltype(g_fft.Plan1d)>(dlsym(h, "hipfftPlan1d"));
    g_fft.SetStream = reinterpret_cast<decltype(g_fft.SetStream)>(dlsym(h, "hipfftSetStream"));
    g_fft.ExecR2C = reinterpret_cast<decltype(g_fft.ExecR2C)>(dlsym(h, "hipfftExecR2C"));
    g_fft.ExecC2R = reinterpret_cast<decltype(g_fft.ExecC2R)>(dlsym(h, "hipfftExecC2R"));
    if (!g_fft.Plan1d || !g_fft.SetStream || !g_fft.ExecR2C || !g_fft.ExecC2R) return DASP_ERR_UNSUPPORTED;
    g_fft.handle = h;
    return DASP_OK;
}
int dasp_fft_ready(void) { return g_fft.handle != nullptr; }

/* sizes[0] = Lb (block length), [1] = n1 (FFT length), [2] = nfreq, [3] = nblk,
 * [4] = complex elements of Fspec (twiddles + band spectra of the filter bank), [5] = filter-bank windows per batch item,
 * [6] = floats of z / xpad (2B*nblk rows of n1), [7] = complex elements of Xf (2B*nblk rows of nfreq),
 * [8] = floats of ir_pad (2B rows of n1), [9] = complex elements of H (2B rows of nfreq),
 * [10] = mix partial chunks per signal, [11] = floats of the gain/decay partial sums */
int dasp_reverb_sizes(int B, long N, int L, int taps, int nb, long* sizes) {
    if (!sizes || B <= 0 || N <= 0 || L <= 0 || taps <= 0 || nb <= 0 || nb > RV_BANDS_MAX) return DASP_ERR_ARG;
    const RvDims d = rv_dims(B, N, L, taps);
    if (d.VQ < 1) return DASP_ERR_UNSUPPORTED;      // filters longer than 3585 taps do not fit the 4096-point window
    sizes[0] = d.Lb; sizes[1] = d.n1; sizes[2] = d.nfreq; sizes[3] = d.nblk;
    sizes[4] = (long)(nb + 1) * FFT_N; sizes[5] = d.nwin;
    sizes[6] = d.R * d.nblk * d.n1; sizes[7] = d.R * d.nblk * d.nfreq;
    sizes[8] = d.R * d.n1; sizes[9] = d.R * d.nfreq;
    sizes[10] = 64; sizes[11] = (long)B * d.nwin * nb * 2;
    return DASP_OK;
}

/* filters (nb, taps) fp32 -> Fspec (sizes[4] complex): the 4096-point twiddle table followed by conj(FFT(filter)) / 4096 per band */
int dasp_reverb_filter_spectrum(const float* filters, int nb, int taps, void* Fspec, void* stream) {
    if (!filters || !Fspec || nb <= 0 || nb > RV_BANDS_MAX || taps <= 0) return DASP_ERR_ARG;
    if (taps - 1 > FFT_N - 512) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fb_twiddle_kernel, dim3(FFT_N / 256), dim3(256), 0, (hipStream_t)stream, (f2*)Fspec);
    hipLaunchKernelGGL(fb_spectrum_kernel, dim3(nb), dim3(FFT_T), 0, (hipStream_t)stream, filters, (f2*)Fspec, taps);
    return rv_check();
}

/* Forward.  x (B,2,N); noise (2B, nb, L+taps-1); Fspec (sizes[4] complex); gains, decays (B, nb); mix (B); y (B,2,N).
 * Saved for backward: Xf (sizes[7] complex), H (sizes[9] complex), z (sizes[6] floats) (and the caller's noise, Fspec).
 * Scratch: yspec (sizes[7] complex), ir_pad (sizes[8] floats). */
int dasp_reverb_forward(const float* x, const float* noise, const void* Fspec, const float* gains, const float* decays, const float* mix,
                        float* y, void* Xf, void* H, float* z, void* yspec, float* ir_pad, int B, long N, int L, int taps, int nb,
                        void* stream) {
    if (!x || !noise || !Fspec || !gains || !decays || !mix || !y || !Xf || !H || !z || !yspec || !ir_pad || B <= 0 || N <= 0 || L <= 0 ||
        taps <= 0 || nb <= 0 || nb > RV_BANDS_MAX)
        return DASP_ERR_ARG;
    const RvDims d = rv_dims(B, N, L, taps);
    hipStream_t st = (hipStream_t)stream;
    const long xrows = d.R * d.nblk;
    if (d.VQ < 1 || B > 65535 || xrows > 65535) return DASP_ERR_UNSUPPORTED;
    // 1 + 2. filter bank, envelope, gains, mean over bands -> zero-padded impulse responses (functional.py:551-567), then their spectra
    hipLaunchKernelGGL(fb_fused_kernel<0>, dim3((unsigned)d.nwin, (unsigned)B), dim3(FFT_T), 0, st, noise, (const f2*)Fspec, gains, decays, ir_pad,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, nb, L, taps, d.n1, d.Lb, d.VQ, 0.f);
    hipLaunchKernelGGL(zero_tail_kernel, rv_grid(d.n1 - L, d.R), dim3(RV_T), 0, st, ir_pad, L, d.n1);
    RV_TRY(rv_check());
    RV_TRY(fft_r2c(d.n1, d.R, ir_pad, (cpx*)H, st));
    // 3. overlap-add convolution (:570-572) and wet/dry mix (:575)
    hipLaunchKernelGGL(pad_rows_kernel, rv_grid(d.n1, xrows), dim3(RV_T), 0, st, x, z, xrows, d.n1, 1, 0L, 0, N, d.Lb, d.nblk, (const float*)nullptr);
    RV_TRY(rv_check());
    RV_TRY(fft_r2c(d.n1, xrows, z, (cpx*)Xf, st));
    hipLaunchKernelGGL(cmul_rows_kernel, rv_grid(d.nfreq, xrows), dim3(RV_T), 0, st, (const cpx*)Xf, (const cpx*)H, (cpx*)yspec, d.nfreq, d.nblk, (int)d.R, 0);
    RV_TRY(rv_check());
    RV_TRY(fft_c2r(d.n1, xrows, (cpx*)yspec, z, st));
    hipLaunchKernelGGL(ola_mix_kernel, rv_grid(N, d.R, 256), dim3(RV_T), 0, st, x, z, mix, y, N, d.Lb, d.nblk, d.n1);
    return rv_check();
}

/* Backward.  gx (B,2,N); ggain, gdecay (B, nb); gmix (B).
 * Scratch: gpad / cc (sizes[6] floats), Gf (sizes[7] complex), cspec (sizes[7] complex), PQ (2 * sizes[9] complex),
 * pq (2 * sizes[8] floats), part (sizes[11] floats), mix_part (2B * sizes[10] floats). */
int dasp_reverb_backward(const float* x, const float* gy, const float* noise, const void* Fspec, const float* gains, const float* decays,
                         const float* mix, const void* Xf, const void* H, const float* z, float* gx, float* ggain, float* gdecay, float* gmix,
                         float* gpad, void* Gf, void* cspec, void* PQ, float* pq, float* part, float* mix_part, int B, long N, int L, int taps,
                         int nb, void* stream) {
    if (!x || !gy || !noise || !Fspec || !gains || !decays || !mix || !Xf || !H || !z || !gx || !ggain || !gdecay || !gmix || !gpad || !Gf ||
        !cspec || !PQ || !pq || !part || !mix_part || B <= 0 || N <= 0 || L <= 0 || taps <= 0 || nb <= 0 || nb > RV_BANDS_MAX)
        return DASP_ERR_ARG;
    const RvDims d = rv_dims(B, N, L, taps);
    hipStream_t st = (hipStream_t)stream;
    const long xrows = d.R * d.nblk;
    if (d.VQ < 1 || B > 65535 || xrows > 65535) return DASP_ERR_UNSUPPORTED;
    // blocks of mix * gy and their spectra
    hipLaunchKernelGGL(pad_rows_kernel, rv_grid(d.n1, xrows), dim3(RV_T), 0, st, gy, gpad, xrows, d.n1, 1, 0L, 0, N, d.Lb, d.nblk, mix);
    RV_TRY(rv_check());
    RV_TRY(fft_r2c(d.n1, xrows, gpad, (cpx*)Gf, st));
    // d/dir: cross-correlations with the input blocks (same block and previous block)
    cpx* P = (cpx*)PQ; cpx* Q = P + d.R * d.nfreq;
    hipLaunchKernelGGL(ir_grad_spec_kernel, rv_grid(d.nfreq, d.R), dim3(RV_T), 0, st, (const cpx*)Gf, (const cpx*)Xf, P, Q, d.nfreq, d.nblk);
    RV_TRY(rv_check());
    RV_TRY(fft_c2r(d.n1, 2 * d.R, P, pq, st));
    // d/dgain, d/ddecay: the filter bank again, weighted by g_ir = (p[n] + q[n + Lb]) / n1 (mix is already folded into G)
    hipLaunchKernelGGL(fb_fused_kernel<1>, dim3((unsigned)d.nwin, (unsigned)B), dim3(FFT_T), 0, st, noise, (const f2*)Fspec, gains, decays,
                       (float*)nullptr, pq, pq + d.R * d.n1, part, nb, L, taps, d.n1, d.Lb, d.VQ, 1.f / (float)d.n1);
    RV_TRY(rv_check());
    // d/dx: correlation with the impulse response
    hipLaunchKernelGGL(cmul_rows_kernel, rv_grid(d.nfreq, xrows), dim3(RV_T), 0, st, (const cpx*)Gf, (const cpx*)H, (cpx*)cspec, d.nfreq, d.nblk, (int)d.R, 1);
    RV_TRY(rv_check());
    RV_TRY(fft_c2r(d.n1, xrows, (cpx*)cspec, gpad, st));
    hipLaunchKernelGGL(bwd_combine_kernel, dim3(64, (unsigned)d.R), dim3(RV_T), 0, st, x, gy, z, gpad, mix, gx, mix_part, N, d.Lb, d.nblk, d.n1);
    RV_TRY(rv_check());
    const int nfin = B * nb > B ? B * nb : B;
    hipLaunchKernelGGL(reverb_finalize_kernel, dim3((nfin + 127) / 128), dim3(128), 0, st, part, mix_part, ggain, gdecay, gmix, B, nb, d.nwin, 64);
    return rv_check();
}

}  // extern "C"
