// Multi-resolution STFT loss (spectral convergence + log-magnitude L1 per resolution, mean over resolutions), forward and the
// gradient with respect to the first signal: the op directly downstream of the effect chain in the reference's training loops
// (auraloss.freq.MultiResolutionSTFTLoss(), call sites examples/style_transfer.py:341,363, auto_eq.py:252, virtual_analog.py:288;
// auraloss is not vendored in the reference: the algorithm of auraloss 0.4.0 with default arguments is restated in
// oracle/dasp_oracle.py:mrstft_loss, "parity unpinned").
//
// One fused kernel per direction, all resolutions in one launch (blockIdx.z): a 512-thread workgroup owns 4096 / n_fft frames of
// one signal row. The frames of the two signals are gathered (reflect padding, periodic Hann window zero-padded to n_fft) as ONE
// complex signal p + i t, transformed with col_fft (fft_lds.hpp: frames side by side in registers + LDS), split into the two
// one-sided spectra through one LDS read of the mirrored bin, reduced to the three sums of a resolution
//   S1 = sum (|T| - |P|)^2,  S2 = sum |T|^2,  S3 = sum |log|P| - log|T||       (|.| = sqrt(max(re^2 + im^2, eps)))
// per workgroup; a finalize kernel adds them in fp64:  loss = mean_r( sqrt(S1)/sqrt(S2) + S3 / count ).
// Backward recomputes the spectra, forms dL/d|P| * P/|P| on the one-sided bins, runs the inverse transform of that half spectrum and
// scatters window * Re(.) back through the frame overlap and the reflect padding with float atomics (each sample receives
// ~n_fft/hop contributions; the summation order, and only that, is not deterministic). Spectrograms never exist in HBM.
#include "common.hpp"
#include "fft_lds.hpp"

namespace dasp {

constexpr int SL_MAXRES = 8;
struct StftRes { int logF, hop, win, frames; };
struct StftSpec { StftRes r[SL_MAXRES]; int nres, groups; float eps; };

__global__ void stft_twiddle_kernel(f2* __restrict__ tw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < FFT_N) {
        double sn, cs;
        sincospi(2.0 * (double)e / (double)FFT_N, &sn, &cs);
        tw[e] = f2{(float)cs, (float)-sn};
    }
}

__device__ __forceinline__ int reflect_index(int s, int N) { return s < 0 ? -s : (s >= N ? 2 * N - 2 - s : s); }
// periodic Hann window of `win` samples centred in a frame of F samples (torch.stft zero-pads the window on both sides)
__device__ __forceinline__ float hann_in_frame(int n, int F, int win) {
    const int m = n - (F - win) / 2;
    return (m >= 0 && m < win) ? 0.5f - 0.5f * cospif(2.f * (float)m / (float)win) : 0.f;
}

// the two one-sided spectra of bin k from Z = FFT(p + i t): P = (Z[k] + conj Z[F-k]) / 2, T = (Z[k] - conj Z[F-k]) / (2i)
struct Bin { float pr, pi, tr, ti; };
__device__ __forceinline__ Bin split_bin(float zr, float zi, float mr, float mi) {
    return Bin{0.5f * (zr + mr), 0.5f * (zi - mi), 0.5f * (zi + mi), -0.5f * (zr - mr)};
}

// gather + window + forward transform of this thread's 8 samples of its frame; afterwards r/i = Z[j + T q] and mr/mi = Z[F - (j + T q)]
__device__ __forceinline__ void frames_to_spectra(const float* __restrict__ prow, const float* __restrict__ trow, int N, int frame, bool live,
                                                  const StftRes& R, const ColCfg& g, const f2* __restrict__ tw, f2* lds, float (&r)[8],
                                                  float (&i)[8], float (&mr)[8], float (&mi)[8]) {
    const int F = 1 << R.logF;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = g.j + g.T * q;
        const float w = hann_in_frame(n, F, R.win);
        r[q] = 0.f; i[q] = 0.f;
        if (live && w != 0.f) {
            const int s = reflect_index(frame * R.hop - F / 2 + n, N);
            r[q] = w * prow[s]; i[q] = w * trow[s];
        }
    }
    col_fft<-1>(r, i, g, tw, lds);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[fft_pad(g.j + g.T * q) * g.TC + g.c] = f2{r[q], i[q]};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f2 m = lds[fft_pad((F - (g.j + g.T * q)) & (F - 1)) * g.TC + g.c];
        mr[q] = m.x; mi[q] = m.y;
    }
    __syncthreads();       // the caller may reuse lds for another transform
}

// partials[((res * rows + row) * groups + group) * 3 + {0, 1, 2}]
__global__ void __launch_bounds__(512)
mrstft_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const f2* __restrict__ tw, float* __restrict__ partials,
                  StftSpec spec, int N) {
    __shared__ f2 lds[ColGeom<12>::LDS];
    __shared__ float red[8][3];
    const StftRes R = spec.r[blockIdx.z];
    const ColCfg g = col_config<12>(R.logF, threadIdx.x);
    const int row = blockIdx.y, F = 1 << R.logF;
    if ((int)blockIdx.x * g.TC >= R.frames) return;          // uniform: this resolution has fewer frame groups than the grid
    const int frame = blockIdx.x * g.TC + g.c;
    const bool live = frame < R.frames;
    float r[8], i[8], mr[8], mi[8];
    frames_to_spectra(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, R, g, tw, lds, r, i, mr, mi);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = g.j + g.T * q;
        if (live && k <= F / 2) {
            const Bin b = split_bin(r[q], i[q], mr[q], mi[q]);
            const float p2 = fmaxf(b.pr * b.pr + b.pi * b.pi, spec.eps), t2 = fmaxf(b.tr * b.tr + b.ti * b.ti, spec.eps);
            const float pm = sqrtf(p2), tm = sqrtf(t2);
            s1 = fmaf(tm - pm, tm - pm, s1);
            s2 += t2;
            s3 += 0.5f * fabsf(logf(p2) - logf(t2));       // |log pm - log tm|
        }
    }
    s1 = wave_sum_uniform(s1); s2 = wave_sum_uniform(s2); s3 = wave_sum_uniform(s3);
    if (lane_id() == 0) { red[wave_id()][0] = s1; red[wave_id()][1] = s2; red[wave_id()][2] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = 0.f;
        for (int v = 0; v < 8; ++v) a += red[v][threadIdx.x];
        partials[(((size_t)blockIdx.z * gridDim.y + row) * spec.groups + blockIdx.x) * 3 + threadIdx.x] = a;
    }
}

// step 1, one workgroup per (resolution, sum): stats[res * 4 + c] = S_c, added up in fp64
__global__ void __launch_bounds__(256)
mrstft_reduce_kernel(const float* __restrict__ partials, StftSpec spec, int rows, float* __restrict__ stats) {
    __shared__ double red[4];
    const int res = blockIdx.x / 3, c = blockIdx.x % 3;
    const int TC = FFT_N >> spec.r[res].logF, ng = (spec.r[res].frames + TC - 1) / TC;
    double s = 0.0;
    for (long e = threadIdx.x; e < (long)rows * ng; e += 256) {
        const long row = e / ng, gi = e % ng;
        s += (double)partials[(((size_t)res * rows + row) * spec.groups + gi) * 3 + c];
    }
    s = wave_sum(s);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) stats[res * 4 + c] = (float)(red[0] + red[1] + red[2] + red[3]);
}
// step 2: stats[res] = (sqrt S1, sqrt S2, count, S3); loss[0] = mean over resolutions of sqrt(S1)/sqrt(S2) + S3/count
__global__ void mrstft_finalize_kernel(StftSpec spec, int rows, float* __restrict__ stats, float* __restrict__ loss) {
    if (threadIdx.x != 0) return;
    double total = 0.0;
    for (int res = 0; res < spec.nres; ++res) {
        const double F = (double)(1 << spec.r[res].logF), count = (double)rows * spec.r[res].frames * (F / 2 + 1);
        const double s1 = sqrt((double)stats[res * 4 + 0]), s2 = sqrt((double)stats[res * 4 + 1]), s3 = (double)stats[res * 4 + 2];
        stats[res * 4 + 0] = (float)s1; stats[res * 4 + 1] = (float)s2; stats[res * 4 + 2] = (float)count; stats[res * 4 + 3] = (float)s3;
        total += s1 / s2 + s3 / count;
    }
    loss[0] = (float)(total / spec.nres);
}

// gpred (rows, N) must be zero on entry; gloss = d(objective)/d(loss), a device scalar
__global__ void __launch_bounds__(512)
mrstft_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const f2* __restrict__ tw, const float* __restrict__ stats,
                  const float* __restrict__ gloss, float* __restrict__ gpred, StftSpec spec, int N) {
    __shared__ f2 lds[ColGeom<12>::LDS];
    const StftRes R = spec.r[blockIdx.z];
    const ColCfg g = col_config<12>(R.logF, threadIdx.x);
    const int row = blockIdx.y, F = 1 << R.logF;
    if ((int)blockIdx.x * g.TC >= R.frames) return;
    const int frame = blockIdx.x * g.TC + g.c;
    const bool live = frame < R.frames;
    float r[8], i[8], mr[8], mi[8];
    frames_to_spectra(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, R, g, tw, lds, r, i, mr, mi);
    const float s1 = stats[blockIdx.z * 4], s2 = stats[blockIdx.z * 4 + 1], count = stats[blockIdx.z * 4 + 2];
    const float gl = gloss[0] / (float)spec.nres;
    const float k_sc = s1 > 0.f ? gl / (s1 * s2) : 0.f, k_lm = gl / count;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = g.j + g.T * q;
        float hr = 0.f, hi = 0.f;
        if (live && k <= F / 2) {
            const Bin b = split_bin(r[q], i[q], mr[q], mi[q]);
            const float praw = b.pr * b.pr + b.pi * b.pi;
            if (praw > spec.eps) {                                   // the clamp has zero slope below eps
                const float tm = sqrtf(fmaxf(b.tr * b.tr + b.ti * b.ti, spec.eps)), pm = sqrtf(praw);
                const float sgn = tm > pm ? 1.f : (tm < pm ? -1.f : 0.f);        // sign(log tm - log pm)
                const float gm = (k_sc * (pm - tm) - k_lm * sgn / pm) / pm;       // dL/d|P| / |P|
                hr = gm * b.pr; hi = gm * b.pi;
            }
        }
        r[q] = hr; i[q] = hi;
    }
    col_fft<1>(r, i, g, tw, lds);                                    // sum_k H[k] e^{+2 pi i k n / F}, H = 0 on the upper half
    float* grow = gpred + (size_t)row * N;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = g.j + g.T * q;
        const float w = hann_in_frame(n, F, R.win);
        if (live && w != 0.f) atomicAdd(grow + reflect_index(frame * R.hop - F / 2 + n, N), w * r[q]);
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
inline int sl_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
bool sl_spec(int N, int nres, const int* fft, const int* hop, const int* win, float eps, StftSpec* out) {
    if (nres <= 0 || nres > SL_MAXRES || !fft || !hop || !win) return false;
    StftSpec s = {};
    s.nres = nres; s.eps = eps; s.groups = 0;
    for (int r = 0; r < nres; ++r) {
        int lg = 0;
        while ((1 << lg) < fft[r]) ++lg;
        if ((1 << lg) != fft[r] || lg < 3 || lg > 12 || hop[r] <= 0 || win[r] <= 0 || win[r] > fft[r] || fft[r] / 2 >= N) return false;
        s.r[r] = StftRes{lg, hop[r], win[r], 1 + N / hop[r]};
        const int TC = FFT_N >> lg, ng = (s.r[r].frames + TC - 1) / TC;
        if (ng > s.groups) s.groups = ng;
    }
    *out = s;
    return true;
}
}  // namespace

extern "C" {

/* floats of `partials` for rows signals of N samples; -1 if the resolutions are not supported (n_fft a power of two in 8..4096,
 * win <= n_fft, n_fft / 2 < N, at most 8 resolutions) */
long dasp_mrstft_partial_floats(long rows, int N, int nres, const int* fft, const int* hop, const int* win) {
    StftSpec s;
    if (!sl_spec(N, nres, fft, hop, win, 0.f, &s)) return -1;
    return (long)nres * rows * s.groups * 3;
}
/* tw: 4096 complex (8192 floats), the twiddle table the transforms read */
int dasp_mrstft_table(void* tw, void* stream) {
    if (!tw) return DASP_ERR_ARG;
    hipLaunchKernelGGL(stft_twiddle_kernel, dim3(FFT_N / 256), dim3(256), 0, (hipStream_t)stream, (f2*)tw);
    return sl_check();
}
/* pred, target (rows, N); stats (4 * nres floats, kept for the backward); loss: 1 float */
int dasp_mrstft_forward(const float* pred, const float* target, const void* tw, float* partials, float* stats, float* loss, int rows, int N,
                        int nres, const int* fft, const int* hop, const int* win, float eps, void* stream) {
    if (!pred || !target || !tw || !partials || !stats || !loss || rows <= 0 || N <= 0) return DASP_ERR_ARG;
    StftSpec s;
    if (!sl_spec(N, nres, fft, hop, win, eps, &s)) return DASP_ERR_UNSUPPORTED;
    if (rows > 65535) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mrstft_fwd_kernel, dim3((unsigned)s.groups, (unsigned)rows, (unsigned)nres), dim3(512), 0, (hipStream_t)stream, pred, target,
                       (const f2*)tw, partials, s, N);
    hipLaunchKernelGGL(mrstft_reduce_kernel, dim3((unsigned)(nres * 3)), dim3(256), 0, (hipStream_t)stream, (const float*)partials, s, rows, stats);
    hipLaunchKernelGGL(mrstft_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s, rows, stats, loss);
    return sl_check();
}
/* gpred (rows, N) is overwritten with gloss * d loss / d pred (gloss: device scalar) */
int dasp_mrstft_backward(const float* pred, const float* target, const void* tw, const float* stats, const float* gloss, float* gpred, int rows,
                         int N, int nres, const int* fft, const int* hop, const int* win, float eps, void* stream) {
    if (!pred || !target || !tw || !stats || !gloss || !gpred || rows <= 0 || N <= 0) return DASP_ERR_ARG;
    StftSpec s;
    if (!sl_spec(N, nres, fft, hop, win, eps, &s)) return DASP_ERR_UNSUPPORTED;
    if (rows > 65535) return DASP_ERR_UNSUPPORTED;
    if (hipMemsetAsync(gpred, 0, (size_t)rows * N * sizeof(float), (hipStream_t)stream) != hipSuccess) return sl_check();
    hipLaunchKernelGGL(mrstft_bwd_kernel, dim3((unsigned)s.groups, (unsigned)rows, (unsigned)nres), dim3(512), 0, (hipStream_t)stream, pred, target,
                       (const f2*)tw, stats, gloss, gpred, s, N);
    return sl_check();
}

}  // extern "C"
