// Multi-resolution STFT loss (spectral convergence + log-magnitude L1 per resolution, mean over resolutions), forward and the
// gradients with respect to either signal (the first: the chain's output at every call site of the reference; the second - auraloss
// differentiates both - through the same kernels with the signals swapped, see grad_bin): the op directly downstream of the effect chain in the reference's training loops
// (auraloss.freq.MultiResolutionSTFTLoss(), call sites examples/style_transfer.py:341,363, auto_eq.py:252, virtual_analog.py:288;
// auraloss is not vendored in the reference: the algorithm of auraloss 0.4.0 with default arguments is restated in
// oracle/dasp_oracle.py:mrstft_loss, "parity unpinned").
//
// One fused kernel per direction and resolution: a 512-thread workgroup owns 4096 / n_fft frames of
// one signal row. The frames of the two signals are gathered (reflect padding, periodic Hann window zero-padded to n_fft) as ONE
// complex signal p + i t and transformed - frames of 512 / 1024 / 2048 points (the default resolutions) as 1 / 2 / 4 interleaved
// 512-point transforms, one per wave, with at most one workgroup barrier (split kernels, below); any other power of two with col_fft
// (fft_lds.hpp: frames side by side in registers + LDS) -, split into the two
// one-sided spectra through the mirrored bin, reduced to the three sums of a resolution
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

// one bin's contribution to the three sums of a resolution
__device__ __forceinline__ void loss_terms(const Bin& b, float eps, float& s1, float& s2, float& s3) {
    const float p2 = fmaxf(b.pr * b.pr + b.pi * b.pi, eps), t2 = fmaxf(b.tr * b.tr + b.ti * b.ti, eps);
    const float pm = __builtin_amdgcn_sqrtf(p2), tm = __builtin_amdgcn_sqrtf(t2);      // operands in [eps, ~1e6]: the plain instructions are exact enough (1 ulp)
    s1 = fmaf(tm - pm, tm - pm, s1);
    s2 += t2;
    s3 += 0.5f * fabsf(__logf(__fdividef(p2, t2)));       // |log pm - log tm| = |log(p2 / t2)| / 2; the ratio stays within 1e-14 .. 1e14
}
// one bin of the gradient spectrum: dL/d|P| * P / |P|
// k_self: the gradient w.r.t. the SECOND signal of the loss (the kernels are then called with the two signals swapped): the spectral
// convergence term is normalised by that signal's own norm, d/d|T| (s1 / s2) = (|T| - |P|) / (s1 s2) - s1 |T| / s2^3 - the first part is what
// the swapped call computes anyway, the second is k_self = - s1 / s2^3 times the spectrum itself; the log-magnitude term is symmetric.
__device__ __forceinline__ void grad_bin(const Bin& b, float eps, float k_sc, float k_lm, float k_self, float& hr, float& hi) {
    hr = 0.f; hi = 0.f;
    const float praw = b.pr * b.pr + b.pi * b.pi;
    if (praw > eps) {                                            // the clamp has zero slope below eps
        const float tm = sqrtf(fmaxf(b.tr * b.tr + b.ti * b.ti, eps)), pm = sqrtf(praw);
        const float sgn = tm > pm ? 1.f : (tm < pm ? -1.f : 0.f);            // sign(log tm - log pm)
        const float gm = (k_sc * (pm - tm) - k_lm * sgn / pm) / pm + k_self;  // dL/d|P| / |P|
        hr = gm * b.pr; hi = gm * b.pi;
    }
}

// gather + window + forward transform of this thread's 8 samples of its frame; afterwards r/i = Z[j + T q] and mr/mi = Z[F - (j + T q)]
// the window of a resolution, once per workgroup (all its frames share it): wlds[n] = hann_in_frame(n), n < F; a barrier follows in the caller's path
__device__ __forceinline__ void window_to_lds(float* wlds, const StftRes& R) {
    const int F = 1 << R.logF;
    for (int n = threadIdx.x; n < F; n += 512) wlds[n] = hann_in_frame(n, F, R.win);
    __syncthreads();
}

__device__ __forceinline__ void frames_to_spectra(const float* __restrict__ prow, const float* __restrict__ trow, int N, int frame, bool live,
                                                  const StftRes& R, const ColCfg& g, const f2* __restrict__ tw, f2* lds, const float* wlds,
                                                  float (&r)[8], float (&i)[8], float (&mr)[8], float (&mi)[8]) {
    const int F = 1 << R.logF;
    // all 16 loads first, at indices that are always valid (a load under `if (w != 0)` is a branch with its own wait: 8 exposed round
    // trips per thread); the window, zero outside its support and for frames past the end, is applied afterwards
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int s = reflect_index(frame * R.hop - F / 2 + g.j + g.T * q, N);
        s = s < 0 ? 0 : (s >= N ? N - 1 : s);
        r[q] = prow[s]; i[q] = trow[s];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float w = live ? wlds[g.j + g.T * q] : 0.f;
        r[q] *= w; i[q] *= w;
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

// One launch per resolution. These two kernels take any power of two 8 .. 4096 (col_fft with run-time geometry); frames of 512, 1024 and
// 2048 points - the default resolutions - run the split kernels further down.
// partials[((res * rows + row) * groups + group) * 3 + {0, 1, 2}]
__global__ void __launch_bounds__(512)
mrstft_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const f2* __restrict__ tw, float* __restrict__ partials,
                  StftSpec spec, int N, int res) {
    __shared__ f2 lds[ColGeom<12>::LDS];
    __shared__ float wlds[FFT_N];
    __shared__ float red[8][3];
    const StftRes R = spec.r[res];
    const ColCfg g = col_config<12>(R.logF, threadIdx.x);
    const int row = blockIdx.y, F = 1 << R.logF;
    if ((int)blockIdx.x * g.TC >= R.frames) return;          // uniform: this resolution has fewer frame groups than the grid
    const int frame = blockIdx.x * g.TC + g.c;
    const bool live = frame < R.frames;
    window_to_lds(wlds, R);
    float r[8], i[8], mr[8], mi[8];
    frames_to_spectra(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, R, g, tw, lds, wlds, r, i, mr, mi);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = g.j + g.T * q;
        if (live && k <= F / 2) {
            loss_terms(split_bin(r[q], i[q], mr[q], mi[q]), spec.eps, s1, s2, s3);
        }
    }
    s1 = wave_sum_uniform(s1); s2 = wave_sum_uniform(s2); s3 = wave_sum_uniform(s3);
    if (lane_id() == 0) { red[wave_id()][0] = s1; red[wave_id()][1] = s2; red[wave_id()][2] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = 0.f;
        for (int v = 0; v < 8; ++v) a += red[v][threadIdx.x];
        partials[(((size_t)res * gridDim.y + row) * spec.groups + blockIdx.x) * 3 + threadIdx.x] = a;
    }
}
// step 1, one workgroup per (resolution, sum): stats[res * 4 + c] = S_c, added up in fp64. 16 waves walk the rows, the lanes the
// frame groups of a row, four loads in flight per lane (one thread per element with an index division: 17 us; this: ~5)
__global__ void __launch_bounds__(1024)
mrstft_reduce_kernel(const float* __restrict__ partials, StftSpec spec, int rows, float* __restrict__ stats) {
    __shared__ double red[16];
    const int res = blockIdx.x / 3, c = blockIdx.x % 3, l = lane_id(), wv = wave_id();
    const int TC = FFT_N >> spec.r[res].logF, ng = (spec.r[res].frames + TC - 1) / TC;
    double s = 0.0;
    for (int row = wv; row < rows; row += 16) {
        const float* p = partials + ((size_t)res * rows + row) * spec.groups * 3 + c;
        for (int g0 = l; g0 < ng; g0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int gi = g0 + 64 * u; v[u] = p[(size_t)(gi < ng ? gi : ng - 1) * 3]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += g0 + 64 * u < ng ? (double)v[u] : 0.0;
        }
    }
    s = wave_sum(s);
    if (l == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int v = 0; v < 16; ++v) t += red[v];
        stats[res * 4 + c] = (float)t;
    }
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
                  const float* __restrict__ gloss, float* __restrict__ gpred, StftSpec spec, int N, int res, int wrt_second) {
    __shared__ f2 lds[ColGeom<12>::LDS];
    __shared__ float wlds[FFT_N];
    const StftRes R = spec.r[res];
    const ColCfg g = col_config<12>(R.logF, threadIdx.x);
    const int row = blockIdx.y, F = 1 << R.logF;
    if ((int)blockIdx.x * g.TC >= R.frames) return;
    const int frame = blockIdx.x * g.TC + g.c;
    const bool live = frame < R.frames;
    window_to_lds(wlds, R);
    float r[8], i[8], mr[8], mi[8];
    frames_to_spectra(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, R, g, tw, lds, wlds, r, i, mr, mi);
    const float s1 = stats[res * 4], s2 = stats[res * 4 + 1], count = stats[res * 4 + 2];
    const float gl = gloss[0] / (float)spec.nres;
    const float k_sc = s1 > 0.f ? gl / (s1 * s2) : 0.f, k_lm = gl / count, k_self = wrt_second ? -gl * s1 / (s2 * s2 * s2) : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = g.j + g.T * q;
        float hr = 0.f, hi = 0.f;
        if (live && k <= F / 2) grad_bin(split_bin(r[q], i[q], mr[q], mi[q]), spec.eps, k_sc, k_lm, k_self, hr, hi);
        r[q] = hr; i[q] = hi;
    }
    col_fft<1>(r, i, g, tw, lds);                                    // sum_k H[k] e^{+2 pi i k n / F}, H = 0 on the upper half
    float* grow = gpred + (size_t)row * N;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = g.j + g.T * q;
        const float w = hann_in_frame(n, F, R.win);     // measured: reading the LDS table here instead costs 7 % of the kernel
        if (live && w != 0.f) atomicAdd(grow + reflect_index(frame * R.hop - F / 2 + n, N), w * r[q]);
    }
}

// ---- n_fft = 512 R, R = 1, 2, 4: R waves per frame ------------------------------------------------------------------------------------
// A frame of F = 512 R points is R interleaved 512-point transforms (decimation in frequency): y_r'[m] = W_F^(m r') sum_rho x[m + 512 rho]
// W_R^(rho r'), X[r' + R kappa] = FFT512(y_r')[kappa]. Thread u of a frame (64 R threads, lanes along consecutive samples: coalesced gather
// and scatter) holds x[m + 512 rho] for 8 / R values of m, does the radix-R step in registers, and one transpose through LDS (the only
// workgroup barrier of the transform; none for R = 1) hands wave r' its y_r', which it transforms on its own (fft512_wave: wave-private
// exchanges). Afterwards lane l, register s of wave r' holds bin k = r' + R (l + 64 s). The one-sided bins k <= F / 2 are exactly the
// registers s < 4 of every wave (plus bin F / 2 in lane 0 of wave 0), so no lane idles through the logarithms, and the mirrored bin F - k
// sits in registers 7 - s of wave (R - r') % R, lane 63 - l (wave 0: lane 64 - l; its lane 0 keeps bin 0): lane permutes when that is the
// same wave (R = 1, 2), one more exchange of the upper registers through LDS for R = 4. 8 / R frames per workgroup, as the generic
// kernels group them (they remain for every other power of two).
template <int R> struct SplitCfg {
    int l, wv, c, rp, u;
    __device__ __forceinline__ SplitCfg() { l = lane_id(); wv = wave_id(); c = wv / R; rp = wv % R; u = rp * 64 + l; }
};

// forward: r/i <- this wave's bins, mr/mi <- the mirrored bins of registers 0..3
template <int R>
__device__ __forceinline__ void split_frame_spectrum(const float* __restrict__ prow, const float* __restrict__ trow, int N, int frame, bool live,
                                                     const StftRes& Rs, const SplitCfg<R>& g, const f2* __restrict__ tw, const Fft512Tw& t5,
                                                     f2* lds, const float* wlds, float (&r)[8], float (&i)[8], float (&mr)[4], float (&mi)[4]) {
    constexpr int F = 512 * R, G = 8 / R;
    f2* row = lds + g.wv * FFT512_LDS;
#pragma unroll
    for (int q = 0; q < 8; ++q) {              // register q = gi * R + rho  <->  sample n = (u + 64 R gi) + 512 rho
        const int n = g.u + 64 * R * (q / R) + 512 * (q % R);
        int s = reflect_index(frame * Rs.hop - F / 2 + n, N);
        s = s < 0 ? 0 : (s >= N ? N - 1 : s);
        r[q] = prow[s]; i[q] = trow[s];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float w = live ? wlds[g.u + 64 * R * (q / R) + 512 * (q % R)] : 0.f;
        r[q] *= w; i[q] *= w;
    }
    if constexpr (R > 1) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int m = g.u + 64 * R * gi;
            if constexpr (R == 2) {
                const float ar = r[2 * gi], ai = i[2 * gi];
                r[2 * gi] = ar + r[2 * gi + 1]; i[2 * gi] = ai + i[2 * gi + 1];
                r[2 * gi + 1] = ar - r[2 * gi + 1]; i[2 * gi + 1] = ai - i[2 * gi + 1];
            } else {
                dft4<-1>(r[4 * gi], i[4 * gi], r[4 * gi + 1], i[4 * gi + 1], r[4 * gi + 2], i[4 * gi + 2], r[4 * gi + 3], i[4 * gi + 3]);
            }
#pragma unroll
            for (int k = 1; k < R; ++k) {
                const f2 w = tw[m * k * (8 / R)];                       // W_F^(m k): m k < F, table step 4096 / F
                cmul_dir<-1>(r[R * gi + k], i[R * gi + k], w.x, w.y);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) lds[(g.c * R + k) * FFT512_LDS + m] = f2{r[R * gi + k], i[R * gi + k]};
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f2 v = row[g.l + 64 * q]; r[q] = v.x; i[q] = v.y; }
    }
    fft512_wave<-1>(r, i, g.l, t5, row);
    if constexpr (R == 4) {
        // the upper registers of every wave through LDS: wave r' reads those of wave (4 - r') % 4
        wave_lds_sync();
#pragma unroll
        for (int s = 4; s < 8; ++s) row[g.l + 64 * s] = f2{r[s], i[s]};
        __syncthreads();
        const f2* prow2 = lds + (g.c * R + ((R - g.rp) & (R - 1))) * FFT512_LDS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kap = g.l + 64 * s;
            const f2 v = prow2[g.rp == 0 ? ((512 - kap) & 511) : 511 - kap];
            const bool self = g.rp == 0 && kap == 0;                    // bin 0 mirrors onto itself (and was not published)
            mr[s] = self ? r[0] : v.x; mi[s] = self ? i[0] : v.y;
        }
        __syncthreads();                                                // the rows are reused (inverse transform / next exchange)
    } else {
        // same wave: wave 0 (and R = 1) pairs lane l with lane 64 - l, its lane 0 with its own registers; wave 1 of R = 2 pairs l with 63 - l
        const bool w0 = g.rp == 0;
        const int src = w0 ? ((64 - g.l) & 63) : 63 - g.l;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float zr = __shfl(r[7 - s], src), zi = __shfl(i[7 - s], src);
            const bool own = w0 && g.l == 0;
            mr[s] = own ? r[(8 - s) & 7] : zr;
            mi[s] = own ? i[(8 - s) & 7] : zi;
        }
    }
}

template <int R>
__global__ void __launch_bounds__(512)
mrstft_fwd_split_kernel(const float* __restrict__ pred, const float* __restrict__ target, const f2* __restrict__ tw, float* __restrict__ partials,
                        StftSpec spec, int N, int res) {
    __shared__ f2 lds[8 * FFT512_LDS];
    __shared__ float wlds[512 * R];
    __shared__ float red[8][3];
    const StftRes Rs = spec.r[res];
    const SplitCfg<R> g;
    const int row = blockIdx.y;
    const int frame = blockIdx.x * (8 / R) + g.c;
    const bool live = frame < Rs.frames;
    window_to_lds(wlds, Rs);
    const Fft512Tw t5 = fft512_twiddles(g.l, tw);
    float r[8], i[8], mr[4], mi[4];
    split_frame_spectrum<R>(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, Rs, g, tw, t5, lds, wlds, r, i, mr, mi);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
#pragma unroll
        for (int s = 0; s < 4; ++s) loss_terms(split_bin(r[s], i[s], mr[s], mi[s]), spec.eps, s1, s2, s3);
        if (g.rp == 0 && g.l == 0) loss_terms(split_bin(r[4], i[4], r[4], i[4]), spec.eps, s1, s2, s3);       // bin F / 2 mirrors onto itself
    }
    s1 = wave_sum_uniform(s1); s2 = wave_sum_uniform(s2); s3 = wave_sum_uniform(s3);
    if (g.l == 0) { red[g.wv][0] = s1; red[g.wv][1] = s2; red[g.wv][2] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = 0.f;
        for (int v = 0; v < 8; ++v) a += red[v][threadIdx.x];
        partials[(((size_t)res * gridDim.y + row) * spec.groups + blockIdx.x) * 3 + threadIdx.x] = a;
    }
}

template <int R>
__global__ void __launch_bounds__(512)
mrstft_bwd_split_kernel(const float* __restrict__ pred, const float* __restrict__ target, const f2* __restrict__ tw, const float* __restrict__ stats,
                        const float* __restrict__ gloss, float* __restrict__ gpred, StftSpec spec, int N, int res, int wrt_second) {
    constexpr int F = 512 * R, G = 8 / R;
    __shared__ f2 lds[8 * FFT512_LDS];
    __shared__ float wlds[512 * R];
    const StftRes Rs = spec.r[res];
    const SplitCfg<R> g;
    const int row = blockIdx.y;
    const int frame = blockIdx.x * (8 / R) + g.c;
    const bool live = frame < Rs.frames;
    window_to_lds(wlds, Rs);
    const Fft512Tw t5 = fft512_twiddles(g.l, tw);
    float r[8], i[8], mr[4], mi[4];
    f2* rowbuf = lds + g.wv * FFT512_LDS;
    split_frame_spectrum<R>(pred + (size_t)row * N, target + (size_t)row * N, N, frame, live, Rs, g, tw, t5, lds, wlds, r, i, mr, mi);
    const float s1 = stats[res * 4], s2 = stats[res * 4 + 1], count = stats[res * 4 + 2];
    const float gl = gloss[0] / (float)spec.nres;
    const float k_sc = s1 > 0.f ? gl / (s1 * s2) : 0.f, k_lm = gl / count, k_self = wrt_second ? -gl * s1 / (s2 * s2 * s2) : 0.f;
    float h4r = 0.f, h4i = 0.f;
    if (live && g.rp == 0 && g.l == 0) grad_bin(split_bin(r[4], i[4], r[4], i[4]), spec.eps, k_sc, k_lm, k_self, h4r, h4i);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float hr = 0.f, hi = 0.f;
        if (live) grad_bin(split_bin(r[s], i[s], mr[s], mi[s]), spec.eps, k_sc, k_lm, k_self, hr, hi);
        r[s] = hr; i[s] = hi;
    }
    r[4] = h4r; i[4] = h4i;
#pragma unroll
    for (int s = 5; s < 8; ++s) { r[s] = 0.f; i[s] = 0.f; }
    // x[n] = sum_k H[k] e^(+2 pi i k n / F), H = 0 above bin F / 2: the forward steps mirrored
    fft512_wave<1>(r, i, g.l, t5, rowbuf);
    if constexpr (R > 1) {
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 8; ++q) rowbuf[g.l + 64 * q] = f2{r[q], i[q]};
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int m = g.u + 64 * R * gi;
#pragma unroll
            for (int k = 0; k < R; ++k) { const f2 v = lds[(g.c * R + k) * FFT512_LDS + m]; r[R * gi + k] = v.x; i[R * gi + k] = v.y; }
#pragma unroll
            for (int k = 1; k < R; ++k) {
                const f2 w = tw[m * k * (8 / R)];
                cmul_dir<1>(r[R * gi + k], i[R * gi + k], w.x, w.y);
            }
            if constexpr (R == 2) {
                const float ar = r[2 * gi], ai = i[2 * gi];
                r[2 * gi] = ar + r[2 * gi + 1]; i[2 * gi] = ai + i[2 * gi + 1];
                r[2 * gi + 1] = ar - r[2 * gi + 1]; i[2 * gi + 1] = ai - i[2 * gi + 1];
            } else {
                dft4<1>(r[4 * gi], i[4 * gi], r[4 * gi + 1], i[4 * gi + 1], r[4 * gi + 2], i[4 * gi + 2], r[4 * gi + 3], i[4 * gi + 3]);
            }
        }
    }
    // (adding the workgroup's overlapping frames up in LDS first - float atomics on LDS, then one global atomic per sample of the span,
    // 2.5x fewer of them - was measured at +90 us per kernel: the global float atomics below are the cheaper ones)
    float* grow = gpred + (size_t)row * N;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = g.u + 64 * R * (q / R) + 512 * (q % R);
        const float w = hann_in_frame(n, F, Rs.win);
        if (live && w != 0.f) atomicAdd(grow + reflect_index(frame * Rs.hop - F / 2 + n, N), w * r[q]);
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
    for (int r = 0; r < nres; ++r) {
        const int TC = FFT_N >> s.r[r].logF;
        const dim3 grid((unsigned)((s.r[r].frames + TC - 1) / TC), (unsigned)rows);
        hipStream_t st = (hipStream_t)stream;
        switch (s.r[r].logF) {          // 512 / 1024 / 2048-point frames: 1 / 2 / 4 waves per frame; any other power of two: col_fft
            case 9: hipLaunchKernelGGL(mrstft_fwd_split_kernel<1>, grid, dim3(512), 0, st, pred, target, (const f2*)tw, partials, s, N, r); break;
            case 10: hipLaunchKernelGGL(mrstft_fwd_split_kernel<2>, grid, dim3(512), 0, st, pred, target, (const f2*)tw, partials, s, N, r); break;
            case 11: hipLaunchKernelGGL(mrstft_fwd_split_kernel<4>, grid, dim3(512), 0, st, pred, target, (const f2*)tw, partials, s, N, r); break;
            default: hipLaunchKernelGGL(mrstft_fwd_kernel, grid, dim3(512), 0, st, pred, target, (const f2*)tw, partials, s, N, r);
        }
    }
    hipLaunchKernelGGL(mrstft_reduce_kernel, dim3((unsigned)(nres * 3)), dim3(1024), 0, (hipStream_t)stream, (const float*)partials, s, rows, stats);
    hipLaunchKernelGGL(mrstft_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s, rows, stats, loss);
    return sl_check();
}
/* gpred (rows, N) is overwritten with gloss * d loss / d pred (gloss: device scalar); dasp_mrstft_backward_target: the same for the
 * second signal, gtarget = gloss * d loss / d target (auraloss differentiates both arguments: a consistency loss between two model
 * outputs needs it; the reference's call sites pass the reference signal there and never ask) */
static int mrstft_backward_impl(const float* first, const float* second, const void* tw, const float* stats, const float* gloss, float* gfirst,
                                int rows, int N, int nres, const int* fft, const int* hop, const int* win, float eps, int wrt_second, void* stream) {
    if (!first || !second || !tw || !stats || !gloss || !gfirst || rows <= 0 || N <= 0) return DASP_ERR_ARG;
    StftSpec s;
    if (!sl_spec(N, nres, fft, hop, win, eps, &s)) return DASP_ERR_UNSUPPORTED;
    if (rows > 65535) return DASP_ERR_UNSUPPORTED;
    if (zero_async(gfirst, (size_t)rows * N * sizeof(float), (hipStream_t)stream) != hipSuccess) return sl_check();
    for (int r = 0; r < nres; ++r) {
        const int TC = FFT_N >> s.r[r].logF;
        const dim3 grid((unsigned)((s.r[r].frames + TC - 1) / TC), (unsigned)rows);
        hipStream_t st = (hipStream_t)stream;
        switch (s.r[r].logF) {
            case 9: hipLaunchKernelGGL(mrstft_bwd_split_kernel<1>, grid, dim3(512), 0, st, first, second, (const f2*)tw, stats, gloss, gfirst, s, N, r, wrt_second); break;
            case 10: hipLaunchKernelGGL(mrstft_bwd_split_kernel<2>, grid, dim3(512), 0, st, first, second, (const f2*)tw, stats, gloss, gfirst, s, N, r, wrt_second); break;
            case 11: hipLaunchKernelGGL(mrstft_bwd_split_kernel<4>, grid, dim3(512), 0, st, first, second, (const f2*)tw, stats, gloss, gfirst, s, N, r, wrt_second); break;
            default: hipLaunchKernelGGL(mrstft_bwd_kernel, grid, dim3(512), 0, st, first, second, (const f2*)tw, stats, gloss, gfirst, s, N, r, wrt_second);
        }
    }
    return sl_check();
}
int dasp_mrstft_backward(const float* pred, const float* target, const void* tw, const float* stats, const float* gloss, float* gpred, int rows,
                         int N, int nres, const int* fft, const int* hop, const int* win, float eps, void* stream) {
    return mrstft_backward_impl(pred, target, tw, stats, gloss, gpred, rows, N, nres, fft, hop, win, eps, 0, stream);
}
int dasp_mrstft_backward_target(const float* pred, const float* target, const void* tw, const float* stats, const float* gloss, float* gtarget,
                                int rows, int N, int nres, const int* fft, const int* hop, const int* win, float eps, void* stream) {
    return mrstft_backward_impl(target, pred, tw, stats, gloss, gtarget, rows, N, nres, fft, hop, win, eps, 1, stream);
}

}  // extern "C"
