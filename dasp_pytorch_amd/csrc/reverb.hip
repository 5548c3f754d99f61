// Noise-shaped reverberation: shaped-noise impulse response + long convolution, forward and adjoint, entirely on the
// library's own register/LDS FFTs (fft_lds.hpp) - no FFT library, no real-to-complex post-processing passes.
//
// Replaces dasp_pytorch.functional.noise_shaped_reverberation (dasp_pytorch/functional.py:406-577):
//   wn_filt = grouped 1023-tap FIR of white noise, 12 bands          (:548-558, direct conv1d)
//   ir      = mean_band(wn_filt * exp(-(10 decay + 1) t) * gain)     (:561-567)
//   y_wet   = causal convolution of x with ir, truncated to N        (:570-572, direct conv1d, 65536 taps)
//   y       = (1 - mix) x + mix y_wet                                (:575)
// The reference's direct convolutions are 99.6 % of its time. Here both are frequency-domain products.
//
// Filter bank (short filters, 2B*12 independent noise rows): ONE fused kernel - overlap-save windows of 4096 samples,
// the left/right rows of an item packed as one complex signal, forward FFT and product with the band's spectrum inside
// the workgroup; the decay envelope is folded into the transform's inputs, so the band sum happens in the frequency
// domain and a window costs 12 forward transforms + 1 inverse (fb_fused_kernel). The filtered noise never exists in
// HBM; the backward pass re-runs the forward transforms and takes its sums over frequency (Parseval) against the
// transform of d loss / d ir - no inverse transform at all.
//
// Long convolution (L taps, L up to 2^20): overlap-add over blocks of Lb = nextpow2(L) samples, n1 = 2 Lb-point complex
// transforms of PAIRS of consecutive blocks of one signal (block 2p real, block 2p+1 imaginary; both meet the same real
// impulse response, so one complex product serves both and no Hermitian split is ever needed). Each transform is a
// four-step FFT n1 = NA x 512 (time index = ja * 512 + jb, frequency index = ka + NA * kb) in three kernels:
//   conv_load_kernel   column transforms over ja (col_fft), input gathered from the signal        -> A[ka][jb]
//   conv_rows_kernel   per row ka (one wave each): twiddle, 512-point transform, product with the stored spectrum,
//                      inverse transform, conjugate twiddle - spectra stay in the permuted [ka][kb] order, which
//                      pointwise products do not care about, so there is no transpose anywhere
//   conv_cols_kernel   inverse column transforms, fused with overlap-add + wet/dry mix (forward; the workgroup walks
//                      the pairs of its signal and carries the overlap in registers), or with the gx / mix-gradient
//                      epilogue, or with the extraction of d loss / d ir (backward)
// Backward uses overlapped (not zero-padded) windows of mix * gy: gx_k = first half of IFFT(GW_k conj(H)) and
// d ir = first half of IFFT(sum_k GW_k conj(X_k)), with X_k recomputed from the saved column transforms A.
#include "common.hpp"
#include "fft_lds.hpp"

namespace dasp {

constexpr int RV_BANDS_MAX = 16;
constexpr int CV_NB = 512;                // row length of the four-step split
constexpr int LOAD_LOG = 12, COLS_LOG = 13;   // workgroup size (log2 elements) of the forward / inverse column kernels: measured per kernel
typedef ColGeom<LOAD_LOG> LoadGeom;
typedef ColGeom<COLS_LOG> ColsGeom;

// Workgroups are dealt round-robin to the 8 XCDs in launch order: physical index bx of nx -> logical index, a contiguous run per XCD
// (identity when nx is not a multiple of 8). Used where neighbouring logical workgroups share cache lines: neighbouring column tiles of the
// four-step kernels touch neighbouring (at small NA: the same) 128-byte lines of the signal; the windows of an item share its band spectra.
__device__ __forceinline__ int xcd_tile(int bx, int nx) { return (nx & 7) ? bx : (bx & 7) * (nx >> 3) + (bx >> 3); }

// ---- counter-based white noise (device_noise): the reference's torch.randn(bs * 2, 12, num_samples + taps - 1) (functional.py:548) ---------
// generated where it is consumed. Written to memory by torch.randn and read by the forward and the backward filter bank it was 0.82 GB of
// HBM traffic three times over per step; the stream below is a pure function of (seed, batch item, band, sample index), so both kernels -
// and every overlapping window inside them - recompute the same values. One 32-bit hash per (item, band, index) gives BOTH noise rows of the
// item (rows 2b and 2b+1, which enter the transform as one complex signal anyway) as a Box-Muller pair:
//   (a, c) = stream key of (seed, b * nb + band): 24-bit odd multiplier, 32-bit offset       h = lowbias32(m * a + c)   (m < 2^24)
//   u1 = ((h >> 16) + 0.5) / 65536, u2 = (h & 0xffff) / 65536;   r = sqrt(-2 ln u1);   (re, im) = r (cos, sin)(2 pi u2)
// lowbias32 = Wellons' two-round multiply-xorshift finaliser. Different streams are different affine index maps into the hash: they share
// at most isolated values, never a run. Executable specification and its statistics: oracle/noise_stream.py, tests/test_oracle_cpu.py.
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct NoiseKey { unsigned a, c; };
__device__ __forceinline__ NoiseKey noise_key(unsigned long long seed, unsigned sid) {
    const unsigned k0 = lowbias32(sid + 0x9E3779B9U);
    const unsigned k1 = lowbias32(k0 ^ (unsigned)seed);
    const unsigned k2 = lowbias32(k1 ^ (unsigned)(seed >> 32) ^ 0x85EBCA6BU);
    return NoiseKey{(k1 & 0xFFFFFFU) | 1U, k2};
}
// Developer bounds for the round-6 filter-bank decision (profiles/r06/fb_bounds.log; timing only - the results are meaningless):
//   -DDASP_FB_BOUND=1  the noise costs nothing (no hash, no Box-Muller): what ANY cheaper Gaussian could save at most
//   -DDASP_FB_BOUND=2  ... and the 12 forward transforms per window are skipped: what synthesising the bands in the frequency domain
//                      could save at most (it would still have to generate one complex Gaussian per bin - not counted here)
#ifndef DASP_FB_BOUND
#define DASP_FB_BOUND 0
#endif
__device__ __forceinline__ void noise_pair(NoiseKey k, unsigned m, float& re, float& im) {
#if DASP_FB_BOUND
    re = __builtin_bit_cast(float, 0x3f800000u | (m & 0x7fffffu)) - 1.5f; im = re * 0.5f + __builtin_bit_cast(float, 0x3f000000u | (k.c & 0xffffu));
    return;
#endif
    const unsigned h = lowbias32(__umul24(m, k.a) + k.c);
    const float u1 = ((float)(h >> 16) + 0.5f) * (1.f / 65536.f);
    const float t = (float)(h & 0xFFFFU) * (1.f / 65536.f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // v_log_f32 is log2: -2 ln 2 log2(u1)
    re = r * __builtin_amdgcn_cosf(t);                                                             // v_cos / v_sin take revolutions
    im = r * __builtin_amdgcn_sinf(t);
}
// The same stream written out in the reference's layout (2B, nb, row_len): the test hook that lets the explicit-noise path and the oracle
// see what the kernels generate (dasp_reverb_noise).
__global__ void reverb_noise_kernel(unsigned long long seed, const unsigned long long* __restrict__ seed_dev, float* __restrict__ out, int nb, int row_len) {
    if (seed_dev) seed += *seed_dev;
    const int sid = blockIdx.y;                      // b * nb + band
    const int b = sid / nb, band = sid % nb;
    const NoiseKey key = noise_key(seed, (unsigned)sid);
    float* rl = out + ((long)(2 * b) * nb + band) * row_len;
    float* rr = out + ((long)(2 * b + 1) * nb + band) * row_len;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < row_len; m += gridDim.x * blockDim.x) {
        float re, im;
        noise_pair(key, (unsigned)m, re, im);
        rl[m] = re; rr[m] = im;
    }
}

// ---- fused filter bank ---------------------------------------------------------------------------------------------
// spec layout (complex): [0, 4096) forward twiddles exp(-2 pi i e / 4096); then nb rows of 4096: conj(FFT(filter_band)) / 4096
// in the split order of fft4096_split_fwd; then the taps themselves, nb * taps floats (fb_wspectrum_kernel weights them per item).
__global__ void fb_twiddle_kernel(f2* __restrict__ spec) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < FFT_N) {
        double sn, cs;
        sincospi(2.0 * (double)e / (double)FFT_N, &sn, &cs);
        spec[e] = f2{(float)cs, (float)-sn};
    }
}
__global__ __launch_bounds__(FFT_T) void fb_spectrum_kernel(const float* __restrict__ filters, f2* __restrict__ spec, int taps) {
    __shared__ f2 lds[2 * FFT_LDS];
    const int j = threadIdx.x, band = blockIdx.x;
    float r[8], i[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int idx = j + 512 * q; r[q] = idx < taps ? filters[(long)band * taps + idx] : 0.f; i[q] = 0.f; }
    float* keep = reinterpret_cast<float*>(spec + (long)(gridDim.x + 1) * FFT_N) + (long)band * taps;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int idx = j + 512 * q; if (idx < taps) keep[idx] = r[q]; }
    const SplitTw tw = split_twiddles(j, spec);
    fft4096_split_fwd(r, i, j, tw, lds);
    constexpr float inv = 1.f / (float)FFT_N;
    const int so = (j >> 6) * 512 + (j & 63);          // split order (fft_lds.hpp), the order fb_fused_kernel multiplies in
#pragma unroll
    for (int q = 0; q < 8; ++q) spec[FFT_N + (long)band * FFT_N + so + 64 * q] = f2{r[q] * inv, -i[q] * inv};
}

// The envelope inside the transform. The decay exp(-d t_n), d = 10 decay + 1, splits over a window starting at n0 as
//   exp(-d t_n0) exp(-rho idx),   rho = d / (L - 1),   and   exp(-rho idx) sum_k f[k] z[idx + k] = sum_k (f[k] exp(rho k)) (z[idx + k] exp(-rho (idx + k))):
// weighting the noise by exp(-rho m) on the way in and the band filter by exp(+rho k) makes the transform deliver the ENVELOPED band
// output, so the sum over the bands is taken in the frequency domain and a window needs 12 forward transforms + ONE inverse instead of
// 12 + 12 (forward), or 12 + 2 forward transforms of the weight and no inverse at all (backward: the sums over n become sums over
// frequency by Parseval's identity). The weighted band spectra depend on the item's decays: fb_wspectrum_kernel computes them per
// (item, band) at the start of every call (1 transform per 22 windows' worth of noise transforms).
// The weights span exp(|rho| 4096) inside a window, which the fp32 transform's rounding floor (relative to the largest element) does
// not see: the route is taken while |rho| 4096 <= RV_WEIGHT_LIMIT (a factor e^2 = 7.4; the default L = 65536 with decay in [0, 1]
// has |rho| 4096 <= 0.69), decided per batch item inside the kernel; short impulse responses with fast decays take the per-band route.
constexpr float RV_WEIGHT_LIMIT = 2.f;

// grid (nb, B): wspec[(b nb + band)] = conj(FFT(f_band[k] exp(rho k))) / 4096 in split order; taps_f (nb, taps) raw filter taps
__global__ __launch_bounds__(FFT_T) void fb_wspectrum_kernel(const f2* __restrict__ spec, const float* __restrict__ taps_f, const float* __restrict__ decays,
                                                             f2* __restrict__ wspec, int nb, int taps, float tstep) {
    __shared__ f2 lds[FFT_LDS];
    const int j = threadIdx.x, band = blockIdx.x, b = blockIdx.y;
    const float rho = (10.f * decays[b * nb + band] + 1.f) * tstep;
    float r[8], i[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {           // loads first (clamped index), weights afterwards: see fb_fused_kernel
        const int idx = j + 512 * q;
        r[q] = taps_f[(long)band * taps + (idx < taps ? idx : taps - 1)];
        i[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = j + 512 * q;
        r[q] = idx < taps ? r[q] * expf(fminf(rho * (float)idx, 80.f)) : 0.f;      // the clamp only matters off the weighted route
    }
    const SplitTw tw = split_twiddles(j, spec);
    fft4096_split_fwd(r, i, j, tw, lds);
    constexpr float inv = 1.f / (float)FFT_N;
    const int so = (j >> 6) * 512 + (j & 63);
    f2* out = wspec + ((long)b * nb + band) * FFT_N + so;
#pragma unroll
    for (int q = 0; q < 8; ++q) out[64 * q] = f2{r[q] * inv, -i[q] * inv};
}

// One workgroup = one (batch item b, window w): output samples n = w V + idx, idx < V = 512 VQ <= 4096 - (taps - 1).
//   z_band[idx] = noise[b,0,band][n0 + idx] + i noise[b,1,band][n0 + idx]          (functional.py:548; both rows share the band filter)
//   o_band     = IFFT(FFT(z_band) conj(F_band))  -> valid cross-correlations for idx < V   (:551-558)
//   MODE 0:  ir[b,c][n] = 1/nb sum_band gain env_band(t_n) o_band                   (:561-567)
//   MODE 1:  part[(b, w), band] = (sum_n gir o env / nb,  sum_n gir o env gain (-10 t_n) / nb),  gir (2B, L) = d loss / d ir
// computed with the envelope inside the transform (ROUTE 1, above) or, for |rho| beyond the limit, band by band in the time domain
// (ROUTE 0). Both instantiations are launched over the same grid and a workgroup returns at once when its item belongs to the other
// route (one kernel holding both loops spills ~140 registers at the 128 it may use; the empty workgroups cost a few microseconds).
// Very few batch items: a workgroup's loop over the bands is its whole run time, and B * windows workgroups may not fill
// the chip (1 item: 22 workgroups). bsplit > 1 deals the bands out to gridDim.z workgroups per (item, window): MODE 0 then adds its
// bands' share into ir with float atomics (ir zeroed by the caller; the order of the additions is not deterministic), MODE 1 writes the
// partial sums of its own bands only.
// GEN: the noise is generated (stream above, `seed`) instead of read from `noise`.
template <int MODE, int ROUTE, bool GEN>
__global__ __launch_bounds__(FFT_T, 4) void fb_fused_kernel(const float* __restrict__ noise, const f2* __restrict__ spec, const f2* __restrict__ wspec,
                                                         const float* __restrict__ gains, const float* __restrict__ decays, float* __restrict__ ir,
                                                         const float* __restrict__ gir, float* __restrict__ part, int nb, int L, int taps, int VQ, float limit,
                                                         unsigned long long seed, const unsigned long long* __restrict__ seed_dev, int force) {
    if (GEN && seed_dev) seed += *seed_dev;      // per-replay offset of a captured launch (the by-value seed is frozen at capture time)
    const int bsplit = gridDim.z, bper = (nb + bsplit - 1) / bsplit, band_lo = blockIdx.z * bper, band_hi = band_lo + bper < nb ? band_lo + bper : nb;
    __shared__ f2 lds[2 * FFT_LDS];
    __shared__ float red[FFT_T / 64][RV_BANDS_MAX][2];
    // Workgroups go round-robin to the 8 XCDs in launch order. All windows of an item read the same 12 weighted band spectra (384 KB,
    // private to the item): each XCD takes a contiguous run of items so that they meet in one L2 instead of being fetched by all eight.
    const int lin = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int j = threadIdx.x, w = lin % (int)gridDim.x, b = lin / (int)gridDim.x;
    const int V = VQ * 512, n0 = w * V, row_len = L + taps - 1;
    const float tstep = L > 1 ? 1.f / (float)(L - 1) : 0.f, inv_nb = 1.f / (float)nb;
    const int so = (j >> 6) * 512 + (j & 63);
    const float t0 = (float)(n0 + j) * tstep, t512 = 512.f * tstep;      // t_n = n / (L - 1), torch.linspace(0, 1, L)
    const float tw0 = (float)n0 * tstep;                                 // time of the window's first sample
    float dmax = 0.f;
    for (int band = 0; band < nb; ++band) dmax = fmaxf(dmax, fabsf(10.f * decays[b * nb + band] + 1.f));
    if (!force && (dmax * tstep * (float)FFT_N <= limit) != (ROUTE == 1)) return;        // uniform over the workgroup (and over the item's workgroups)
    // (force: the caller vouches that every item belongs to this route - the other route's launch, empty but not free, is not issued)
    float accr[8], acci[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = n0 + j + 512 * q;
        accr[q] = 0.f; acci[q] = 0.f;
        if (MODE == 1 && q < VQ && n < L) {          // weights: gradient w.r.t. the two impulse responses of this item
            accr[q] = gir[(long)(2 * b) * L + n];
            acci[q] = gir[(long)(2 * b + 1) * L + n];
        }
    }
    if constexpr (ROUTE == 1) {
        // The band loop keeps spectra in registers across its transforms (MODE 0: the running sum over the bands; MODE 1: two weight
        // spectra): the transform variant with 6 twiddle registers (fft_lds.hpp) leaves room to have a band's 16 noise loads and 8
        // spectrum loads in flight together. The one inverse transform of MODE 0 builds its twiddles after the loop.
        const SplitTwLean tw = split_twiddles_lean(j, spec);
        auto fwd = [&](float (&rr_)[8], float (&ii_)[8], f2* buf) { if (DASP_FB_BOUND != 2) fft4096_split_fwd_lean(rr_, ii_, j, tw, buf); };
        float g2r[8], g2i[8];
        if (MODE == 1) {                             // G1 = FFT(gir), G2 = FFT(window time * gir); zero outside the valid range
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float tq = (float)(j + 512 * q) * tstep;
                g2r[q] = accr[q] * tq; g2i[q] = acci[q] * tq;
            }
            fwd(accr, acci, lds);
            fwd(g2r, g2i, lds + FFT_LDS);
        }
        int flip = 0;
        for (int band = band_lo; band < band_hi; ++band, flip ^= 1) {
            const float g = gains[b * nb + band], d = 10.f * decays[b * nb + band] + 1.f, rho = d * tstep;
            float r[8], i[8];
            {
                if (GEN) {
                    const NoiseKey key = noise_key(seed, (unsigned)(b * nb + band));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int idx = n0 + j + 512 * q;
                        float vl, vr;
                        noise_pair(key, (unsigned)idx, vl, vr);
                        r[q] = idx < row_len ? vl : 0.f;
                        i[q] = idx < row_len ? vr : 0.f;
                    }
                } else {
                    const float* rl = noise + ((long)(2 * b) * nb + band) * row_len;
                    const float* rr = noise + ((long)(2 * b + 1) * nb + band) * row_len;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {       // all 16 loads in flight before the first use (a product inside the select makes each a branch + wait)
                        const int idx = n0 + j + 512 * q, ic = idx < row_len ? idx : row_len - 1;      // clamped address: a plain load, no branch
                        const float vl = rl[ic], vr = rr[ic];
                        r[q] = idx < row_len ? vl : 0.f;
                        i[q] = idx < row_len ? vr : 0.f;
                    }
                }
                float wq = __expf(-rho * (float)j);
                const float ws = __expf(-rho * 512.f);
#pragma unroll
                for (int q = 0; q < 8; ++q) { r[q] *= wq; i[q] *= wq; wq *= ws; }
            }
            const f2* F = wspec + ((long)b * nb + band) * FFT_N + so;
            fwd(r, i, lds + flip * FFT_LDS);
            const float e0 = __expf(-d * tw0) * inv_nb;
            if (MODE == 0) {
                const float c = g * e0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f2 f = F[64 * q];
                    accr[q] = fmaf(c, r[q] * f.x - i[q] * f.y, accr[q]);
                    acci[q] = fmaf(c, r[q] * f.y + i[q] * f.x, acci[q]);
                }
            } else {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f2 f = F[64 * q];
                    const float pr = r[q] * f.x - i[q] * f.y, pi = r[q] * f.y + i[q] * f.x;
                    s1 = fmaf(accr[q], pr, fmaf(acci[q], pi, s1));
                    s2 = fmaf(g2r[q], pr, fmaf(g2i[q], pi, s2));
                }
                s1 = wave_sum_uniform(s1); s2 = wave_sum_uniform(s2);
                if (lane_id() == 0) { red[wave_id()][band][0] = e0 * s1; red[wave_id()][band][1] = -10.f * g * e0 * fmaf(tw0, s1, s2); }
            }
        }
        if constexpr (MODE == 0) fft4096_split_inv(accr, acci, j, split_twiddles(j, spec), lds + flip * FFT_LDS);
    } else {
        const SplitTw tw = split_twiddles(j, spec);
        float sumr[8], sumi[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sumr[q] = 0.f; sumi[q] = 0.f; }
        for (int band = band_lo; band < band_hi; ++band) {
            float r[8], i[8];
            if (GEN) {
                const NoiseKey key = noise_key(seed, (unsigned)(b * nb + band));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = n0 + j + 512 * q;
                    float vl, vr;
                    noise_pair(key, (unsigned)idx, vl, vr);
                    r[q] = idx < row_len ? vl : 0.f;
                    i[q] = idx < row_len ? vr : 0.f;
                }
            } else {
                const float* rl = noise + ((long)(2 * b) * nb + band) * row_len;
                const float* rr = noise + ((long)(2 * b + 1) * nb + band) * row_len;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = n0 + j + 512 * q;
                    r[q] = idx < row_len ? rl[idx] : 0.f;
                    i[q] = idx < row_len ? rr[idx] : 0.f;
                }
            }
            fft4096_split_fwd(r, i, j, tw, lds);
            const f2* F = spec + FFT_N + (long)band * FFT_N + so;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f2 f = F[64 * q];
                const float t = r[q] * f.x - i[q] * f.y;
                i[q] = r[q] * f.y + i[q] * f.x;
                r[q] = t;
            }
            fft4096_split_inv(r, i, j, tw, lds + FFT_LDS);
            const float g = gains[b * nb + band], d = 10.f * decays[b * nb + band] + 1.f;
            if (MODE == 0) {
                const float gs = g * inv_nb;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float e = __expf(-d * fmaf((float)q, t512, t0)) * gs;
                    sumr[q] = fmaf(e, r[q], sumr[q]);
                    sumi[q] = fmaf(e, i[q], sumi[q]);
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
                sg = wave_sum_uniform(sg); sd = wave_sum_uniform(sd);
                if (lane_id() == 0) { red[wave_id()][band][0] = sg; red[wave_id()][band][1] = sd; }
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { accr[q] = sumr[q]; acci[q] = sumi[q]; }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = n0 + j + 512 * q;
            if (q < VQ && n < L) {
                if (bsplit == 1) { ir[(long)(2 * b) * L + n] = accr[q]; ir[(long)(2 * b + 1) * L + n] = acci[q]; }
                else { atomicAdd(ir + (long)(2 * b) * L + n, accr[q]); atomicAdd(ir + (long)(2 * b + 1) * L + n, acci[q]); }
            }
        }
    } else {
        __syncthreads();
        if (j < nb * 2 && (j >> 1) >= band_lo && (j >> 1) < band_hi) {
            const int band = j >> 1, k = j & 1;
            float a = 0.f;
            for (int v = 0; v < FFT_T / 64; ++v) a += red[v][band][k];
            part[(((long)b * gridDim.x + w) * nb + band) * 2 + k] = a;
        }
    }
}

// ---- long convolution ----------------------------------------------------------------------------------------------
// Frames of n1 = NA x 512 points, NA = 2^logNA: blocks of Lb = n1 / 2 samples. (Round 4 also built frames of 3 x 2^k points - one
// decimation-in-frequency radix-3 step around the power-of-two column transforms, a signal of N = 4 Lp samples as one 3 Lp-point frame
// instead of two 2 Lp-point frames - verified them against this path and measured the long convolution at 1.79 ms against 1.60
// (profiles/r04/reverb_r3_kernels.log): removed in round 5, the code is in the history of this file.)
struct ConvDims { int logNA, NA, n1, Lb, npairs; long N; };

// forward twiddle w_n1^(ka (j + 64 q)), q = 0..7, as a chain from two accurate sincospi evaluations
__device__ __forceinline__ void fourstep_twiddles(int ka, int j, int n1, float (&wr)[8], float (&wi)[8]) {
    const float inv = 2.f / (float)n1;        // power-of-two n1: exact (the reduced exponents are below 2^24); 3 x 2^k: rounded once (6e-8 relative)
    float s0, c0, s1, c1;
    sincospif(-(float)((ka * j) % n1) * inv, &s0, &c0);
    sincospif(-(float)((ka * 64) % n1) * inv, &s1, &c1);
    wr[0] = c0; wi[0] = s0;
#pragma unroll
    for (int q = 1; q < 8; ++q) { wr[q] = wr[q - 1] * c1 - wi[q - 1] * s1; wi[q] = wr[q - 1] * s1 + wi[q - 1] * c1; }
}


// Column pass, time -> A[ka][jb]. grid (NA * 512 / 4096 column tiles, pairs, signals), 512 threads; thread (j, c): column jb = tile * TC + c,
// elements ja = j + (NA / 8) q.   MODE 0: zero-padded blocks 2p (real) and 2p+1 (imaginary) of x (:570)
//                                 MODE 1: overlapped windows [k Lb, k Lb + 2 Lb) of gy, k = 2p, 2p+1
//                                 MODE 2: the zero-padded impulse responses of one batch item (L samples per row) as ONE complex frame,
//                                         left = real part, right = imaginary part (blockIdx.z = item); the row pass splits the
//                                         spectrum of the pair by its Hermitian symmetry (pair_spectrum_rows)
// src_shift: 1 = mono input, both signals of an item read row (sig >> 1) of src (the reference duplicates a mono input to stereo,
// functional.py:493-495: here the duplicate never exists); 0 otherwise.
template <int MODE>
__global__ __launch_bounds__(LoadGeom::T, 1) void conv_load_kernel(const float* __restrict__ src, const float* __restrict__ mix, const f2* __restrict__ tw,
                                                          f2* __restrict__ A, ConvDims d, int L, int src_shift = 0) {
    __shared__ f2 lds[LoadGeom::LDS];
    const ColCfg g = col_config<LOAD_LOG>(d.logNA, threadIdx.x);
    const int p = blockIdx.y;
    const long sig = blockIdx.z, srow = sig >> src_shift;
    const int jb = xcd_tile(blockIdx.x, gridDim.x) * g.TC + g.c;
    (void)mix;
    constexpr float scale = 1.f;
    // every load first, at an address that is always valid, the bounds applied to the values afterwards: a load under its own bounds
    // check is a branch with a wait in it - eight exposed round trips per thread (column loads of x / gy / the impulse responses: 196 -> 167, 248 -> 187, 96 -> 81 us)
    float r[8], i[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int tt = (g.j + g.T * q) * CV_NB + jb;          // time index inside the n1-point frame
        r[q] = 0.f; i[q] = 0.f;
        if (MODE == 2) {
            r[q] = src[2 * sig * L + (tt < L ? tt : L - 1)];
            i[q] = src[(2 * sig + 1) * L + (tt < L ? tt : L - 1)];
        } else if (MODE == 1 || q < 4) {                        // MODE 0: the upper half of the frame is padding
            const long n0 = (long)(2 * p) * d.Lb + tt, n1i = n0 + d.Lb;
            if (!(MODE == 1 && q >= 4)) r[q] = src[srow * d.N + (n0 < d.N ? n0 : d.N - 1)];      // MODE 1, q >= 4: same sample as i[q - 4]
            i[q] = src[srow * d.N + (n1i < d.N ? n1i : d.N - 1)];
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int tt = (g.j + g.T * q) * CV_NB + jb;
        if (MODE == 2) {
            r[q] = tt < L ? r[q] : 0.f;
            i[q] = tt < L ? i[q] : 0.f;
        } else if (MODE == 1 || q < 4) {
            const long n0 = (long)(2 * p) * d.Lb + tt, n1i = n0 + d.Lb;
            r[q] = n0 < d.N ? r[q] : 0.f;
            i[q] = n1i < d.N ? i[q] : 0.f;
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int q = 4; q < 8; ++q) r[q] = i[q - 4];
    }
    col_fft<-1>(r, i, g, tw, lds);
    f2* out = A + (sig * d.npairs + p) * (long)d.n1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {       // measured: the streaming hint pays for the impulse-response transforms only (99.6 -> 67 us)
        f2* o = out + (long)(g.j + g.T * q) * CV_NB + jb;
        if (MODE == 2) st_stream(o, f2{r[q], i[q]}); else *o = f2{r[q], i[q]};
    }
}

// Row pass. 512 threads = 8 waves, wave = one row ka of one signal, lane j holds columns j + 64 q.
//   MODE 0  forward:   Wout[p] = rowIFFT(rowFFT(A[p] tw) H) conj(tw)                        for every pair p of the signal
//   MODE 1  backward:  Wout[p] = rowIFFT(rowFFT(Ag[p] tw) conj(H)) conj(tw);  P = sum_p rowFFT(Ag[p] tw) conj(rowFFT(Ax[p] tw))
//   MODE 2  spectrum:  Z = rowFFT(A tw)   (pairs = 1)
// The two real impulse responses of a batch item (and, backward, the two real gradients w.r.t. them) travel as ONE complex frame,
// left + i right: half the frames to transform, store and read. H_left[k] = (Z[k] + conj Z[-k]) / 2, H_right[k] = (Z[k] - conj Z[-k]) / 2i;
// backward, d/d ir = Re IFFT(P) of each signal (the imaginary part is cross-talk of the two blocks packed into a frame), i.e. the
// transform of P's Hermitian part, so Pout = Herm(P_left) + i Herm(P_right), Herm(P)[k] = (P[k] + conj P[-k]) / 2. In the permuted
// order of the four-step transform, k = ka + NA kb, the mirrored frequency -k sits in row NA - ka at column 511 - kb (row 0: column
// (512 - kb) % 512; rows 0 and NA / 2 mirror themselves), so a workgroup takes rows together with their mirrors:
//   MODE 0, 2  grid (NA / 8, signals / items):  waves = 4 rows below NA / 2, their 4 mirrors (the two signals of an item run on the same
//              XCD - their workgroup ids differ by a multiple of 8 - so the second fetch of a Z row is served by its L2)
//   MODE 1     grid (NA / 4, items):  waves = (2 rows, their 2 mirrors) x the item's 2 signals: the four P rows that make one row of
//              Pout meet in one workgroup's LDS.
// x_shift: 1 = mono input: the two signals of an item are convolved from ONE set of column transforms of x (frames of item = sig >> 1;
// conv_load_kernel<0> is then launched per item): MODE 0 reads A, MODE 1 reads Ax, at the item's index.
template <int MODE>
__global__ __launch_bounds__(FFT_T, MODE == 1 ? 2 : 4) void conv_rows_kernel(const f2* __restrict__ A, const f2* __restrict__ Ax, const f2* __restrict__ tw,
                                                          f2* __restrict__ H, f2* __restrict__ Wout, f2* __restrict__ Pout, ConvDims d, int x_shift = 0) {
    __shared__ f2 lds_all[FFT_T / 64][FFT512_LDS];
    __shared__ f2 pex_all[MODE == 1 ? FFT_T / 64 : 1][MODE == 1 ? CV_NB : 1];      // MODE 1: the P rows meet here (one barrier; the transforms' own images stay private)
    constexpr int RW = MODE == 1 ? 2 : 4;                                          // rows below NA / 2 per workgroup
    const int j = lane_id(), v = wave_id();
    const int rw = v % RW, mir = (v / RW) & 1;
    const bool self = blockIdx.x == 0 && rw == 0;                                  // rows 0 and NA / 2
    int ka = blockIdx.x * RW + rw;
    if (mir) ka = self ? d.NA / 2 : d.NA - ka;
    const int ch = MODE == 1 ? v / (2 * RW) : MODE == 0 ? (int)(blockIdx.y & 1) : 0;
    const long item = MODE == 0 ? blockIdx.y >> 1 : blockIdx.y, sig = MODE == 2 ? item : 2 * item + ch;
    f2* lds = lds_all[v];
    const Fft512Tw t5 = fft512_twiddles(j, tw);
    float wr[8], wi[8];
    fourstep_twiddles(ka, j, d.n1, wr, wi);
    const long rowoff = (long)ka * CV_NB + j;
    float hr[8], hi[8], pr[8], pi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { pr[q] = 0.f; pi[q] = 0.f; hr[q] = 0.f; hi[q] = 0.f; }
    if (MODE != 2) {
        const f2* Z = H + item * (long)d.n1;
        const long mrow = (long)(ka ? d.NA - ka : 0) * CV_NB;
        f2 z[8], zm[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kb = j + 64 * q;
            z[q] = Z[rowoff + 64 * q];
            zm[q] = Z[mrow + (ka == 0 ? (CV_NB - kb) & (CV_NB - 1) : CV_NB - 1 - kb)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (ch == 0) { hr[q] = 0.5f * (z[q].x + zm[q].x); hi[q] = 0.5f * (z[q].y - zm[q].y); }
            else         { hr[q] = 0.5f * (z[q].y + zm[q].y); hi[q] = 0.5f * (zm[q].x - z[q].x); }
            if (MODE == 1) hi[q] = -hi[q];
        }
    }
    auto load_spec = [&](const f2* base, float (&r)[8], float (&i)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f2 v2 = base[rowoff + 64 * q];
            r[q] = v2.x * wr[q] - v2.y * wi[q];
            i[q] = v2.x * wi[q] + v2.y * wr[q];
        }
        fft512_wave<-1>(r, i, j, t5, lds);
    };
    auto store_time = [&](f2* base, float (&r)[8], float (&i)[8]) {
        fft512_wave<1>(r, i, j, t5, lds);
#pragma unroll
        for (int q = 0; q < 8; ++q) base[rowoff + 64 * q] = f2{r[q] * wr[q] + i[q] * wi[q], i[q] * wr[q] - r[q] * wi[q]};
    };
    for (int p = 0; p < d.npairs; ++p) {
        const long off = (sig * d.npairs + p) * (long)d.n1, xoff = ((sig >> x_shift) * d.npairs + p) * (long)d.n1;
        float r[8], i[8];
        load_spec(A + (MODE == 0 ? xoff : off), r, i);
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) H[sig * d.n1 + rowoff + 64 * q] = f2{r[q], i[q]};
        } else {
            if (MODE == 1) {
                float xr[8], xi[8];
                load_spec(Ax + xoff, xr, xi);
#pragma unroll
                for (int q = 0; q < 8; ++q) {          // P += G conj(X)
                    pr[q] += r[q] * xr[q] + i[q] * xi[q];
                    pi[q] += i[q] * xr[q] - r[q] * xi[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float t = r[q] * hr[q] - i[q] * hi[q];
                i[q] = r[q] * hi[q] + i[q] * hr[q];
                r[q] = t;
            }
            store_time(Wout + off, r, i);
        }
    }
    if (MODE == 1) {
        // the four P rows (this row and its mirror, left and right signal) -> rows ka of Pout, by the waves of the left signal
        f2* pex = pex_all[MODE == 1 ? v : 0];
#pragma unroll
        for (int q = 0; q < 8; ++q) pex[j + 64 * q] = f2{pr[q], pi[q]};
        __syncthreads();
        if (ch == 0) {
            const int vm = self ? v : v ^ RW;                                      // the wave holding the mirrored row of the left signal
            const f2* pR = pex_all[MODE == 1 ? v + 2 * RW : 0];
            const f2* mL = pex_all[MODE == 1 ? vm : 0];
            const f2* mR = pex_all[MODE == 1 ? vm + 2 * RW : 0];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kb = j + 64 * q, kbm = ka == 0 ? (CV_NB - kb) & (CV_NB - 1) : CV_NB - 1 - kb;
                const f2 b = pR[kb], cl = mL[kbm], cr = mR[kbm];
                const float qr = 0.5f * ((pr[q] + cl.x) - (b.y - cr.y));
                const float qi = 0.5f * ((pi[q] - cl.y) + (b.x + cr.x));
                pr[q] = qr; pi[q] = qi;
            }
            store_time(Pout + item * (long)d.n1, pr, pi);
        }
    }
}

// (launch bounds: 8 waves per SIMD = two workgroups per CU, <= 64 VGPRs; left to itself the forward instance took 66 and ran alone on its CU: 217 -> 250 us)
template <int MODE>
__global__ __launch_bounds__(ColsGeom::T, 8) void conv_cols_kernel(const f2* __restrict__ W, const f2* __restrict__ tw, const float* __restrict__ x,
                                                          const float* __restrict__ gy, const float* __restrict__ mix,
                                                          float* __restrict__ out, float* __restrict__ mix_part, ConvDims d, int L, int x_shift = 0) {
    __shared__ f2 lds[ColsGeom::LDS];
    __shared__ float red[ColsGeom::T / 64];
    const ColCfg g = col_config<COLS_LOG>(d.logNA, threadIdx.x);
    const long sig = MODE == 1 ? (long)blockIdx.z : blockIdx.y, xrow = sig >> x_shift;          // MODE 2: sig = batch item
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const float inv = 1.f / (float)d.n1;
    const float m = mix[MODE == 2 ? sig : sig >> 1];
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    float macc = 0.f;
    const int p_lo = MODE == 1 ? (int)blockIdx.y : 0, p_hi = MODE == 0 ? d.npairs : p_lo + 1;
    for (int p = p_lo; p < p_hi; ++p) {
        // the thread coordinates pass through an opaque move once per pair: without it every LDS / global address of the loop body is
        // hoisted out of the loop and the kernel needs 208 VGPRs (one workgroup per CU) instead of ~100
        ColCfg gl = g;
        if (MODE == 0) { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); gl.j += z; }
        const int jbl = tile * gl.TC + gl.c;
        const f2* in = W + (sig * d.npairs + p) * (long)d.n1;
        float r[8], i[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f2 v = in[(long)(gl.j + gl.T * q) * CV_NB + jbl]; r[q] = v.x; i[q] = v.y; }
        col_fft<1>(r, i, gl, tw, lds);
        // the signal values the epilogue combines with the frame: all requested together at clamped addresses (under the bounds checks
        // below each was a load, a wait and a store in turn; 255 -> 225 us forward, 285 -> 273 backward. Requesting them before the
        // transform instead gained nothing forward and cost occupancy backward)
        float xa[4], xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = (gl.j + gl.T * q) * CV_NB + jbl;
            if (MODE == 2) {                                   // x = the impulse responses (2 items, L)
                const int o = rr < L ? rr : L - 1;
                xa[q] = x[2 * sig * L + o]; xb[q] = x[(2 * sig + 1) * L + o];
            } else {
                const long na = (long)(2 * p) * d.Lb + rr, nbk = na + d.Lb;
                const long oa = na < d.N ? na : d.N - 1, ob = nbk < d.N ? nbk : d.N - 1;
                if (MODE == 0) { xa[q] = x[xrow * d.N + oa]; xb[q] = x[xrow * d.N + ob]; }
                else { xa[q] = gy[sig * d.N + oa]; xb[q] = gy[sig * d.N + ob]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = (gl.j + gl.T * q) * CV_NB + jbl;          // < Lb
            if (MODE == 2) {
                if (rr < L) {
                    const float ca = r[q] * inv, cb = i[q] * inv;
                    out[2 * sig * L + rr] = m * ca;
                    out[(2 * sig + 1) * L + rr] = m * cb;
                    macc = fmaf(xa[q], ca, fmaf(xb[q], cb, macc));
                    if (rr == 0) macc -= ca + cb;
                }
            } else {
                const long na = (long)(2 * p) * d.Lb + rr, nbk = na + d.Lb;
                if (MODE == 0) {
                    const float wa = fmaf(r[q], inv, carry[q]), wb = (i[q] + r[q + 4]) * inv;
                    carry[q] = i[q + 4] * inv;
                    if (na < d.N) out[sig * d.N + na] = fmaf(m, wa - xa[q], xa[q]);
                    if (nbk < d.N) out[sig * d.N + nbk] = fmaf(m, wb - xb[q], xb[q]);
                } else {
                    const float ca = r[q] * inv, cb = i[q] * inv;
                    if (na < d.N) out[sig * d.N + na] = fmaf(m, ca - xa[q], xa[q]);
                    if (nbk < d.N) out[sig * d.N + nbk] = fmaf(m, cb - xb[q], xb[q]);
                }
            }
        }
    }
    if (MODE == 2) {
        const float s = wave_sum(macc);
        if (lane_id() == 0) red[wave_id()] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f;
            for (int v = 0; v < ColsGeom::T / 64; ++v) a += red[v];
            mix_part[sig * gridDim.x + blockIdx.x] = a;
        }
    }
}


// ggain, gdecay (B, nb) and gmix (B) from the per-block partial sums, in fp64. One wave per output value, its lanes across the partial
// sums (one round trip to memory per output; a thread walking its partial sums alone paid one per element: 17 -> 4 us)
__global__ __launch_bounds__(256) void reverb_finalize_kernel(const float* __restrict__ part, const float* __restrict__ mix_part, float* __restrict__ ggain,
                                                              float* __restrict__ gdecay, float* __restrict__ gmix, int B, int nb, int chunks,
                                                              int mix_chunks) {
    const int o = blockIdx.x * (blockDim.x / 64) + wave_id(), l = lane_id();
    if (o < B * nb) {
        const int b = o / nb, k = o % nb;
        double a = 0.0, c = 0.0;
        for (int j = l; j < chunks; j += 64) {
            const float* p = part + (((long)b * chunks + j) * nb + k) * 2;
            a += (double)p[0]; c += (double)p[1];
        }
        a = wave_sum(a); c = wave_sum(c);
        if (l == 0) { ggain[o] = (float)a; gdecay[o] = (float)c; }
    } else if (o < B * nb + B) {
        const int b = o - B * nb;
        double m = 0.0;
        for (long j = l; j < mix_chunks; j += 64) m += (double)mix_part[(long)b * mix_chunks + j];
        m = wave_sum(m);
        if (l == 0) gmix[b] = (float)m;
    }
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
struct RvDims { ConvDims c; int nblk, VQ, nwin, ltiles, ctiles, rowgroups, chunk; long R; };
// Plan overrides (dasp_reverb_plan, dasp_hip.h): explicit arguments of the C ABI for the three choices the planner makes, -1 = the planner's
// own (rounds 2 - 4 read them from the environment). Process-wide; tests and developer A/Bs set them and set them back.
long g_plan_chunk = -1;
float g_plan_weight_limit = -1.f;
int g_plan_band_split = -1;
// Signals per pass of the long-convolution pipeline: all of them by default. Passes over chunks of signals that reuse chunk-sized scratch
// buffers were built to keep the 8 B per frame point the three kernels of a pass hand to each other in the 256 MB last-level cache; measured
// at (128, 2, 262144) it only cost time (fwd + bwd 2.85 ms in one pass, 2.91 / 2.94 / 2.98 / 3.28 ms with 128 / 64 / 32 / 16 signals per
// pass): the passes run at the rate of their HBM traffic either way and the extra launches and partially filled waves of small grids
// are not paid back. dasp_reverb_plan(chunk = signals per pass, ..) keeps the experiment reproducible.
inline int rv_chunk(long R, long frame_elems_per_signal) {
    (void)frame_elems_per_signal;
    long c = g_plan_chunk > 0 ? g_plan_chunk : R;
    if (c <= 0 || c > R) c = R;
    c &= ~1L;                                        // the two signals of a batch item stay together (they share mix)
    return (int)(c < 2 ? 2 : c);
}
// largest |rho| * 4096 served with the envelope inside the transform (fb_fused_kernel); dasp_reverb_plan(.., weight_limit = 0, ..) sends every
// item the per-band way (developer A/B and the route-equality test)
inline float rv_weight_limit() {
    if (g_plan_weight_limit >= 0.f && g_plan_weight_limit <= 8.f) return g_plan_weight_limit == 0.f ? -1.f : g_plan_weight_limit;
    return RV_WEIGHT_LIMIT;
}
// decay_bound > 0: the caller vouches that no band decay exceeds it (Processor.process_normalized: the validated upper end of the parameter
// range). If even that decay keeps an item on the envelope-inside-the-transform route, the per-band route's launch is not issued and the
// kernel takes every item down route 1 without looking (a value beyond the bound then costs accuracy - the fp32 transform of widely spread
// weights - not correctness of the launch structure). 0 = unknown: both launches, each item picks its route.
inline int rv_only_route1(float decay_bound, float tstep, float limit) {
    return decay_bound > 0.f && limit > 0.f && (10.f * decay_bound + 1.f) * tstep * (float)FFT_N <= limit;
}
// workgroups per (item, window) of the filter-bank kernel: the bands are dealt out when B * windows would leave most of the chip idle
inline int rv_band_split(int B, int nwin, int nb) {
    if (g_plan_band_split >= 1 && g_plan_band_split <= nb) return g_plan_band_split;
    // every share repeats the window's inverse transform (forward) and adds its part with atomics, so the bands are only dealt out until
    // ~128 workgroups exist (measured at 131072 samples, fwd + bwd: 1 item 0.148 -> 0.113 ms, 2 items 0.155 -> 0.129, 4 items 0.164 -> 0.146,
    // 8 items and more: no split is fastest; profiles/r02/reverb_band_split.log)
    int split = 1;
    while (split < nb && (long)B * nwin * split < 128) ++split;
    while (nb % split) ++split;             // equal shares
    if ((long)B * nwin < 128) return split;
    // From 128 workgroups on (round 5): a workgroup is compute-bound and 256 CUs take them in rounds, so 264 .. 511 workgroups (12 .. 23
    // items at the default sizes) leave most CUs idle in the second round. Rounds of 256 workgroups per unit of work, with 8 % per extra
    // share for the repeated inverse transform and the atomics: (12 / 16, c, 131072) fwd + bwd 0.269 / 0.296 -> 0.260 / 0.288 ms with two
    // shares, (20 / 24 / 32 items) stay at one or tie (profiles/r05/fb_split_sweep.log).
    // Only below 512 workgroups and only for a predicted gain of 10 % (beyond that one share per window was measured fastest or tied).
    if ((long)B * nwin >= 512) return 1;
    const double one = (double)(((long)B * nwin + 255) / 256);
    int best = 1;
    double best_cost = 0.9 * one;
    for (int s = 2; s <= 4 && s <= nb; ++s) {
        if (nb % s) continue;
        const long wg = (long)B * nwin * s;
        const double cost = (double)((wg + 255) / 256) / s * (1.0 + 0.08 * (s - 1));
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}
#define DASP_RV_LOAD(MODE_, GRID_, ...) hipLaunchKernelGGL((conv_load_kernel<MODE_>), GRID_, dim3(LoadGeom::T), 0, st, __VA_ARGS__)
#define DASP_RV_COLS(MODE_, GRID_, ...) hipLaunchKernelGGL(conv_cols_kernel<MODE_>, GRID_, dim3(ColsGeom::T), 0, st, __VA_ARGS__)
inline bool rv_dims(int B, long N, int L, int taps, RvDims* o) {
    RvDims d;
    long Lb = ColsGeom::N / 2;                       // n1 >= 8192 keeps every workgroup of the four-step kernels full
    while (Lb < L) Lb <<= 1;
    if (Lb > (1L << 20)) return false;               // NA = n1 / 512 <= 4096
    // Frame plan: blocks of Lb samples (the impulse response padded to a power of two) in 2 Lb-point frames
    d.c.Lb = (int)Lb; d.c.n1 = (int)(2 * Lb); d.c.NA = d.c.n1 / CV_NB;
    const long napow = d.c.NA;                       // column-transform length
    d.c.logNA = 0; while ((1L << d.c.logNA) < napow) ++d.c.logNA;
    d.c.N = N;
    d.nblk = (int)((N + d.c.Lb - 1) / d.c.Lb);
    d.c.npairs = (d.nblk + 1) / 2;
    d.ltiles = (int)(napow * CV_NB / LoadGeom::N);   // column tiles of the forward / inverse column kernels
    d.ctiles = (int)(napow * CV_NB / ColsGeom::N);
    d.rowgroups = d.c.NA / 8;                        // 8 rows (waves) per workgroup of the row pass
    d.R = 2L * B;
    d.VQ = (FFT_N - (taps - 1)) / 512;              // valid outputs per filter-bank window = 512 VQ
    if (d.VQ < 1) return false;                      // filters longer than 3585 taps do not fit the 4096-point window
    d.nwin = (L + 512 * d.VQ - 1) / (512 * d.VQ);
    if (d.R > 65535 || d.c.npairs > 65535 || B > 65535) return false;
    d.chunk = rv_chunk(d.R, (long)d.c.npairs * d.c.n1);
    *o = d;
    return true;
}
}  // namespace

extern "C" {

/* sizes[0] = Lb (block length), [1] = n1 (transform length), [2] = pairs of blocks per signal, [3] = blocks per signal,
 * [4] = complex elements of Fspec (twiddle table + band spectra of the filter bank + the taps), [5] = filter-bank windows per batch item,
 * [6] = complex elements of A (2B * pairs * n1), [7] = complex elements of H (B * n1: one complex frame per item = both impulse responses),
 * [8] = floats of ir / gir (2B * L), [9] = signals per pass of the long-convolution pipeline (chunk),
 * [10] = floats of mix_part, [11] = floats of the gain / decay partial sums,
 * [12] = complex elements of the scratch buffers W / Ag (chunk * pairs * n1, at least B * nb * 4096), [13] = complex elements of the scratch buffers Ah / P (chunk / 2 * n1: one complex frame per item of a pass) */
int dasp_reverb_plan(long chunk, float weight_limit, int band_split) {
    g_plan_chunk = chunk; g_plan_weight_limit = weight_limit; g_plan_band_split = band_split;
    return DASP_OK;
}
int dasp_reverb_sizes(int B, long N, int L, int taps, int nb, long* sizes) {
    if (!sizes || B <= 0 || N <= 0 || L <= 0 || taps <= 0 || nb <= 0 || nb > RV_BANDS_MAX) return DASP_ERR_ARG;
    RvDims d;
    if (!rv_dims(B, N, L, taps, &d)) return DASP_ERR_UNSUPPORTED;
    sizes[0] = d.c.Lb; sizes[1] = d.c.n1; sizes[2] = d.c.npairs; sizes[3] = d.nblk;
    sizes[4] = (long)(nb + 1) * FFT_N + ((long)nb * taps + 1) / 2; sizes[5] = d.nwin;
    sizes[6] = d.R * d.c.npairs * d.c.n1; sizes[7] = (long)B * d.c.n1;
    sizes[8] = d.R * L; sizes[9] = d.chunk;
    sizes[10] = (long)B * d.ctiles; sizes[11] = (long)B * d.nwin * nb * 2;
    sizes[12] = (long)d.chunk * d.c.npairs * d.c.n1; sizes[13] = (long)(d.chunk / 2) * d.c.n1;
    if (sizes[12] < (long)B * nb * FFT_N) sizes[12] = (long)B * nb * FFT_N;       // W / Ag also hold the per-item weighted band spectra of the filter bank
    return DASP_OK;
}

/* filters (nb, taps) fp32 -> Fspec (sizes[4] complex): the 4096-point twiddle table, conj(FFT(filter)) / 4096 per band, the taps */
int dasp_reverb_filter_spectrum(const float* filters, int nb, int taps, void* Fspec, void* stream) {
    if (!filters || !Fspec || nb <= 0 || nb > RV_BANDS_MAX || taps <= 0) return DASP_ERR_ARG;
    if (taps - 1 > FFT_N - 512) return DASP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fb_twiddle_kernel, dim3(FFT_N / 256), dim3(256), 0, (hipStream_t)stream, (f2*)Fspec);
    hipLaunchKernelGGL(fb_spectrum_kernel, dim3(nb), dim3(FFT_T), 0, (hipStream_t)stream, filters, (f2*)Fspec, taps);
    return rv_check();
}

/* Forward.  x (B,Cx,N), Cx = 2, or 1 for a mono input that the reference duplicates to stereo (functional.py:493-495; here the copy never
 * exists: both output channels read the one row); noise (2B, nb, L+taps-1); Fspec (sizes[4] complex); gains, decays (B, nb); mix (B); y (B,2,N).
 * Saved for backward: H (sizes[7] complex), ir (sizes[8] floats: the impulse responses) and, when A is not NULL, A (sizes[6] complex: the column transforms of x; pass NULL when no
 * gradient is needed and they go to a chunk-sized scratch instead: W2, sizes[12] complex).
 * Scratch: W (sizes[12] complex), Ah (sizes[13] complex). */
static int reverb_forward_impl(const float* x, const float* noise, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains,
                               const float* decays, const float* mix, float* y, void* A, void* H, void* W, void* W2, void* Ah, float* ir, int B,
                               int Cx, long N, int L, int taps, int nb, float decay_bound, void* stream) {
    if (!x || !Fspec || !gains || !decays || !mix || !y || (!A && !W2) || !H || !W || !Ah || !ir || B <= 0 || N <= 0 || L <= 0 || taps <= 0 ||
        nb <= 0 || nb > RV_BANDS_MAX)
        return DASP_ERR_ARG;
    RvDims d;
    if (Cx != 1 && Cx != 2) return DASP_ERR_ARG;
    if (!rv_dims(B, N, L, taps, &d)) return DASP_ERR_UNSUPPORTED;
    const int xs = Cx == 1 ? 1 : 0;         // mono input: both signals of an item read the one row of x (no duplicated copy)
    hipStream_t st = (hipStream_t)stream;
    const f2* tw = (const f2*)Fspec;
    const ConvDims one = ConvDims{d.c.logNA, d.c.NA, d.c.n1, d.c.Lb, 1, d.c.N};
    // 1. filter bank, envelope, gains, mean over bands -> impulse responses (functional.py:551-567)
    const int bsplit = rv_band_split(B, d.nwin, nb);
    if (bsplit > 1) {
        const hipError_t e = zero_async(ir, sizeof(float) * (size_t)d.R * L, st);
        if (e != hipSuccess) return (int)e;
    }
    const float* taps_f = reinterpret_cast<const float*>(tw + (long)(nb + 1) * FFT_N);
    const float limit = rv_weight_limit();
    const float tstep = L > 1 ? 1.f / (float)(L - 1) : 0.f;
    const int only1 = rv_only_route1(decay_bound, tstep, limit);
    hipLaunchKernelGGL(fb_wspectrum_kernel, dim3((unsigned)nb, (unsigned)B), dim3(FFT_T), 0, st, tw, taps_f, decays, (f2*)W, nb, taps, tstep);
    if ((long)L + taps - 1 >= (1L << 24)) return DASP_ERR_UNSUPPORTED;        // (the generator's sample index is a 24-bit factor; L <= 2^20 anyway)
    const dim3 fbgrid((unsigned)d.nwin, (unsigned)B, (unsigned)bsplit);
#define DASP_FB_FWD(ROUTE_, GEN_)                                                                                                           \
    hipLaunchKernelGGL((fb_fused_kernel<0, ROUTE_, GEN_>), fbgrid, dim3(FFT_T), 0, st, noise, tw, (const f2*)W, gains, decays, ir, (const float*)nullptr, \
                       (float*)nullptr, nb, L, taps, d.VQ, limit, seed, seed_dev, only1)
    if (noise) { DASP_FB_FWD(1, false); if (!only1) DASP_FB_FWD(0, false); } else { DASP_FB_FWD(1, true); if (!only1) DASP_FB_FWD(0, true); }
#undef DASP_FB_FWD
    for (long s0 = 0; s0 < d.R; s0 += d.chunk) {
        const unsigned ns = (unsigned)(d.R - s0 < d.chunk ? d.R - s0 : d.chunk);
        f2* Hc = (f2*)H + (s0 / 2) * d.c.n1;
        // 2. spectra of the chunk's impulse responses - one complex frame per batch item - in the permuted four-step order
        DASP_RV_LOAD(2, dim3((unsigned)d.ltiles, 1, ns / 2), (const float*)ir + s0 * L, (const float*)nullptr, tw, (f2*)Ah, one, L, 0);
        hipLaunchKernelGGL(conv_rows_kernel<2>, dim3((unsigned)d.rowgroups, ns / 2), dim3(FFT_T), 0, st, (const f2*)Ah, (const f2*)nullptr, tw, Hc,
                           (f2*)nullptr, (f2*)nullptr, one);
        // 3. overlap-add convolution (:570-572) and wet/dry mix (:575)
        // mono input: one set of column transforms per ITEM (both of its signals are convolutions of the same x), half the frames of A
        f2* Acx = A ? (f2*)A + (s0 >> xs) * d.c.npairs * d.c.n1 : (f2*)W2;
        DASP_RV_LOAD(0, dim3((unsigned)d.ltiles, (unsigned)d.c.npairs, ns >> xs), x + (s0 >> xs) * N, (const float*)nullptr, tw, Acx, d.c, L, 0);
        hipLaunchKernelGGL(conv_rows_kernel<0>, dim3((unsigned)d.rowgroups, ns), dim3(FFT_T), 0, st, (const f2*)Acx, (const f2*)nullptr, tw, Hc,
                           (f2*)W, (f2*)nullptr, d.c, xs);
        DASP_RV_COLS(0, dim3((unsigned)d.ctiles, ns), (const f2*)W, tw, x + (s0 >> xs) * N, (const float*)nullptr, mix + s0 / 2, y + s0 * N, (float*)nullptr, d.c, L, xs);
    }
    return rv_check();
}

/* Backward.  ir = the impulse responses the forward call left in its `ir` buffer (sizes[8] floats); gx (B,2,N) (mono input: the gradient
 * w.r.t. x is the sum of its two rows, left to the caller); ggain, gdecay (B, nb); gmix (B). x itself is not read (its column transforms A are).
 * Scratch: Ag, W (sizes[12] complex each), P (sizes[13] complex), gir (sizes[8] floats), part (sizes[11] floats), mix_part (sizes[10] floats). */
static int reverb_backward_impl(const float* ir, const float* gy, const float* noise, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains,
                                const float* decays, const float* mix, const void* A, const void* H, float* gx, float* ggain, float* gdecay,
                                float* gmix, void* Ag, void* W, void* P, float* gir, float* part, float* mix_part, int B, int Cx, long N, int L,
                                int taps, int nb, float decay_bound, void* stream) {
    if (!ir || !gy || !Fspec || !gains || !decays || !mix || !A || !H || !gx || !ggain || !gdecay || !gmix || !Ag || !W || !P ||
        !gir || !part || !mix_part || B <= 0 || N <= 0 || L <= 0 || taps <= 0 || nb <= 0 || nb > RV_BANDS_MAX)
        return DASP_ERR_ARG;
    RvDims d;
    if (Cx != 1 && Cx != 2) return DASP_ERR_ARG;
    if (!rv_dims(B, N, L, taps, &d)) return DASP_ERR_UNSUPPORTED;
    const int xs = Cx == 1 ? 1 : 0;         // mono input: one set of column transforms A per item; gx stays (B, 2, N), the caller adds its two rows
    hipStream_t st = (hipStream_t)stream;
    const f2* tw = (const f2*)Fspec;
    const ConvDims one = ConvDims{d.c.logNA, d.c.NA, d.c.n1, d.c.Lb, 1, d.c.N};
    for (long s0 = 0; s0 < d.R; s0 += d.chunk) {
        const unsigned ns = (unsigned)(d.R - s0 < d.chunk ? d.R - s0 : d.chunk);
        // overlapped windows of gy -> column transforms
        DASP_RV_LOAD(1, dim3((unsigned)d.ltiles, (unsigned)d.c.npairs, ns), gy + s0 * N, (const float*)nullptr, tw, (f2*)Ag, d.c, L, 0);
        // correlation with the impulse response (-> gx) and with the input blocks (-> d/dir, one complex frame per item), one row pass
        hipLaunchKernelGGL(conv_rows_kernel<1>, dim3((unsigned)d.rowgroups * 2, ns / 2), dim3(FFT_T), 0, st, (const f2*)Ag, (const f2*)A + (s0 >> xs) * d.c.npairs * d.c.n1, tw,
                           (f2*)H + (s0 / 2) * d.c.n1, (f2*)W, (f2*)P, d.c, xs);
        DASP_RV_COLS(1, dim3((unsigned)d.ctiles, (unsigned)d.c.npairs, ns), (const f2*)W, tw, (const float*)nullptr, gy + s0 * N, mix + s0 / 2, gx + s0 * N,
                     (float*)nullptr, d.c, L, 0);
        DASP_RV_COLS(2, dim3((unsigned)d.ctiles, ns / 2), (const f2*)P, tw, ir + s0 * L, (const float*)nullptr, mix + s0 / 2, gir + s0 * L,
                     mix_part + (s0 / 2) * d.ctiles, one, L, 0);
    }
    // d/dgain, d/ddecay: the filter bank again, weighted by gir
    const float* taps_f = reinterpret_cast<const float*>(tw + (long)(nb + 1) * FFT_N);
    const float limit = rv_weight_limit();
    const int only1 = rv_only_route1(decay_bound, L > 1 ? 1.f / (float)(L - 1) : 0.f, limit);
    hipLaunchKernelGGL(fb_wspectrum_kernel, dim3((unsigned)nb, (unsigned)B), dim3(FFT_T), 0, st, tw, taps_f, decays, (f2*)Ag, nb, taps,
                       L > 1 ? 1.f / (float)(L - 1) : 0.f);
    const dim3 fbgrid((unsigned)d.nwin, (unsigned)B, (unsigned)rv_band_split(B, d.nwin, nb));
#define DASP_FB_BWD(ROUTE_, GEN_)                                                                                                           \
    hipLaunchKernelGGL((fb_fused_kernel<1, ROUTE_, GEN_>), fbgrid, dim3(FFT_T), 0, st, noise, tw, (const f2*)Ag, gains, decays, (float*)nullptr,         \
                       (const float*)gir, part, nb, L, taps, d.VQ, limit, seed, seed_dev, only1)
    if (noise) { DASP_FB_BWD(1, false); if (!only1) DASP_FB_BWD(0, false); } else { DASP_FB_BWD(1, true); if (!only1) DASP_FB_BWD(0, true); }
#undef DASP_FB_BWD
    const int nfin = B * nb + B;                      // one wave per output value
    hipLaunchKernelGGL(reverb_finalize_kernel, dim3((nfin + 3) / 4), dim3(256), 0, st, part, mix_part, ggain, gdecay, gmix, B, nb, d.nwin,
                       d.ctiles);
    return rv_check();
}

int dasp_reverb_forward(const float* x, const float* noise, const void* Fspec, const float* gains, const float* decays, const float* mix,
                        float* y, void* A, void* H, void* W, void* W2, void* Ah, float* ir, int B, int Cx, long N, int L, int taps, int nb,
                        float decay_bound, void* stream) {
    if (!noise) return DASP_ERR_ARG;
    return reverb_forward_impl(x, noise, 0ULL, nullptr, Fspec, gains, decays, mix, y, A, H, W, W2, Ah, ir, B, Cx, N, L, taps, nb, decay_bound, stream);
}
int dasp_reverb_backward(const float* ir, const float* gy, const float* noise, const void* Fspec, const float* gains, const float* decays,
                         const float* mix, const void* A, const void* H, float* gx, float* ggain, float* gdecay, float* gmix,
                         void* Ag, void* W, void* P, float* gir, float* part, float* mix_part, int B, int Cx, long N, int L, int taps, int nb,
                         float decay_bound, void* stream) {
    if (!noise) return DASP_ERR_ARG;
    return reverb_backward_impl(ir, gy, noise, 0ULL, nullptr, Fspec, gains, decays, mix, A, H, gx, ggain, gdecay, gmix, Ag, W, P, gir, part, mix_part, B, Cx, N,
                                L, taps, nb, decay_bound, stream);
}
/* The same two calls with the white noise generated inside the filter-bank kernels from `seed` (the counter-based stream documented at
 * the top of reverb.hip) instead of read from memory: nothing of size (2B, nb, L + taps - 1) exists. Forward and backward must be given
 * the same seed. seed_dev (may be NULL): one 64-bit word in device memory that is ADDED to `seed` when the kernels run - a launch captured
 * into a HIP graph has its by-value seed frozen, the word lets every replay draw new noise (the caller bumps it between replays).
 * dasp_reverb_noise writes the stream out in the reference's layout, out (2B, nb, L + taps - 1) - a test hook. */
int dasp_reverb_forward_rng(const float* x, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains, const float* decays, const float* mix,
                            float* y, void* A, void* H, void* W, void* W2, void* Ah, float* ir, int B, int Cx, long N, int L, int taps, int nb,
                            float decay_bound, void* stream) {
    return reverb_forward_impl(x, nullptr, seed, seed_dev, Fspec, gains, decays, mix, y, A, H, W, W2, Ah, ir, B, Cx, N, L, taps, nb, decay_bound, stream);
}
int dasp_reverb_backward_rng(const float* ir, const float* gy, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains, const float* decays,
                             const float* mix, const void* A, const void* H, float* gx, float* ggain, float* gdecay, float* gmix,
                             void* Ag, void* W, void* P, float* gir, float* part, float* mix_part, int B, int Cx, long N, int L, int taps, int nb,
                             float decay_bound, void* stream) {
    return reverb_backward_impl(ir, gy, nullptr, seed, seed_dev, Fspec, gains, decays, mix, A, H, gx, ggain, gdecay, gmix, Ag, W, P, gir, part, mix_part, B, Cx, N,
                                L, taps, nb, decay_bound, stream);
}
int dasp_reverb_noise(unsigned long long seed, const unsigned long long* seed_dev, float* out, int B, int nb, long row_len, void* stream) {
    if (!out || B <= 0 || nb <= 0 || nb > RV_BANDS_MAX || row_len <= 0 || row_len >= (1L << 24) || (long)B * nb > 65535) return DASP_ERR_ARG;
    const unsigned gx = (unsigned)((row_len + 255) / 256 < 64 ? (row_len + 255) / 256 : 64);
    hipLaunchKernelGGL(reverb_noise_kernel, dim3(gx, (unsigned)(B * nb)), dim3(256), 0, (hipStream_t)stream, seed, seed_dev, out, nb, (int)row_len);
    return rv_check();
}

}  // extern "C"
