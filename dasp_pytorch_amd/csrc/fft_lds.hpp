// Complex FFTs held in registers + LDS (gfx950), 8 elements per thread, radix-8 Stockham passes with LDS exchanges in
// between. Element `idx = j + T q` (T = length / 8) lives in register q of thread j both before and after a transform
// (natural order in, natural order out), so a forward transform, a pointwise product and the inverse transform chain
// through registers without touching LDS in between. Three shapes, all used by reverb.hip:
//   fft512_wave  one 512-point transform per wave, no workgroup barriers      (row pass of the four-step long FFT)
//   fft4096_split_fwd / _inv  one 4096-point transform per 512-thread workgroup as radix-8 x 512 with a single
//                workgroup barrier                                             (filter bank, functional.py:548-558)
//   col_fft      4096 / P or 8192 / P transforms of P = 8..4096 points side by side in a 512- / 1024-thread workgroup, radix-8 passes plus
//                one radix-2/4 pass                                           (column pass of the four-step long FFT)
#pragma once
#include "common.hpp"

namespace dasp {

constexpr int FFT_N = 4096;          // transform length
constexpr int FFT_T = 512;           // threads per transform
constexpr int FFT512_LDS = 512 + 64;          // padded float2 elements of one 512-point transform
constexpr int FFT_LDS = FFT_N + FFT_N / 8;   // float2 elements of one padded exchange buffer (36,864 B)

// one pad slot per 8 elements: spreads the stride-8 and stride-64 scatters of the Stockham passes over the LDS banks
__device__ __forceinline__ int fft_pad(int i) { return i + (i >> 3); }

// ---- packed complex arithmetic (round 4 experiment, OFF by default) ------------------------------------------------------------------------
// A complex number is one f2 = one 64-bit register pair, and gfx950's packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: two fp32 operations per lane and instruction at the scalar instructions' issue rate - SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
// stays at 1.07 quad-cycles with 43 % of the instructions packed) do a complex add in ONE instruction and a complex product in TWO: their
// op_sel / neg modifiers read a source's halves swapped and negated for free, which is what "times +-i" and the cross terms of a product
// are. A radix-8 butterfly 54 -> 28 instructions, a twiddle product 4 -> 2; per kernel (static counts, scripts/kernel_valu.sh): filter bank
// 1211 -> 893 / 1642 -> 1423, row passes 895 -> 644 / 1576 -> 1124, inverse columns 980 -> 768, STFT loss 768 -> 645 / 1672 -> 1466.
// MEASURED, same box, interleaved (profiles/r04/ab_packed_fft.log): reverb (128, 2, 262144) 2.265 against 2.248 ms, (8, 2, 131072) 0.197
// against 0.201 ms, chain step unchanged, STFT loss 0.769 against 0.730 ms. A quarter fewer vector instructions buy nothing: the counters
// put these kernels at ~1.8 GHz of 2.4 (SQ_BUSY_CYCLES over the duration; profiles/r04/reverb_sq_counters.log) - they run at the power
// limit, the arithmetic done per joule is the same, and the clock gives back what the issue slots gained. Kept as a
// build-time A/B (-DDASP_FFT_PACKED=1); the scalar interfaces below (float r[8], i[8]) are the same either way.
#ifndef DASP_FFT_PACKED
#define DASP_FFT_PACKED 0
#endif
// a + rot(b), a - rot(b) with rot(b) = b * (-i) for DIR < 0 (b.y, -b.x) and b * (+i) for DIR > 0 (-b.y, b.x)
template <int DIR> __device__ __forceinline__ f2 cx_add_rot(f2 a, f2 b) {
    f2 o;
    if (DIR < 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
template <int DIR> __device__ __forceinline__ f2 cx_sub_rot(f2 a, f2 b) { return cx_add_rot<-DIR>(a, b); }
// rot(t) - t  (what w8^3 = (-1 -+ i) / sqrt 2 does to t, up to the factor 1 / sqrt 2)
template <int DIR> __device__ __forceinline__ f2 cx_rot_minus(f2 t) {
    f2 o;
    if (DIR < 0) asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(o) : "v"(t));    // (t.y - t.x, -t.x - t.y)
    else asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[0,1]" : "=v"(o) : "v"(t));          // (-t.y - t.x, t.x - t.y)
    return o;
}
// z * w (DIR < 0) or z * conj(w) (DIR > 0)
template <int DIR> __device__ __forceinline__ f2 cx_mul(f2 z, f2 w) {
    f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o) : "v"(z), "v"(w));                                               // (z.x w.x, z.y w.x)
    if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(o) : "v"(z), "v"(w));   // + (-z.y w.y, z.x w.y)
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(o) : "v"(z), "v"(w));          // + (z.y w.y, -z.x w.y)
    return o;
}
template <int DIR> __device__ __forceinline__ void dft4v(f2& a, f2& b, f2& c, f2& d) {        // 4-point DFT in place, natural order
    const f2 s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
    a = s0 + s1; c = s0 - s1;
    b = cx_add_rot<DIR>(d0, d1); d = cx_sub_rot<DIR>(d0, d1);
}
template <int DIR> __device__ __forceinline__ void radix8v(f2 (&z)[8]) {                      // X[q] = sum_r x[r] exp(DIR 2 pi i q r / 8)
    constexpr float H = 0.70710678118654752440f;
    f2 a[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = z[k] + z[k + 4]; a[k + 4] = z[k] - z[k + 4]; }
    // odd half times w8^k: w8 = (1 -+ i) / sqrt 2 -> (t + rot t) H;  w8^2 = -+i folded into the butterflies below;  w8^3 -> (rot t - t) H
    a[5] = cx_add_rot<DIR>(a[5], a[5]) * H;
    a[7] = cx_rot_minus<DIR>(a[7]) * H;
    dft4v<DIR>(a[0], a[1], a[2], a[3]);
    {
        const f2 s0 = cx_add_rot<DIR>(a[4], a[6]), d0 = cx_sub_rot<DIR>(a[4], a[6]), s1 = a[5] + a[7], d1 = a[5] - a[7];
        a[4] = s0 + s1; a[6] = s0 - s1;
        a[5] = cx_add_rot<DIR>(d0, d1); a[7] = cx_sub_rot<DIR>(d0, d1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { z[2 * k] = a[k]; z[2 * k + 1] = a[4 + k]; }
}

// z * (-i) for DIR < 0, z * (+i) for DIR > 0
template <int DIR> __device__ __forceinline__ void rot90(float& re, float& im) {
    const float t = re;
    if (DIR < 0) { re = im; im = -t; } else { re = -im; im = t; }
}

// 4-point DFT of (c0..c3) in place, outputs in natural order
template <int DIR>
__device__ __forceinline__ void dft4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3) {
#if DASP_FFT_PACKED
    f2 a = f2{r0, i0}, b = f2{r1, i1}, c = f2{r2, i2}, d = f2{r3, i3};
    dft4v<DIR>(a, b, c, d);
    r0 = a.x; i0 = a.y; r1 = b.x; i1 = b.y; r2 = c.x; i2 = c.y; r3 = d.x; i3 = d.y;
#else
    const float s0r = r0 + r2, s0i = i0 + i2, d0r = r0 - r2, d0i = i0 - i2;
    const float s1r = r1 + r3, s1i = i1 + i3;
    float d1r = r1 - r3, d1i = i1 - i3;
    rot90<DIR>(d1r, d1i);
    r0 = s0r + s1r; i0 = s0i + s1i;
    r2 = s0r - s1r; i2 = s0i - s1i;
    r1 = d0r + d1r; i1 = d0i + d1i;
    r3 = d0r - d1r; i3 = d0i - d1i;
#endif
}

// 8-point DFT, X[q] = sum_r x[r] exp(DIR 2 pi i q r / 8), natural order in and out
template <int DIR> __device__ __forceinline__ void radix8(float (&r)[8], float (&i)[8]) {
#if DASP_FFT_PACKED
    f2 z[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = f2{r[k], i[k]};
    radix8v<DIR>(z);
#pragma unroll
    for (int k = 0; k < 8; ++k) { r[k] = z[k].x; i[k] = z[k].y; }
#else
    constexpr float H = 0.70710678118654752440f;
    float ar[8], ai[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ar[k] = r[k] + r[k + 4]; ai[k] = i[k] + i[k + 4];
        ar[k + 4] = r[k] - r[k + 4]; ai[k + 4] = i[k] - i[k + 4];
    }
    // odd half: a[4 + k] *= w8^k
    {
        const float p = ar[5], q = ai[5];
        if (DIR < 0) { ar[5] = (p + q) * H; ai[5] = (q - p) * H; } else { ar[5] = (p - q) * H; ai[5] = (p + q) * H; }
    }
    rot90<DIR>(ar[6], ai[6]);
    {
        const float p = ar[7], q = ai[7];
        if (DIR < 0) { ar[7] = (q - p) * H; ai[7] = -(p + q) * H; } else { ar[7] = -(p + q) * H; ai[7] = (p - q) * H; }
    }
    dft4<DIR>(ar[0], ai[0], ar[1], ai[1], ar[2], ai[2], ar[3], ai[3]);
    dft4<DIR>(ar[4], ai[4], ar[5], ai[5], ar[6], ai[6], ar[7], ai[7]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[2 * k] = ar[k]; i[2 * k] = ai[k];
        r[2 * k + 1] = ar[4 + k]; i[2 * k + 1] = ai[4 + k];
    }
#endif
}

// w^1..w^7 of one pass's twiddle (forward sign), powers by repeated products (each within ~2 ulp); kept as pairs: a twiddle is one
// 64-bit operand of the packed product
struct Tw8 { f2 w[7]; };
__device__ __forceinline__ Tw8 tw_powers(f2 w) {
    Tw8 t;
    t.w[0] = w;
#if DASP_FFT_PACKED
    t.w[1] = cx_mul<-1>(w, w); t.w[2] = cx_mul<-1>(t.w[1], w); t.w[3] = cx_mul<-1>(t.w[1], t.w[1]); t.w[4] = cx_mul<-1>(t.w[3], w);
    t.w[5] = cx_mul<-1>(t.w[2], t.w[2]); t.w[6] = cx_mul<-1>(t.w[3], t.w[2]);
#else
    auto mul = [](f2 a, f2 b) { return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; };
    auto sqr = [](f2 a) { return f2{a.x * a.x - a.y * a.y, 2.f * a.x * a.y}; };
    t.w[1] = sqr(w); t.w[2] = mul(t.w[1], w); t.w[3] = sqr(t.w[1]); t.w[4] = mul(t.w[3], w); t.w[5] = sqr(t.w[2]); t.w[6] = mul(t.w[3], t.w[2]);
#endif
    return t;
}
// x[k] *= w^k (DIR < 0) or conj(w)^k (DIR > 0), k = 1..7
template <int DIR> __device__ __forceinline__ void twiddle8(float (&r)[8], float (&i)[8], const Tw8& w) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#if DASP_FFT_PACKED
        const f2 o = cx_mul<DIR>(f2{r[k], i[k]}, w.w[k - 1]);
        r[k] = o.x; i[k] = o.y;
#else
        const float wr = w.w[k - 1].x, wi = w.w[k - 1].y;
        float t;
        if (DIR < 0) { t = r[k] * wr - i[k] * wi; i[k] = r[k] * wi + i[k] * wr; }
        else { t = r[k] * wr + i[k] * wi; i[k] = i[k] * wr - r[k] * wi; }
        r[k] = t;
#endif
    }
}

// ---- 512-point transform held by one wave --------------------------------------------------------------------------
struct Fft512Tw { Tw8 s1, s2; };
__device__ __forceinline__ Fft512Tw fft512_twiddles(int j, const f2* __restrict__ tw) {      // j = lane, tw = 4096-entry table
    Fft512Tw t;
    t.s1 = tw_powers(tw[(j & 7) * 64]);       // Ns = 8: w_512^(8 (j % 8))
    t.s2 = tw_powers(tw[j * 8]);              // Ns = 64: w_512^j
    return t;
}
__device__ __forceinline__ void wave_exchange(float (&r)[8], float (&i)[8], f2* lds, int wbase, int wstride, int rbase) {
    wave_lds_sync();
    f2* wp = lds + wbase;
#pragma unroll
    for (int q = 0; q < 8; ++q) wp[q * wstride] = f2{r[q], i[q]};
    wave_lds_sync();
    const f2* rp = lds + rbase;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f2 v = rp[q * 72]; r[q] = v.x; i[q] = v.y; }
}
template <int DIR>
__device__ __forceinline__ void fft512_wave(float (&r)[8], float (&i)[8], int j, const Fft512Tw& tw, f2* lds) {
    const int rb = j + (j >> 3);              // fft_pad(j + 64 q) = rb + 72 q
    radix8<DIR>(r, i);
    wave_exchange(r, i, lds, 9 * j, 1, rb);
    twiddle8<DIR>(r, i, tw.s1);
    radix8<DIR>(r, i);
    wave_exchange(r, i, lds, (j >> 3) * 72 + (j & 7), 9, rb);
    twiddle8<DIR>(r, i, tw.s2);
    radix8<DIR>(r, i);
}

// ---- 4096 = 8 x 512 split: one workgroup barrier per transform ------------------------------------------------------
// forward:  radix-8 over r of x[j + 512 r] in registers, twiddle w_4096^(j k1), transpose through LDS so that wave k1 owns all j,
//           512-point transform inside the wave. Output: wave k1, lane l, register s = X[k1 + 8 (l + 64 s)] ("split order").
// inverse:  the mirror image, from split order back to thread j, register r = N x[j + 512 r].
// A pointwise product between the two only needs the other factor in split order. Each direction uses one buffer of FFT_LDS
// elements (8 rows of 576): the workgroup-wide transpose, then each wave's private exchanges inside its own row; with the forward
// transform on one buffer and the inverse on the other, one barrier per transform is enough: a buffer is written again only by threads that have passed the other
// buffer's barrier, which every thread reaches after executing (LDS is in order per wave) its reads of the first.
struct SplitTw { Tw8 outer; Fft512Tw inner; };
__device__ __forceinline__ SplitTw split_twiddles(int j, const f2* __restrict__ tw) {
    SplitTw t;
    t.outer = tw_powers(tw[j]);
    t.inner = fft512_twiddles(j & 63, tw);
    return t;
}
__device__ __forceinline__ void fft4096_split_fwd(float (&r)[8], float (&i)[8], int j, const SplitTw& tw, f2* buf) {
    radix8<-1>(r, i);
    twiddle8<-1>(r, i, tw.outer);
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[q * FFT512_LDS + j] = f2{r[q], i[q]};
    __syncthreads();
    f2* row = buf + (j >> 6) * FFT512_LDS;
    const int l = j & 63;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f2 v = row[l + 64 * q]; r[q] = v.x; i[q] = v.y; }
    fft512_wave<-1>(r, i, l, tw.inner, row);
}
__device__ __forceinline__ void fft4096_split_inv(float (&r)[8], float (&i)[8], int j, const SplitTw& tw, f2* buf) {
    f2* row = buf + (j >> 6) * FFT512_LDS;
    const int l = j & 63;
    fft512_wave<1>(r, i, l, tw.inner, row);
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) row[l + 64 * q] = f2{r[q], i[q]};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f2 v = buf[q * FFT512_LDS + j]; r[q] = v.x; i[q] = v.y; }
    twiddle8<1>(r, i, tw.outer);
    radix8<1>(r, i);
}

// ---- the same forward transform holding 6 twiddle registers instead of 42 --------------------------------------------------
// For kernels that keep several spectra in registers across a loop of transforms: only the three base twiddles stay live, their powers
// are rebuilt in front of each pass (18 complex products per transform, the same operations as tw_powers: results identical to
// fft4096_split_fwd). The empty asm keeps the compiler from hoisting the powers out of the caller's loop, which would bring the 42
// registers back.
struct SplitTwLean { f2 outer, s1, s2; };
__device__ __forceinline__ SplitTwLean split_twiddles_lean(int j, const f2* __restrict__ tw) {
    SplitTwLean t;
    t.outer = tw[j]; t.s1 = tw[(j & 7) * 64]; t.s2 = tw[(j & 63) * 8];
    return t;
}
__device__ __forceinline__ Tw8 tw_powers_here(f2 w) {
    asm volatile("" : "+v"(w.x), "+v"(w.y));
    return tw_powers(w);
}
__device__ __forceinline__ void fft4096_split_fwd_lean(float (&r)[8], float (&i)[8], int j, const SplitTwLean& tw, f2* buf) {
    radix8<-1>(r, i);
    twiddle8<-1>(r, i, tw_powers_here(tw.outer));
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[q * FFT512_LDS + j] = f2{r[q], i[q]};
    __syncthreads();
    f2* row = buf + (j >> 6) * FFT512_LDS;
    const int l = j & 63, rb = l + (l >> 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f2 v = row[l + 64 * q]; r[q] = v.x; i[q] = v.y; }
    radix8<-1>(r, i);
    wave_exchange(r, i, row, 9 * l, 1, rb);
    twiddle8<-1>(r, i, tw_powers_here(tw.s1));
    radix8<-1>(r, i);
    wave_exchange(r, i, row, (l >> 3) * 72 + (l & 7), 9, rb);
    twiddle8<-1>(r, i, tw_powers_here(tw.s2));
    radix8<-1>(r, i);
}

// ---- TC = 2^LOGN / P transforms of P points side by side (batch index fastest in LDS and across lanes) ------------
// A workgroup of 2^LOGN / 8 threads holds 2^LOGN elements. LOGN = 12: 512 threads; LOGN = 13: 1024 threads, at P = 256 a tile is then 32
// columns wide, i.e. whole 128-byte lines of a float signal (pays in the epilogue kernels that read and write several float streams).
template <int LOGN> struct ColGeom { static constexpr int N = 1 << LOGN, T = N / 8, LDS = N + N / 8; };
struct ColCfg { int P, logP, T, TC, j, c; };      // thread (j, c): transform c, elements j + T q
template <int LOGN> __device__ __forceinline__ ColCfg col_config(int logP, int t) {
    ColCfg g;
    g.logP = logP; g.P = 1 << logP; g.T = g.P >> 3; g.TC = (1 << LOGN) >> logP;
    g.c = t & (g.TC - 1); g.j = t >> (LOGN - logP);
    return g;
}
__device__ __forceinline__ void col_exchange(float (&r)[8], float (&i)[8], f2* lds, const ColCfg& g, int wbase, int Ns) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[fft_pad(wbase + q * Ns) * g.TC + g.c] = f2{r[q], i[q]};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f2 v = lds[fft_pad(g.j + q * g.T) * g.TC + g.c]; r[q] = v.x; i[q] = v.y; }
}
template <int DIR> __device__ __forceinline__ void cmul_dir(float& re, float& im, float wr, float wi) {
#if DASP_FFT_PACKED
    const f2 o = cx_mul<DIR>(f2{re, im}, f2{wr, wi});
    re = o.x; im = o.y;
#else
    const float t = DIR < 0 ? re * wr - im * wi : re * wr + im * wi;
    im = DIR < 0 ? re * wi + im * wr : im * wr - re * wi;
    re = t;
#endif
}
// tw = the 4096-entry forward table; the twiddles of a pass are formed right before it (workgroups that run one transform per
// thread have nothing to amortise them over, and keeping all of them live costs ~50 VGPRs)
template <int DIR>
__device__ __forceinline__ void col_fft(float (&r)[8], float (&i)[8], const ColCfg& g, const f2* __restrict__ tw, f2* lds) {
    const int a = g.logP / 3, mul = FFT_N >> g.logP;          // w_P^e = tw[e * mul]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < a) {                                   // workgroup-uniform
            const int Ns = 1 << (3 * k);
            if (k > 0) twiddle8<DIR>(r, i, tw_powers(tw[((g.j & (Ns - 1)) * (g.P >> (3 * k + 3))) * mul]));
            radix8<DIR>(r, i);
            if (Ns * 8 < g.P) col_exchange(r, i, lds, g, ((g.j >> (3 * k)) << (3 * k + 3)) + (g.j & (Ns - 1)), Ns);
        }
    }
    const int R = g.P >> (3 * a);
    if (R == 4) {                                      // two radix-4 butterflies on registers (m, m+2, m+4, m+6), u = j + T m
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const f2 w = tw[(((g.j + g.T * m) * k) & (g.P - 1)) * mul];
                cmul_dir<DIR>(r[m + 2 * k], i[m + 2 * k], w.x, w.y);
            }
            dft4<DIR>(r[m], i[m], r[m + 2], i[m + 2], r[m + 4], i[m + 4], r[m + 6], i[m + 6]);
        }
    } else if (R == 2) {                               // four radix-2 butterflies on registers (m, m+4), u = j + T m
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f2 w = tw[(g.j + g.T * m) * mul];
            cmul_dir<DIR>(r[m + 4], i[m + 4], w.x, w.y);
            const float ar = r[m], ai = i[m];
            r[m] = ar + r[m + 4]; i[m] = ai + i[m + 4];
            r[m + 4] = ar - r[m + 4]; i[m + 4] = ai - i[m + 4];
        }
    }
}

}  // namespace dasp
