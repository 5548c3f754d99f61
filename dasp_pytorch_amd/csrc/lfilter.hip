// lfilter_via_fsm for filters longer than one biquad (gfx950): K = 4 .. 16 coefficients, i.e. orders 3 .. 15.
//
// Replaces dasp_pytorch/signal.py:95-133 for the orders the cascaded-biquad kernels (sosfilt.hip, K <= 3) do not take. The reference
// evaluates H = rfft(b) / rfft(a) on a zero-padded grid and multiplies spectra (frequency sampling); as everywhere in this library the
// filter is run as the exact recurrence instead - the two agree to rounding once the impulse response has decayed within the signal.
// The reference's only caller uses K = 2 (functional.py:372-380), so this is the boundary's long tail: plain double arithmetic whatever
// the dtype of x (a direct form of order 15 in fp32 would lose most of its digits), one thread per (row, chunk of time), every line
// checkable against the formulas it restates.
//
//   forward   direct form II:  w[n] = x[n] - sum_{j>=1} a_j w[n-j],  y[n] = sum_k b_k w[n-k]      (a normalised to a_0 = 1 by the caller)
//   backward  gx = the transposed form run backwards in time over gy (the adjoint of an LTI filter is the filter itself, time reversed);
//             dL/db_k = sum_n gy[n] w[n-k],  dL/da_k = - sum_n gx[n] w[n-k]   (k >= 1; w = the all-pole signal saved by the forward pass)
//
// Time is cut into P chunks of L samples that run side by side (a single thread walking 262144 samples took 85 - 170 ms: one wave issues
// an instruction every ~6 cycles whatever else the chip is doing). The recurrences are linear, so a chunk's end state is
// (its end state from a zero start state) + Phi (its true start state), Phi = the state transition over L samples:
//   pass 0  every chunk from a zero state, end states E only; M more threads per row run L samples without input from the unit states:
//           the columns of Phi
//   chain   S_{c+1} = Phi S_c + E_c along the row (16 lanes per row, one per state component; P - 1 steps)
//   pass 1  every chunk again from its true start state, with the outputs
// wsave is time-major, (N, rows): consecutive lanes = consecutive rows read consecutive doubles.
#include "common.hpp"

namespace dasp {

constexpr int LF_BL = 16;          // samples per register block: one round trip to memory per block instead of per sample

// PASS 0: threads [0, rows P): chunk end states from a zero start -> E; threads [rows P, rows P + rows (MK - 1)): unit-state runs -> Phi
// PASS 1: threads [0, rows P): the chunks from their start states S (null: zero - the single-chunk case) with y and wsave
// states: (chunk, row, MK - 1) doubles, component j = w[n0 - 1 - j];  Phi: (row, MK - 1 out, MK - 1 in)
template <int MK, typename T, int PASS>
__global__ __launch_bounds__(64) void lfilt_fwd_kernel(const T* __restrict__ x, const double* __restrict__ bn, const double* __restrict__ an,
                                                       T* __restrict__ y, double* __restrict__ wsave, const double* __restrict__ S,
                                                       double* __restrict__ E, double* __restrict__ Phi, int rows, int bcast, long N, int K,
                                                       long L, int P) {
    constexpr int M = MK - 1;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const long nmain = (long)rows * P;
    if (t >= nmain + (PASS == 0 ? (long)rows * M : 0)) return;
    const bool unit = PASS == 0 && t >= nmain;
    const int r = (int)(unit ? (t - nmain) % rows : t % rows);
    const int c = (int)(unit ? 0 : t / rows), ui = (int)(unit ? (t - nmain) / rows : -1);
    const double* bp = bn + (bcast ? 0 : (size_t)r * K);
    const double* ap = an + (bcast ? 0 : (size_t)r * K);
    double b[MK], a[MK], w[MK];                 // w[j] = w[n-1-j]  (w[MK-1] is never read)
#pragma unroll
    for (int k = 0; k < MK; ++k) { b[k] = k < K ? bp[k] : 0.0; a[k] = k < K && k > 0 ? ap[k] : 0.0; w[k] = 0.0; }
    if (unit) {
#pragma unroll
        for (int j = 0; j < M; ++j) w[j] = j == ui ? 1.0 : 0.0;
    } else if (PASS == 1 && S) {
#pragma unroll
        for (int j = 0; j < M; ++j) w[j] = S[((size_t)c * rows + r) * M + j];
    }
    const long n_lo = (long)c * L, n_hi = unit ? L : (n_lo + L < N ? n_lo + L : N);
    const T* xr = x + (size_t)r * N;
    T* yr = PASS == 1 ? y + (size_t)r * N : nullptr;
    T xb[LF_BL], xn[LF_BL];
#pragma unroll
    for (int s = 0; s < LF_BL; ++s) { const long m = n_lo + s; xb[s] = unit ? (T)0 : xr[m < N ? m : N - 1]; }
    for (long n0 = n_lo; n0 < n_hi; n0 += LF_BL) {
#pragma unroll
        for (int s = 0; s < LF_BL; ++s) { const long m = n0 + LF_BL + s; xn[s] = unit ? (T)0 : xr[m < N ? m : N - 1]; }      // the next block, requested now
#pragma unroll
        for (int s = 0; s < LF_BL; ++s) {
            const long n = n0 + s;
            if (n < n_hi) {
                double wn = (double)xb[s];
#pragma unroll
                for (int j = MK - 1; j >= 1; --j) wn = fma(-a[j], w[j - 1], wn);        // (the newest state last: one dependent operation per sample)
                if (PASS == 1) {
                    double yn = b[0] * wn;
#pragma unroll
                    for (int k = 1; k < MK; ++k) yn = fma(b[k], w[k - 1], yn);
                    yr[n] = (T)yn;
                    if (wsave) wsave[(size_t)n * rows + r] = wn;
                }
#pragma unroll
                for (int j = MK - 1; j > 0; --j) w[j] = w[j - 1];
                w[0] = wn;
            }
        }
#pragma unroll
        for (int s = 0; s < LF_BL; ++s) xb[s] = xn[s];
    }
    if (PASS == 0) {
        if (unit) {
#pragma unroll
            for (int j = 0; j < M; ++j) Phi[((size_t)r * M + j) * M + ui] = w[j];
        } else {
#pragma unroll
            for (int j = 0; j < M; ++j) E[((size_t)c * rows + r) * M + j] = w[j];
        }
    }
}

// The same three steps backwards in time: state = the M registers z of the transposed form; chunk c starts at its upper end from Z_c.
// PASS 1 adds the chunk's share of the coefficient correlations to gb, ga (rows, K) with double atomics (zeroed by the caller); ga[:, 0] = 0.
template <int MK, typename T, int PASS>
__global__ __launch_bounds__(64) void lfilt_bwd_kernel(const T* __restrict__ gy, const double* __restrict__ bn, const double* __restrict__ an,
                                                       const double* __restrict__ wsave, T* __restrict__ gx, double* __restrict__ gb,
                                                       double* __restrict__ ga, const double* __restrict__ S, double* __restrict__ E,
                                                       double* __restrict__ Phi, int rows, int bcast, long N, int K, long L, int P) {
    constexpr int M = MK - 1;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const long nmain = (long)rows * P;
    if (t >= nmain + (PASS == 0 ? (long)rows * M : 0)) return;
    const bool unit = PASS == 0 && t >= nmain;
    const int r = (int)(unit ? (t - nmain) % rows : t % rows);
    const int c = (int)(unit ? 0 : t / rows), ui = (int)(unit ? (t - nmain) / rows : -1);
    const double* bp = bn + (bcast ? 0 : (size_t)r * K);
    const double* ap = an + (bcast ? 0 : (size_t)r * K);
    double b[MK], a[MK], z[MK];                                   // (z[MK-1] stays 0)
#pragma unroll
    for (int k = 0; k < MK; ++k) { b[k] = k < K ? bp[k] : 0.0; a[k] = k < K && k > 0 ? ap[k] : 0.0; z[k] = 0.0; }
    if (unit) {
#pragma unroll
        for (int j = 0; j < M; ++j) z[j] = j == ui ? 1.0 : 0.0;
    } else if (PASS == 1 && S) {
#pragma unroll
        for (int j = 0; j < M; ++j) z[j] = S[((size_t)c * rows + r) * M + j];
    }
    double sb[PASS == 1 ? MK : 1], sa[PASS == 1 ? MK : 1];
    if (PASS == 1) {
#pragma unroll
        for (int k = 0; k < MK; ++k) { sb[k] = 0.0; sa[k] = 0.0; }
    }
    const long n_lo = (long)c * L, n_hi = unit ? L : (n_lo + L < N ? n_lo + L : N);
    const T* gr = gy + (size_t)r * N;
    T* or_ = PASS == 1 && gx ? gx + (size_t)r * N : nullptr;
    const long nblk = (n_hi - n_lo + LF_BL - 1) / LF_BL;
    for (long blk = nblk - 1; blk >= 0; --blk) {
        const long n0 = n_lo + blk * LF_BL;
        // the block's gy and the window w[n0 - (MK - 1) .. n0 + LF_BL - 1] it is correlated with: all requested together (out-of-range
        // entries at a clamped address, zeroed afterwards); win[i] = w[n0 - (MK - 1) + i]
        T gblk[LF_BL];
        double win[PASS == 1 ? LF_BL + MK - 1 : 1];
#pragma unroll
        for (int s = 0; s < LF_BL; ++s) { const long n = n0 + s; gblk[s] = unit ? (T)0 : gr[n < N ? n : N - 1]; }
        if (PASS == 1) {
#pragma unroll
            for (int i = 0; i < LF_BL + MK - 1; ++i) {
                const long m = n0 - (MK - 1) + i;
                win[i] = wsave[(size_t)(m < 0 ? 0 : (m < N ? m : N - 1)) * rows + r];
            }
#pragma unroll
            for (int i = 0; i < LF_BL + MK - 1; ++i) { const long m = n0 - (MK - 1) + i; win[i] = (m >= 0 && m < N) ? win[i] : 0.0; }
        }
#pragma unroll
        for (int s = LF_BL - 1; s >= 0; --s) {
            const long n = n0 + s;
            if (n < n_hi) {
                const double g = (double)gblk[s];
                const double o = fma(b[0], g, z[0]);            // transposed direct form II, time reversed
#pragma unroll
                for (int i = 0; i < MK - 1; ++i) z[i] = fma(b[i + 1], g, fma(-a[i + 1], o, z[i + 1]));
                if (PASS == 1) {
                    if (or_) or_[n] = (T)o;
#pragma unroll
                    for (int k = 0; k < MK; ++k) {              // w[n - k] = win[s + MK - 1 - k]; taps k >= K meet zero coefficients only
                        sb[k] = fma(g, win[s + MK - 1 - k], sb[k]);
                        sa[k] = fma(-o, win[s + MK - 1 - k], sa[k]);
                    }
                }
            }
        }
    }
    if (PASS == 0) {
        if (unit) {
#pragma unroll
            for (int j = 0; j < M; ++j) Phi[((size_t)r * M + j) * M + ui] = z[j];
        } else {
#pragma unroll
            for (int j = 0; j < M; ++j) E[((size_t)c * rows + r) * M + j] = z[j];
        }
    } else {
        for (int k = 0; k < K; ++k) {
            atomicAdd(gb + (size_t)r * K + k, sb[k]);
            if (k > 0) atomicAdd(ga + (size_t)r * K + k, sa[k]);
        }
    }
}

// Start states of the chunks from the zero-start end states: 16 lanes per row, lane i = component i of the state.
// dir > 0: S_0 = 0, S_{c+1} = Phi S_c + E_c (forward in time);  dir < 0: S_{P-1} = 0, S_{c-1} = Phi S_c + E_c (the adjoint, backwards)
__global__ __launch_bounds__(64) void lfilt_chain_kernel(const double* __restrict__ Phi, const double* __restrict__ E, double* __restrict__ S,
                                                         int rows, int M, int P, int dir) {
    const int lane = threadIdx.x & 15, r = blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool live = r < rows && lane < M;
    double ph[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) ph[j] = (live && j < M) ? Phi[((size_t)r * M + lane) * M + j] : 0.0;
    double s = 0.0;
    int c = dir > 0 ? 0 : P - 1;
    const int dc = dir > 0 ? 1 : -1;
    if (live) S[((size_t)c * rows + r) * M + lane] = 0.0;
    // The chain is the one sequential part of a filter operation (P - 1 dependent steps per row). The zero-start end states E do not
    // depend on it: eight steps' worth are requested together, so a step costs its arithmetic (15 lane exchanges, two sums of products)
    // and not a round trip to memory as well (round 3: one load inside every step, ~1 us per step - the chain was 0.5 of the 1.2 ms of a
    // forward + backward pass at (16, 1, 262144) and ruled out shorter chunks).
    constexpr int PF = 8;
    for (int step0 = 0; step0 < P - 1; step0 += PF) {
        double e[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            int cu = c + u * dc;
            cu = cu < 0 ? 0 : (cu > P - 1 ? P - 1 : cu);                       // clamped: the tail's extra loads are never used
            e[u] = live ? E[((size_t)cu * rows + r) * M + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (step0 + u < P - 1) {                                           // uniform over the wave
                double v[15];
#pragma unroll
                for (int j = 0; j < 15; ++j) v[j] = __shfl(s, j, 16);         // every exchange before the first product
                double a0 = e[u], a1 = 0.0;
#pragma unroll
                for (int j = 0; j < 14; j += 2) { a0 = fma(ph[j], v[j], a0); a1 = fma(ph[j + 1], v[j + 1], a1); }
                s = fma(ph[14], v[14], a0) + a1;
                c += dc;
                if (live) S[((size_t)c * rows + r) * M + lane] = s;
            }
        }
    }
}

}  // namespace dasp

using namespace dasp;

namespace {
inline int lf_check() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
#ifndef DASP_LFILTER_CHUNK_DEFAULT
#define DASP_LFILTER_CHUNK_DEFAULT 1024
#endif
constexpr long LF_CHUNK_DEFAULT = DASP_LFILTER_CHUNK_DEFAULT;
inline int lf_mk(int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : 16); }
inline long lf_work_doubles(int rows, int P, int M) { return P > 1 ? 2L * P * rows * M + (long)rows * M * M : 0; }
// chunk length: 1024 samples (512 up to 64 rows) from 16 chunks on; shorter signals in 16 chunks of at least 64 samples; one chunk below 128 samples
// chunk > 0: the caller's chunk length (tests: chunk boundaries at odd places). The library reads no environment variable here: the size
// query, the forward and the backward call of one filter operation must cut time the same way, so the caller passes the same value thrice.
inline void lf_plan(int rows, long N, long chunk, long* L, int* P) {
    if (chunk >= 1) { *L = chunk; *P = (int)((N + chunk - 1) / chunk); return; }
    // GPU time of forward + backward against the chunk length (profiles/r04/lfilter_graph_time.log, K = 5, N = 262144): 16 rows 0.85 /
    // 0.68 / 0.81 / 1.30 ms at 256 / 512 / 1024 / 2048 samples per chunk, 256 rows 2.71 / 2.41 / 2.12 / 2.34 ms: few rows want more chunks
    // (threads = rows x chunks), until the sequential chain over the chunks (0.3 us per step and direction) takes over
    long l = rows <= 64 ? LF_CHUNK_DEFAULT / 2 : LF_CHUNK_DEFAULT;
    if (N < 16 * l) { l = (N + 15) / 16; l = (l + 15) / 16 * 16; if (l < 64) l = 64; }
    *L = l; *P = (int)((N + l - 1) / l);
}

#define LF_DISPATCH(KERNEL, PASS_, ...)                                                                                              \
    do {                                                                                                                              \
        if (mk == 4) hipLaunchKernelGGL((KERNEL<4, T, PASS_>), grid, dim3(64), 0, st, __VA_ARGS__);                                   \
        else if (mk == 8) hipLaunchKernelGGL((KERNEL<8, T, PASS_>), grid, dim3(64), 0, st, __VA_ARGS__);                              \
        else hipLaunchKernelGGL((KERNEL<16, T, PASS_>), grid, dim3(64), 0, st, __VA_ARGS__);                                          \
    } while (0)

template <typename T>
int lfilt_forward_t(const T* x, const double* bn, const double* an, T* y, double* wsave, double* work, long work_doubles, int rows, int bcast,
                    long N, int K, long chunk, hipStream_t st) {
    const int mk = lf_mk(K), M = mk - 1;
    long L; int P;
    lf_plan(rows, N, chunk, &L, &P);
    double *E = nullptr, *S = nullptr, *Phi = nullptr;
    if (P > 1) {
        if (!work || work_doubles < lf_work_doubles(rows, P, M)) return DASP_ERR_ARG;
        E = work; S = E + (size_t)P * rows * M; Phi = S + (size_t)P * rows * M;
        dim3 grid((unsigned)(((long)rows * P + (long)rows * M + 63) / 64));
        LF_DISPATCH(lfilt_fwd_kernel, 0, x, bn, an, (T*)nullptr, (double*)nullptr, (const double*)nullptr, E, Phi, rows, bcast, N, K, L, P);
        hipLaunchKernelGGL(lfilt_chain_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, (const double*)Phi, (const double*)E, S, rows, M, P, 1);
    }
    dim3 grid((unsigned)(((long)rows * P + 63) / 64));
    LF_DISPATCH(lfilt_fwd_kernel, 1, x, bn, an, y, wsave, (const double*)S, (double*)nullptr, (double*)nullptr, rows, bcast, N, K, L, P);
    return lf_check();
}
template <typename T>
int lfilt_backward_t(const T* gy, const double* bn, const double* an, const double* wsave, T* gx, double* gb, double* ga, double* work,
                     long work_doubles, int rows, int bcast, long N, int K, long chunk, hipStream_t st) {
    const int mk = lf_mk(K), M = mk - 1;
    long L; int P;
    lf_plan(rows, N, chunk, &L, &P);
    if (P > 1 && (!work || work_doubles < lf_work_doubles(rows, P, M))) return DASP_ERR_ARG;
    hipError_t e = zero_async(gb, sizeof(double) * (size_t)rows * K, st);
    if (e == hipSuccess) e = zero_async(ga, sizeof(double) * (size_t)rows * K, st);
    if (e != hipSuccess) return (int)e;
    double *E = nullptr, *S = nullptr, *Phi = nullptr;
    if (P > 1) {
        E = work; S = E + (size_t)P * rows * M; Phi = S + (size_t)P * rows * M;
        dim3 grid((unsigned)(((long)rows * P + (long)rows * M + 63) / 64));
        LF_DISPATCH(lfilt_bwd_kernel, 0, gy, bn, an, (const double*)nullptr, (T*)nullptr, (double*)nullptr, (double*)nullptr, (const double*)nullptr, E, Phi,
                    rows, bcast, N, K, L, P);
        hipLaunchKernelGGL(lfilt_chain_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, st, (const double*)Phi, (const double*)E, S, rows, M, P, -1);
    }
    dim3 grid((unsigned)(((long)rows * P + 63) / 64));
    LF_DISPATCH(lfilt_bwd_kernel, 1, gy, bn, an, wsave, gx, gb, ga, (const double*)S, (double*)nullptr, (double*)nullptr, rows, bcast, N, K, L, P);
    return lf_check();
}
#undef LF_DISPATCH
}  // namespace

extern "C" {

/* include/dasp_hip.h: filters of K = 4 .. 16 coefficients per row. x, y, gy, gx: (rows, N) float (f64 = 0) or double (f64 = 1);
 * bn, an: (Bs, K) doubles, Bs = rows or 1, normalised so that an[:, 0] = 1 (an FIR filter: an = 1, 0, ...); wsave: (N, rows) doubles or
 * NULL when no backward pass follows; work: dasp_lfilter_work_doubles(rows, N, K) doubles of scratch (chunk states);
 * gb, ga: (rows, K) doubles; gx may be NULL. chunk: samples per chunk of time, 0 = the library's plan - the SAME value must go into the size
 * query, the forward and the backward call; work_doubles: the size of `work`, checked against the plan. */
long dasp_lfilter_work_doubles(int rows, long N, int K, long chunk) {
    if (rows <= 0 || N <= 0 || K < 1 || K > 16) return -1;
    const int M = lf_mk(K) - 1;
    long L; int P;
    lf_plan(rows, N, chunk, &L, &P);
    return lf_work_doubles(rows, P, M);
}
int dasp_lfilter_forward(const void* x, const double* bn, const double* an, int Bs, void* y, double* wsave, double* work, long work_doubles,
                         int rows, long N, int K, int f64, long chunk, void* stream) {
    if (!x || !bn || !an || !y || rows <= 0 || N <= 0 || (Bs != 1 && Bs != rows)) return DASP_ERR_ARG;
    if (K < 1 || K > 16) return DASP_ERR_UNSUPPORTED;
    const int bcast = Bs == 1 && rows > 1;
    return f64 ? lfilt_forward_t<double>((const double*)x, bn, an, (double*)y, wsave, work, work_doubles, rows, bcast, N, K, chunk, (hipStream_t)stream)
               : lfilt_forward_t<float>((const float*)x, bn, an, (float*)y, wsave, work, work_doubles, rows, bcast, N, K, chunk, (hipStream_t)stream);
}
int dasp_lfilter_backward(const void* gy, const double* bn, const double* an, int Bs, const double* wsave, void* gx, double* gb, double* ga,
                          double* work, long work_doubles, int rows, long N, int K, int f64, long chunk, void* stream) {
    if (!gy || !bn || !an || !wsave || !gb || !ga || rows <= 0 || N <= 0 || (Bs != 1 && Bs != rows)) return DASP_ERR_ARG;
    if (K < 1 || K > 16) return DASP_ERR_UNSUPPORTED;
    const int bcast = Bs == 1 && rows > 1;
    return f64 ? lfilt_backward_t<double>((const double*)gy, bn, an, wsave, (double*)gx, gb, ga, work, work_doubles, rows, bcast, N, K, chunk, (hipStream_t)stream)
               : lfilt_backward_t<float>((const float*)gy, bn, an, wsave, (float*)gx, gb, ga, work, work_doubles, rows, bcast, N, K, chunk, (hipStream_t)stream);
}

}  // extern "C"
