// Pieces of the compressor / expander shared by dynamics.hip and chainfwd.hip (the fused EQ -> compressor forward kernel): per-item
// constants, the static gain computer (dasp_pytorch/functional.py:350-369) and the dB constants.
#pragma once
#include "common.hpp"

namespace dasp {

constexpr float DB_PER_LOG2 = 6.020599913279624f;      // 20 / log2(10)
constexpr float LOG2_PER_DB = 0.16609640474436813f;    // log2(10) / 20
constexpr float LN10_20 = 0.11512925464970228f;        // ln(10) / 20
constexpr float DB_SLOPE = 8.685889638065037f;         // 20 / ln(10)

struct DynItem {   // per-item constants, wave-uniform
    float thr, inv_ratio, ratio, knee, makeup, eps;
    float alpha, beta, a4, a8, a16, a32, a256, a1024;
};

// static gain computer: level in dB -> gain in dB (g_c = x_sc - x_db) and, if D, its partial derivatives
template <int MODE, bool D>
__device__ __forceinline__ float gain_computer(float x_db, const DynItem& it, float& d_x, float& d_t, float& d_r, float& d_w) {
    const float half = 0.5f * it.knee, lo = it.thr - half, hi = it.thr + half;
    const bool in_knee = (x_db >= lo) && (x_db <= hi) && (it.knee > 0.f);
    float g = 0.f;
    if (D) { d_x = d_t = d_r = d_w = 0.f; }
    if (MODE == 0) {   // compressor, functional.py:350-369
        const float sl = it.inv_ratio - 1.f;   // 1/R - 1
        if (x_db > hi) {
            g = (x_db - it.thr) * sl;
            if (D) { d_x = sl; d_t = -sl; d_r = -(x_db - it.thr) * it.inv_ratio * it.inv_ratio; }
        } else if (in_knee) {
            const float q = x_db - lo, iw = 1.f / it.knee, h = 0.5f * q * q * iw;
            g = sl * h;
            if (D) { d_x = sl * q * iw; d_t = -d_x; d_r = -h * it.inv_ratio * it.inv_ratio; d_w = sl * (0.5f * q * iw - h * iw); }
        }
    } else {           // downward expander: x_sc = T + (x_db - T) R below the knee
        const float sl = 1.f - it.ratio;       // 1 - R
        if (x_db < lo) {
            g = -(x_db - it.thr) * sl;
            if (D) { d_x = -sl; d_t = sl; d_r = x_db - it.thr; }
        } else if (in_knee) {
            const float q = x_db - hi, iw = 1.f / it.knee, h = 0.5f * q * q * iw;
            g = sl * h;
            if (D) { d_x = sl * q * iw; d_t = -d_x; d_r = -h; d_w = sl * (-0.5f * q * iw - h * iw); }
        }
    }
    return g;
}

// Where the five controls the kernels read live - threshold_db, ratio, attack_ms, knee_db, makeup_gain_db: control i of item b is
// p[i][b * s]. (B, 5) rows (the C ABI's `ctl`): p[i] = ctl + i, s = 5; five separate vectors of B values (the reference's own
// signature, functional.py:275-286 - no stacking launch in front of the kernels): s = 1. Passed to the kernels by value.
struct DynCtl {
    const float* p[5];
    int s;
    __host__ __device__ float at(int i, int b) const { return p[i][(size_t)b * s]; }
};
inline DynCtl dyn_ctl_rows(const float* ctl) { return DynCtl{{ctl, ctl + 1, ctl + 2, ctl + 3, ctl + 4}, 5}; }
// ... and where the five control gradients go (same order); `zero` (may be null): B more values that are set to zero - the gradient of
// release_ms, which has no path to the output (functional.py:340,343-344)
struct DynGrad {
    float* p[5];
    int s;
    float* zero;
};
inline DynGrad dyn_grad_rows(float* gctl) { return DynGrad{{gctl, gctl + 1, gctl + 2, gctl + 3, gctl + 4}, 5, nullptr}; }

__device__ __forceinline__ DynItem load_item(const DynCtl& ctl, int b, double sample_rate, float eps) {
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ctl.at(i, b);
    DynItem it;
    it.thr = c[0]; it.ratio = c[1]; it.inv_ratio = 1.f / c[1]; it.knee = c[3]; it.makeup = c[4]; it.eps = eps;
    const double nat = sample_rate * ((double)c[2] / 1e3);                    // functional.py:339
    const double a = exp(-2.1972245773362196 / nat);                          // :341-342, ln 9
    const double a2 = a * a, a4 = a2 * a2, a8 = a4 * a4, a16 = a8 * a8, a32 = a16 * a16, a64 = a32 * a32, a128 = a64 * a64,
                 a256 = a128 * a128, a512 = a256 * a256;
    it.alpha = (float)a; it.beta = (float)(1.0 - a);
    it.a4 = (float)a4; it.a8 = (float)a8; it.a16 = (float)a16; it.a32 = (float)a32; it.a256 = (float)a256; it.a1024 = (float)(a512 * a512);
    return it;
}
__device__ __forceinline__ DynItem load_item(const float* __restrict__ ctl, int b, double sample_rate, float eps) {      // (B, 5) rows
    return load_item(DynCtl{{ctl, ctl + 1, ctl + 2, ctl + 3, ctl + 4}, 5}, b, sample_rate, eps);
}

}  // namespace dasp
