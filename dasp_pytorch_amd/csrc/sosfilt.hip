// Cascaded-biquad IIR (second-order sections) forward + hand-derived adjoint for gfx950.
//
// Replaces the reference's frequency-sampling filter `sosfilt_via_fsm`
// (dasp_pytorch/signal.py:136-166, called from functional.parametric_eq, functional.py:267)
// and the autograd graph behind it with a true recurrence evaluated as a chunked parallel scan.
// Executable fp64 specification of exactly this algorithm: oracle/chunkscan_model.py.
//
// Work decomposition
//   row   = one (batch item, channel) signal of N samples           -> one workgroup of W waves
//   tile  = 64*L consecutive samples of a row                       -> one wave, tiles round-robin
//   chunk = L consecutive samples of a tile                         -> one lane
// Each biquad is realised in a *normal* state-space form (rotation/symmetric 2x2 state matrix,
// see prep kernel) which is ~1000x less noisy in fp32 than direct forms for low-frequency poles.
// Per tile: (1) coalesced float4 loads -> LDS transpose -> L samples per lane;
// (2) z = G x : zero-state end state of every chunk (table G from the prep kernel, SGPR operands);
// (3) per section k: forcing f = z_k + sum_{j<k} M_kj s_j (block-lower-triangular coupling), then a
//     6-level Kogge-Stone scan over lanes with the 2x2 powers (M_kk)^(2^l);
// (4) the tile carry K_k is handed from the wave that owns tile t-1 through an LDS mailbox; only
//     K' = e_63 + M_kk^64 K sits on that serial chain, the per-lane fix-up M_kk^(lane+1) K is off it;
// (5) every lane runs the 6-section cascade over its L samples from its exact start state;
// (6) LDS transpose back -> coalesced float4 stores.
// The backward kernel walks the tiles in reverse: it recomputes the forward chunk states from the
// per-tile carries the forward pass saved, keeps s2_k[n] (= om_k * w_k[n-2], the all-pole signal)
// in registers, runs the adjoint cascade (sections reversed, transposed realisation) and accumulates
// the five coefficient correlations per section; a finalize kernel reduces them in fp64.
#include "common.hpp"
#include <type_traits>

namespace dasp {

// ------------------------------------------------------------------------------------------------
// Per-item fp32 table layout (floats). S sections, chunk length L.
template <int S, int L>
struct SosLayout {
    static constexpr int S2 = 2 * S;
    static constexpr int COEF = 0;                  // [S][8]: sg, om, kom, g1, g2, d, kappa, pad
    static constexpr int G = COEF + S * 8;          // [2S][L]  forward chunk table
    static constexpr int M = G + S2 * L;            // [2S][2S] Phi^L (block lower triangular)
    static constexpr int P = M + S2 * S2;           // [S][7][4]: (p, q, kappa*q, pad) of M_kk^(2^l)
    static constexpr int PW = P + S * 7 * 4;        // [S][2][64]: (p, q) of M_kk^(lane+1)
    static constexpr int GA = PW + S * 2 * 64;      // [2S][L]  adjoint chunk table, natural n order
    static constexpr int MA = GA + S2 * L;          // [2S][2S] adjoint Phi^L (adjoint section order)
    static constexpr int TOTAL = MA + S2 * S2;
};
// fp64 side table for the finalize kernel, per (item, section)
constexpr int DT_OM = 0, DT_B0 = 1, DT_A1 = 4, DT_A0 = 6, DT_J = 8, DT_STRIDE = 24;

// (p, q) x (p', q') for matrices [[p, -k q], [q, p]] (closed under multiplication for fixed k)
__device__ __forceinline__ void nmul(double k, double p1, double q1, double p2, double q2, double& p, double& q) {
    p = p1 * p2 - k * q1 * q2;
    q = p1 * q2 + q1 * p2;
}

// ---- forward-mode dual numbers (3 partials) for the RBJ design Jacobian ---------------------------
struct D3 {
    double v, d[3];
};
__device__ __forceinline__ D3 dconst(double c) { return {c, {0, 0, 0}}; }
__device__ __forceinline__ D3 dvar(double c, int i) { D3 r = {c, {0, 0, 0}}; r.d[i] = 1; return r; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
    return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
    const double iv = 1.0 / b.v, q = a.v * iv;
    return {q, {(a.d[0] - q * b.d[0]) * iv, (a.d[1] - q * b.d[1]) * iv, (a.d[2] - q * b.d[2]) * iv}};
}
__device__ __forceinline__ D3 operator*(double c, D3 a) { return {c * a.v, {c * a.d[0], c * a.d[1], c * a.d[2]}}; }
__device__ __forceinline__ D3 operator+(double c, D3 a) { return {c + a.v, {a.d[0], a.d[1], a.d[2]}}; }
__device__ __forceinline__ D3 operator-(double c, D3 a) { return {c - a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a) { return {-a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
__device__ __forceinline__ D3 dchain(D3 a, double f, double fp) { return {f, {fp * a.d[0], fp * a.d[1], fp * a.d[2]}}; }
__device__ __forceinline__ D3 dsin(D3 a) { return dchain(a, sin(a.v), cos(a.v)); }
__device__ __forceinline__ D3 dcos(D3 a) { return dchain(a, cos(a.v), -sin(a.v)); }
__device__ __forceinline__ D3 dsqrt(D3 a) { const double s = sqrt(a.v); return dchain(a, s, 0.5 / s); }
__device__ __forceinline__ D3 dexp(D3 a) { const double e = exp(a.v); return dchain(a, e, e); }

struct PeqSpec {
    int types[8];        // 0 peaking, 1 low_shelf, 2 high_shelf, 3 low_pass, 4 high_pass
    double sample_rate;
};

// RBJ cookbook design, same formulas as dasp_pytorch/signal.py:255-304, in fp64 with the Jacobian
// d(b0,b1,b2,a1,a2 normalised)/d(gain_db, cutoff_freq, q_factor).
__device__ void rbj_design(int type, double sample_rate, double gain_db, double fc, double qf, double* c5, double* J /*[5][3]*/) {
    const D3 g = dvar(gain_db, 0), f = dvar(fc, 1), q = dvar(qf, 2);
    const D3 A = dexp((2.302585092994045684 / 40.0) * g);
    const D3 w0 = (2.0 * 3.14159265358979323846 / sample_rate) * f;
    const D3 alpha = dsin(w0) / (2.0 * q);
    const D3 cw = dcos(w0);
    const D3 sA = dsqrt(A);
    D3 b0, b1, b2, a0, a1, a2;
    if (type == 2) {  // high_shelf
        b0 = A * ((A + dconst(1)) + (A - dconst(1)) * cw + 2.0 * (sA * alpha));
        b1 = -2.0 * (A * ((A - dconst(1)) + (A + dconst(1)) * cw));
        b2 = A * ((A + dconst(1)) + (A - dconst(1)) * cw - 2.0 * (sA * alpha));
        a0 = (A + dconst(1)) - (A - dconst(1)) * cw + 2.0 * (sA * alpha);
        a1 = 2.0 * ((A - dconst(1)) - (A + dconst(1)) * cw);
        a2 = (A + dconst(1)) - (A - dconst(1)) * cw - 2.0 * (sA * alpha);
    } else if (type == 1) {  // low_shelf
        b0 = A * ((A + dconst(1)) - (A - dconst(1)) * cw + 2.0 * (sA * alpha));
        b1 = 2.0 * (A * ((A - dconst(1)) - (A + dconst(1)) * cw));
        b2 = A * ((A + dconst(1)) - (A - dconst(1)) * cw - 2.0 * (sA * alpha));
        a0 = (A + dconst(1)) + (A - dconst(1)) * cw + 2.0 * (sA * alpha);
        a1 = -2.0 * ((A - dconst(1)) + (A + dconst(1)) * cw);
        a2 = (A + dconst(1)) + (A - dconst(1)) * cw - 2.0 * (sA * alpha);
    } else if (type == 0) {  // peaking
        b0 = 1.0 + alpha * A;
        b1 = -2.0 * cw;
        b2 = 1.0 - alpha * A;
        a0 = 1.0 + alpha / A;
        a1 = -2.0 * cw;
        a2 = 1.0 - alpha / A;
    } else if (type == 3) {  // low_pass
        b0 = 0.5 * (1.0 - cw);
        b1 = 1.0 - cw;
        b2 = 0.5 * (1.0 - cw);
        a0 = 1.0 + alpha;
        a1 = -2.0 * cw;
        a2 = 1.0 - alpha;
    } else {  // high_pass
        b0 = 0.5 * (1.0 + cw);
        b1 = -(1.0 + cw);
        b2 = 0.5 * (1.0 + cw);
        a0 = 1.0 + alpha;
        a1 = -2.0 * cw;
        a2 = 1.0 - alpha;
    }
    const D3 n[5] = {b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
    for (int c = 0; c < 5; ++c) {
        c5[c] = n[c].v;
        for (int i = 0; i < 3; ++i) J[c * 3 + i] = n[c].d[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Prep kernel: one workgroup per batch item. Builds the realisation and all chunk tables in fp64.
constexpr double OM_MIN = 1e-5;

template <int S, int L>
__global__ void __launch_bounds__(256)
sos_prep_kernel(const float* __restrict__ sos, const float* __restrict__ params, PeqSpec spec,
                float* __restrict__ tab, double* __restrict__ dtab) {
    using LY = SosLayout<S, L>;
    constexpr int S2 = 2 * S, NN = S2 * S2;
    __shared__ double sec[S][8];        // sg, om, kom, g1, g2, d, kappa
    __shared__ double Phi[2][NN], T1[2][NN], T2[2][NN];
    __shared__ double vv[2][2][S2];
    __shared__ double Pd[S][7][2];
    const int tid = threadIdx.x, item = blockIdx.x;
    float* tb = tab + (size_t)item * LY::TOTAL;
    double* dt = dtab + (size_t)item * S * DT_STRIDE;

    if (tid < S) {
        const int k = tid;
        double c5[5], a0 = 1.0, J[15];
        for (int i = 0; i < 15; ++i) J[i] = 0.0;
        if (params) {
            const float* p = params + ((size_t)item * S + k) * 3;
            rbj_design(spec.types[k], spec.sample_rate, (double)p[0], (double)p[1], (double)p[2], c5, J);
        } else {
            const float* s = sos + ((size_t)item * S + k) * 6;
            a0 = (double)s[3];
            c5[0] = s[0] / a0; c5[1] = s[1] / a0; c5[2] = s[2] / a0; c5[3] = s[4] / a0; c5[4] = s[5] / a0;
        }
        const double b0 = c5[0], b1 = c5[1], b2 = c5[2], a1 = c5[3], a2 = c5[4];
        const double sg = -0.5 * a1, disc = sg * sg - a2;
        const double kap = disc < 0 ? 1.0 : -1.0;
        double om = sqrt(fabs(disc));
        om = om < OM_MIN ? OM_MIN : om;
        const double g1 = b1 - b0 * a1, g2 = ((b2 - b0 * a2) + g1 * sg) / om;
        sec[k][0] = sg; sec[k][1] = om; sec[k][2] = kap * om; sec[k][3] = g1; sec[k][4] = g2; sec[k][5] = b0; sec[k][6] = kap;
        float* cf = tb + LY::COEF + k * 8;
        cf[0] = (float)sg; cf[1] = (float)om; cf[2] = (float)(kap * om); cf[3] = (float)g1; cf[4] = (float)g2;
        cf[5] = (float)b0; cf[6] = (float)kap; cf[7] = 0.f;
        double* d = dt + k * DT_STRIDE;
        d[DT_OM] = om;
        for (int c = 0; c < 5; ++c) d[DT_B0 + c] = c5[c];
        d[DT_A0] = a0; d[7] = kap;
        for (int i = 0; i < 15; ++i) d[DT_J + i] = J[i];
        d[23] = 0.0;
    }
    __syncthreads();

    // Phi for the forward system (sys 0) and the adjoint system (sys 1: sections reversed, A^T, B<->C)
    for (int e = tid; e < 2 * NN; e += 256) {
        const int sys = e / NN, i = (e % NN) / S2, j = e % S2;
        const int kk = i / 2, r = i % 2, jj = j / 2, c = j % 2;
        const int fk = sys ? S - 1 - kk : kk, fj = sys ? S - 1 - jj : jj;  // forward section ids
        double v = 0.0;
        if (jj == kk) {
            const double sg = sec[fk][0], om = sec[fk][1], kom = sec[fk][2];
            const double A[2][2] = {{sg, -kom}, {om, sg}};
            v = sys ? A[c][r] : A[r][c];
        } else if (jj < kk) {
            const double Bk = sys ? sec[fk][3 + r] : (r == 0 ? 1.0 : 0.0);   // B of section at position kk
            const double Cj = sys ? (c == 0 ? 1.0 : 0.0) : sec[fj][3 + c];   // C of section at position jj
            double gain = 1.0;
            for (int m = jj + 1; m < kk; ++m) gain *= sec[sys ? S - 1 - m : m][5];
            v = Bk * gain * Cj;
        }
        Phi[sys][i * S2 + j] = v;
        T1[sys][i * S2 + j] = v;
    }
    if (tid < 2 * S2) {
        const int sys = tid / S2, i = tid % S2, kk = i / 2, r = i % 2;
        const int fk = sys ? S - 1 - kk : kk;
        double gain = 1.0;
        for (int m = 0; m < kk; ++m) gain *= sec[sys ? S - 1 - m : m][5];
        vv[0][sys][i] = (sys ? sec[fk][3 + r] : (r == 0 ? 1.0 : 0.0)) * gain;
    }
    __syncthreads();

    // G tables: v_m = Phi^m Bx ; forward G[:, L-1-m] = v_m ; adjoint (natural order) Ga[:, m] = v_m
    for (int m = 0; m < L; ++m) {
        const int cur = m & 1;
        if (tid < 2 * S2) {
            const int sys = tid / S2, i = tid % S2;
            const double v = vv[cur][sys][i];
            if (sys == 0) tb[LY::G + i * L + (L - 1 - m)] = (float)v;
            else tb[LY::GA + i * L + m] = (float)v;
            double acc = 0.0;
            for (int j = 0; j < S2; ++j) acc += Phi[sys][i * S2 + j] * vv[cur][sys][j];
            vv[cur ^ 1][sys][i] = acc;
        }
        __syncthreads();
    }

    // M = Phi^L by repeated squaring (L is a power of two)
    {
        double (*src)[NN] = T1;
        double (*dst)[NN] = T2;
        for (int step = 1; step < L; step <<= 1) {
            for (int e = tid; e < 2 * NN; e += 256) {
                const int sys = e / NN, i = (e % NN) / S2, j = e % S2;
                double acc = 0.0;
                for (int m = 0; m < S2; ++m) acc += src[sys][i * S2 + m] * src[sys][m * S2 + j];
                dst[sys][i * S2 + j] = acc;
            }
            __syncthreads();
            double (*tmp)[NN] = src; src = dst; dst = tmp;
        }
        for (int e = tid; e < 2 * NN; e += 256) {
            const int sys = e / NN, ij = e % NN;
            tb[(sys ? LY::MA : LY::M) + ij] = (float)src[sys][ij];
        }
        if (tid < S) {  // diagonal-block powers (p, q): M_kk = [[p, -kap q], [q, p]]
            const int k = tid;
            const double kap = sec[k][6];
            double p = src[0][(2 * k) * S2 + 2 * k], q = src[0][(2 * k + 1) * S2 + 2 * k];
            for (int l = 0; l < 7; ++l) {
                Pd[k][l][0] = p; Pd[k][l][1] = q;
                float* o = tb + LY::P + (k * 7 + l) * 4;
                o[0] = (float)p; o[1] = (float)q; o[2] = (float)(kap * q); o[3] = 0.f;
                double p2, q2;
                nmul(kap, p, q, p, q, p2, q2);
                p = p2; q = q2;
            }
        }
    }
    __syncthreads();
    // per-lane powers M_kk^(c+1), c = 0..63
    for (int e = tid; e < S * 64; e += 256) {
        const int k = e / 64, c = e % 64, m = c + 1;
        const double kap = sec[k][6];
        double p = 1.0, q = 0.0;
        for (int l = 0; l < 7; ++l)
            if (m & (1 << l)) {
                double p2, q2;
                nmul(kap, p, q, Pd[k][l][0], Pd[k][l][1], p2, q2);
                p = p2; q = q2;
            }
        tb[LY::PW + (2 * k) * 64 + c] = (float)p;
        tb[LY::PW + (2 * k + 1) * 64 + c] = (float)q;
    }
}

// ------------------------------------------------------------------------------------------------
// shared pieces of the forward / backward tile code

// z[j] = sum_n T[j][n] * X[n]   (T wave-uniform -> scalar loads, SGPR operands)
template <int S2, int L>
__device__ __forceinline__ void table_apply(const float* __restrict__ T, const float (&X)[L], float (&z)[S2]) {
#pragma unroll
    for (int j = 0; j < S2; ++j) {
        float a = 0.f;
#pragma unroll
        for (int n = 0; n < L; ++n) a = fmaf(T[j * L + n], X[n], a);
        z[j] = a;
    }
}

// ------------------------------------------------------------------------------------------------
template <int S, int L, int W>
__global__ void __launch_bounds__(64 * W)
sos_fwd_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ x, float* __restrict__ y,
               float* __restrict__ carries, int C, int N, int nt, int vec) {
    using LY = SosLayout<S, L>;
    constexpr int S2 = 2 * S, TS = 64 * L, LP = L + 4;
    __shared__ __attribute__((aligned(16))) float lds[W * 64 * LP + W * S * 4];
    const int lane = lane_id(), wave = wave_id();
    const int row = blockIdx.x;
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : row / C) * LY::TOTAL;
    const float* __restrict__ xr = x + (size_t)row * N;
    float* __restrict__ yr = y + (size_t)row * N;
    float* tbuf = lds + wave * 64 * LP;
    volatile float* mb_in = lds + W * 64 * LP + wave * S * 4;
    volatile float* mb_out = lds + W * 64 * LP + ((wave + 1) % W) * S * 4;
    if (W > 1) {
        for (int i = threadIdx.x; i < W * S * 4; i += 64 * W) lds[W * 64 * LP + i] = 0.f;
        __syncthreads();
    }
    float pwp[S], pwq[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        pwp[k] = tb[LY::PW + (2 * k) * 64 + lane];
        pwq[k] = tb[LY::PW + (2 * k + 1) * 64 + lane];
    }
    float Kreg[S2];
#pragma unroll
    for (int j = 0; j < S2; ++j) Kreg[j] = 0.f;

    float4 cur[L / 4], nxt[L / 4];
    int t = wave;
    if (t < nt) tile_load_global<L>(xr, (long)t * TS, N, vec, cur);
    for (; t < nt; t += W) {
        if (t + W < nt) tile_load_global<L>(xr, (long)(t + W) * TS, N, vec, nxt);
        float X[L];
        tile_to_chunks<L>(tbuf, cur, X);

        float z[S2];
        table_apply<S2, L>(tb + LY::G, X, z);

        float st[S2];  // exact state at the start of this lane's chunk
#pragma unroll
        for (int k = 0; k < S; ++k) {
            float f1 = z[2 * k], f2 = z[2 * k + 1];
#pragma unroll
            for (int j = 0; j < k; ++j) {
                const float* m = tb + LY::M + (2 * k) * S2 + 2 * j;
                f1 = fmaf(m[0], st[2 * j], fmaf(m[1], st[2 * j + 1], f1));
                f2 = fmaf(m[S2], st[2 * j], fmaf(m[S2 + 1], st[2 * j + 1], f2));
            }
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                const float* pp = tb + LY::P + (k * 7 + l) * 4;
                const float t1 = shift_up(f1, 1 << l), t2 = shift_up(f2, 1 << l);
                if (lane >= (1 << l)) {
                    f1 += pp[0] * t1 - pp[2] * t2;
                    f2 += pp[1] * t1 + pp[0] * t2;
                }
            }
            float K1, K2;
            if (W == 1) {
                K1 = Kreg[2 * k]; K2 = Kreg[2 * k + 1];
            } else if (t == 0) {
                K1 = 0.f; K2 = 0.f;
            } else {
                mbox_wait(mb_in + 4 * k, t, K1, K2);
            }
            {   // carry for the next tile: the only work on the cross-wave serial chain
                const float* p64 = tb + LY::P + (k * 7 + 6) * 4;
                const float e1 = read_lane(f1, 63), e2 = read_lane(f2, 63);
                const float n1 = e1 + p64[0] * K1 - p64[2] * K2;
                const float n2 = e2 + p64[1] * K1 + p64[0] * K2;
                if (W == 1) { Kreg[2 * k] = n1; Kreg[2 * k + 1] = n2; }
                else if (t + 1 < nt) mbox_publish(mb_out + 4 * k, n1, n2, t + 1);
            }
            if (carries && lane == 0) {
                float* cs = carries + ((size_t)row * nt + t) * S2 + 2 * k;
                cs[0] = K1; cs[1] = K2;
            }
            const float kap = tb[LY::COEF + k * 8 + 6];
            const float E1 = f1 + pwp[k] * K1 - kap * pwq[k] * K2;
            const float E2 = f2 + pwq[k] * K1 + pwp[k] * K2;
            const float s1 = shift_up(E1, 1), s2 = shift_up(E2, 1);
            st[2 * k] = lane == 0 ? K1 : s1;
            st[2 * k + 1] = lane == 0 ? K2 : s2;
        }

#pragma unroll
        for (int n = 0; n < L; ++n) {
            float u = X[n];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float* cf = tb + LY::COEF + k * 8;
                const float s1 = st[2 * k], s2 = st[2 * k + 1];
                const float yv = fmaf(cf[3], s1, fmaf(cf[4], s2, cf[5] * u));
                st[2 * k] = fmaf(cf[0], s1, fmaf(-cf[2], s2, u));
                st[2 * k + 1] = fmaf(cf[1], s1, cf[0] * s2);
                u = yv;
            }
            X[n] = u;
        }

        float4 out[L / 4];
        chunks_to_tile<L>(tbuf, X, out);
        tile_store_global<L>(yr, (long)t * TS, N, vec, out);
#pragma unroll
        for (int j = 0; j < L / 4; ++j) cur[j] = nxt[j];
    }
}

// ------------------------------------------------------------------------------------------------
template <int S, int L, int W>
__global__ void __launch_bounds__(64 * W)
sos_bwd_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ x,
               const float* __restrict__ gy, const float* __restrict__ carries, float* __restrict__ gx,
               float* __restrict__ partials, int C, int N, int nt, int vec) {
    using LY = SosLayout<S, L>;
    constexpr int S2 = 2 * S, TS = 64 * L, LP = L + 4;
    __shared__ __attribute__((aligned(16))) float lds[W * 64 * LP + W * S * 4];
    const int lane = lane_id(), wave = wave_id();
    const int row = blockIdx.x;
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : row / C) * LY::TOTAL;
    const float* __restrict__ xr = x + (size_t)row * N;
    const float* __restrict__ gr = gy + (size_t)row * N;
    float* __restrict__ gxr = gx + (size_t)row * N;
    float* tbuf = lds + wave * 64 * LP;
    volatile float* mb_in = lds + W * 64 * LP + wave * S * 4;
    volatile float* mb_out = lds + W * 64 * LP + ((wave + 1) % W) * S * 4;
    if (W > 1) {
        for (int i = threadIdx.x; i < W * S * 4; i += 64 * W) lds[W * 64 * LP + i] = 0.f;
        __syncthreads();
    }
    // (M_kk^T)^(64 - lane): same (p, q) as M_kk^(64 - lane), applied transposed
    float pwp[S], pwq[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        pwp[k] = tb[LY::PW + (2 * k) * 64 + (63 - lane)];
        pwq[k] = tb[LY::PW + (2 * k + 1) * 64 + (63 - lane)];
    }
    float Kreg[S2];
#pragma unroll
    for (int j = 0; j < S2; ++j) Kreg[j] = 0.f;
    float accb[S][3], acca[S][2];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        accb[k][0] = accb[k][1] = accb[k][2] = 0.f;
        acca[k][0] = acca[k][1] = 0.f;
    }

    for (int r = wave; r < nt; r += W) {
        const int t = nt - 1 - r;
        float X[L], GY[L];
        {
            float4 v[L / 4];
            tile_load_global<L>(xr, (long)t * TS, N, vec, v);
            tile_to_chunks<L>(tbuf, v, X);
            tile_load_global<L>(gr, (long)t * TS, N, vec, v);
            tile_to_chunks<L>(tbuf, v, GY);
        }
        // ---- forward chunk start states from the carry saved by the forward pass ----
        float st[S2];
        {
            float z[S2];
            table_apply<S2, L>(tb + LY::G, X, z);
            const float* __restrict__ cs = carries + ((size_t)row * nt + t) * S2;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                float f1 = z[2 * k], f2 = z[2 * k + 1];
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    const float* m = tb + LY::M + (2 * k) * S2 + 2 * j;
                    f1 = fmaf(m[0], st[2 * j], fmaf(m[1], st[2 * j + 1], f1));
                    f2 = fmaf(m[S2], st[2 * j], fmaf(m[S2 + 1], st[2 * j + 1], f2));
                }
                const float K1 = cs[2 * k], K2 = cs[2 * k + 1];
                if (lane == 0) {  // fold the known carry into lane 0's forcing (no chain here)
                    const float* p0 = tb + LY::P + (k * 7 + 0) * 4;
                    f1 += p0[0] * K1 - p0[2] * K2;
                    f2 += p0[1] * K1 + p0[0] * K2;
                }
#pragma unroll
                for (int l = 0; l < 6; ++l) {
                    const float* pp = tb + LY::P + (k * 7 + l) * 4;
                    const float t1 = shift_up(f1, 1 << l), t2 = shift_up(f2, 1 << l);
                    if (lane >= (1 << l)) {
                        f1 += pp[0] * t1 - pp[2] * t2;
                        f2 += pp[1] * t1 + pp[0] * t2;
                    }
                }
                const float s1 = shift_up(f1, 1), s2 = shift_up(f2, 1);
                st[2 * k] = lane == 0 ? K1 : s1;
                st[2 * k + 1] = lane == 0 ? K2 : s2;
            }
        }
        // ---- adjoint chunk end states (scan runs from lane 63 down to lane 0) ----
        float lam[S2];  // adjoint section order: i <-> forward section S-1-i
        {
            float z[S2];
            table_apply<S2, L>(tb + LY::GA, GY, z);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const int k = S - 1 - i;
                float f1 = z[2 * i], f2 = z[2 * i + 1];
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    const float* m = tb + LY::MA + (2 * i) * S2 + 2 * j;
                    f1 = fmaf(m[0], lam[2 * j], fmaf(m[1], lam[2 * j + 1], f1));
                    f2 = fmaf(m[S2], lam[2 * j], fmaf(m[S2 + 1], lam[2 * j + 1], f2));
                }
#pragma unroll
                for (int l = 0; l < 6; ++l) {
                    const float* pp = tb + LY::P + (k * 7 + l) * 4;   // transposed: [[p, q], [-kq, p]]
                    const float t1 = shift_down(f1, 1 << l), t2 = shift_down(f2, 1 << l);
                    if (lane + (1 << l) < 64) {
                        f1 += pp[0] * t1 + pp[1] * t2;
                        f2 += pp[0] * t2 - pp[2] * t1;
                    }
                }
                float K1, K2;
                if (W == 1) {
                    K1 = Kreg[2 * i]; K2 = Kreg[2 * i + 1];
                } else if (r == 0) {
                    K1 = 0.f; K2 = 0.f;
                } else {
                    mbox_wait(mb_in + 4 * i, t + 1, K1, K2);
                }
                {
                    const float* p64 = tb + LY::P + (k * 7 + 6) * 4;
                    const float e1 = read_lane(f1, 0), e2 = read_lane(f2, 0);
                    const float n1 = e1 + p64[0] * K1 + p64[1] * K2;
                    const float n2 = e2 + p64[0] * K2 - p64[2] * K1;
                    if (W == 1) { Kreg[2 * i] = n1; Kreg[2 * i + 1] = n2; }
                    else if (t > 0) mbox_publish(mb_out + 4 * i, n1, n2, t);
                }
                const float kap = tb[LY::COEF + k * 8 + 6];
                const float E1 = f1 + pwp[k] * K1 + pwq[k] * K2;
                const float E2 = f2 + pwp[k] * K2 - kap * pwq[k] * K1;
                const float s1 = shift_down(E1, 1), s2 = shift_down(E2, 1);
                lam[2 * i] = lane == 63 ? K1 : s1;
                lam[2 * i + 1] = lane == 63 ? K2 : s2;
            }
        }
        // ---- forward cascade over the chunk, keeping s2_k[n] (n = 0..L+1) ----
        float S2v[S][L + 2];
#pragma unroll
        for (int n = 0; n < L; ++n) {
            float u = X[n];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float* cf = tb + LY::COEF + k * 8;
                const float s1 = st[2 * k], s2 = st[2 * k + 1];
                S2v[k][n] = s2;
                const float yv = fmaf(cf[3], s1, fmaf(cf[4], s2, cf[5] * u));
                st[2 * k] = fmaf(cf[0], s1, fmaf(-cf[2], s2, u));
                st[2 * k + 1] = fmaf(cf[1], s1, cf[0] * s2);
                u = yv;
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const float* cf = tb + LY::COEF + k * 8;
            S2v[k][L] = st[2 * k + 1];
            S2v[k][L + 1] = fmaf(cf[1], st[2 * k], cf[0] * st[2 * k + 1]);  // s2 does not see the input
        }
        // ---- adjoint cascade (descending n) + coefficient correlations ----
#pragma unroll
        for (int n = L - 1; n >= 0; --n) {
            float g = GY[n];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const int k = S - 1 - i;
                const float* cf = tb + LY::COEF + k * 8;
                const float l1 = lam[2 * i], l2 = lam[2 * i + 1];
                accb[k][0] = fmaf(g, S2v[k][n + 2], accb[k][0]);
                accb[k][1] = fmaf(g, S2v[k][n + 1], accb[k][1]);
                accb[k][2] = fmaf(g, S2v[k][n], accb[k][2]);
                const float o = fmaf(cf[5], g, l1);
                lam[2 * i] = fmaf(cf[0], l1, fmaf(cf[1], l2, cf[3] * g));
                lam[2 * i + 1] = fmaf(-cf[2], l1, fmaf(cf[0], l2, cf[4] * g));
                acca[k][0] = fmaf(o, S2v[k][n + 1], acca[k][0]);
                acca[k][1] = fmaf(o, S2v[k][n], acca[k][1]);
                g = o;
            }
            GY[n] = g;
        }
        {
            float4 out[L / 4];
            chunks_to_tile<L>(tbuf, GY, out);
            tile_store_global<L>(gxr, (long)t * TS, N, vec, out);
        }
    }
    // per-wave partial sums -> partials[row][wave][S][5]
    float* po = partials + ((size_t)row * W + wave) * S * 5;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const float v0 = wave_sum(accb[k][0]), v1 = wave_sum(accb[k][1]), v2 = wave_sum(accb[k][2]);
        const float v3 = wave_sum(acca[k][0]), v4 = wave_sum(acca[k][1]);
        if (lane == 0) {
            po[k * 5 + 0] = v0; po[k * 5 + 1] = v1; po[k * 5 + 2] = v2; po[k * 5 + 3] = v3; po[k * 5 + 4] = v4;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Finalize: one thread per (item, section). mode 0: gradient w.r.t. sos (B,S,6) as given (a0 included);
// mode 1: gradient w.r.t. (gain_db, cutoff_freq, q_factor) (B,S,3) through the RBJ design Jacobian.
__global__ void sos_finalize_kernel(const double* __restrict__ dtab, int tab_bcast, const float* __restrict__ partials,
                                    int B, int C, int S, int Wb, int mode, float* __restrict__ gout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * S) return;
    const int item = idx / S, k = idx % S;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < C; ++c)
        for (int w = 0; w < Wb; ++w) {
            const float* p = partials + (((size_t)(item * C + c) * Wb + w) * S + k) * 5;
            for (int i = 0; i < 5; ++i) acc[i] += (double)p[i];
        }
    const double* d = dtab + ((size_t)(tab_bcast ? 0 : item) * S + k) * DT_STRIDE;
    const double iom = 1.0 / d[DT_OM];
    const double g5[5] = {acc[0] * iom, acc[1] * iom, acc[2] * iom, -acc[3] * iom, -acc[4] * iom};
    if (mode == 0) {
        const double a0 = d[DT_A0];
        double dot = 0.0;
        for (int i = 0; i < 5; ++i) dot += g5[i] * d[DT_B0 + i];
        float* o = gout + (size_t)idx * 6;
        o[0] = (float)(g5[0] / a0); o[1] = (float)(g5[1] / a0); o[2] = (float)(g5[2] / a0);
        o[3] = (float)(-dot / a0);
        o[4] = (float)(g5[3] / a0); o[5] = (float)(g5[4] / a0);
    } else {
        float* o = gout + (size_t)idx * 3;
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
            for (int c = 0; c < 5; ++c) s += g5[c] * d[DT_J + c * 3 + i];
            o[i] = (float)s;
        }
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
constexpr int kL = 16;    // samples per lane chunk
constexpr int kWF = 8;    // waves per row, forward
constexpr int kWB = 4;    // waves per row, backward

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename F>
int dispatch_S(int S, F&& f) {
    switch (S) {
        case 2: return f(std::integral_constant<int, 2>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 6: return f(std::integral_constant<int, 6>{});
        case 8: return f(std::integral_constant<int, 8>{});
        default: return DASP_ERR_UNSUPPORTED;
    }
}
}  // namespace

extern "C" {

int dasp_sos_chunk(void) { return kL; }
int dasp_sos_tile(void) { return 64 * kL; }
int dasp_sos_bwd_waves(void) { return kWB; }
int dasp_sos_supported_sections(int S) { return S == 2 || S == 4 || S == 6 || S == 8; }

long dasp_sos_table_floats(int S) {
    long r = -1;
    dispatch_S(S, [&](auto s) { r = SosLayout<decltype(s)::value, kL>::TOTAL; return 0; });
    return r;
}
long dasp_sos_dtab_doubles(int S) { return (long)S * DT_STRIDE; }
long dasp_sos_num_tiles(long N) { return (N + 64 * kL - 1) / (64 * kL); }
long dasp_sos_carry_floats(long rows, long N, int S) { return rows * dasp_sos_num_tiles(N) * 2 * S; }
long dasp_sos_partial_floats(long rows, int S) { return rows * kWB * S * 5; }

// sos: (Bs, S, 6) fp32 rows [b0 b1 b2 a0 a1 a2] (signal.py:141). Builds tables for Bs items.
int dasp_sos_prepare(const float* sos, int Bs, int S, float* tab, double* dtab, void* stream) {
    if (!sos || !tab || !dtab || Bs <= 0) return DASP_ERR_ARG;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        PeqSpec spec = {};
        hipLaunchKernelGGL((sos_prep_kernel<SS, kL>), dim3(Bs), dim3(256), 0, (hipStream_t)stream, sos, nullptr, spec, tab, dtab);
        return check_launch();
    });
}

// params: (Bs, S, 3) fp32 rows [gain_db, cutoff_freq, q_factor]; types[S] as in PeqSpec.
int dasp_peq_prepare(const float* params, int Bs, int S, const int* types, double sample_rate, float* tab,
                     double* dtab, void* stream) {
    if (!params || !types || !tab || !dtab || Bs <= 0 || S > 8) return DASP_ERR_ARG;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        PeqSpec spec = {};
        for (int i = 0; i < S; ++i) {
            if (types[i] < 0 || types[i] > 4) return DASP_ERR_ARG;
            spec.types[i] = types[i];
        }
        spec.sample_rate = sample_rate;
        hipLaunchKernelGGL((sos_prep_kernel<SS, kL>), dim3(Bs), dim3(256), 0, (hipStream_t)stream, nullptr, params, spec, tab, dtab);
        return check_launch();
    });
}

int dasp_sosfilt_forward(const float* tab, int Bs, const float* x, float* y, float* carries, int B, int C, long N,
                         int S, void* stream) {
    if (!tab || !x || !y || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B)) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N);
    const int vec = (N % 4 == 0) && aligned16(x) && aligned16(y);
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF>), dim3(B * C), dim3(64 * kWF), 0, (hipStream_t)stream, tab,
                           Bs == 1 && B != 1, x, y, carries, C, (int)N, nt, vec);
        return check_launch();
    });
}

int dasp_sosfilt_backward(const float* tab, int Bs, const float* x, const float* gy, const float* carries, float* gx,
                          float* partials, int B, int C, long N, int S, void* stream) {
    if (!tab || !x || !gy || !carries || !gx || !partials || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B))
        return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N);
    const int vec = (N % 4 == 0) && aligned16(x) && aligned16(gy) && aligned16(gx);
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipLaunchKernelGGL((sos_bwd_kernel<SS, kL, kWB>), dim3(B * C), dim3(64 * kWB), 0, (hipStream_t)stream, tab,
                           Bs == 1 && B != 1, x, gy, carries, gx, partials, C, (int)N, nt, vec);
        return check_launch();
    });
}

// mode 0: gout (B,S,6) = dL/dsos ; mode 1: gout (B,S,3) = dL/d(gain_db, cutoff_freq, q_factor)
int dasp_sos_grad_finalize(const double* dtab, int Bs, const float* partials, int B, int C, int S, int mode,
                           float* gout, void* stream) {
    if (!dtab || !partials || !gout || B <= 0 || C <= 0 || (Bs != 1 && Bs != B) || (mode != 0 && mode != 1))
        return DASP_ERR_ARG;
    const int n = B * S;
    hipLaunchKernelGGL(sos_finalize_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, dtab,
                       Bs == 1 && B != 1, partials, B, C, S, kWB, mode, gout);
    return check_launch();
}

}  // extern "C"
