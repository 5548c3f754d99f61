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
// Forward kernel (sos_fwd_kernel), per tile: (1) the tile arrives by LDS-DMA (requested a tile ahead) in a swizzled 4 KiB image;
// (2) z = G x : zero-state end state of every chunk, a [2S x L] x [L x 64] product on the matrix cores
//     (16 v_mfma_f32_16x16x4_f32 per tile; its left factor comes from the prep kernel);
// (3) per section k: forcing f = z_k + sum_{j<k} M_kj s_j (block-lower-triangular coupling), then an
//     inclusive scan over the 64 lanes with 2x2 matrix powers of M_kk: four Kogge-Stone levels
//     inside each 16-lane row on DPP row_shr (full-rate VALU, no LDS), two row_bcast levels across
//     rows with per-lane powers;
// (4) the tile carry K_k is handed from the wave that owns tile t-1 through an LDS mailbox; only
//     K' = e_63 + M_kk^64 K sits on that serial chain (it is lane 63's end state and that lane writes the mailbox),
//     the per-lane fix-up M_kk^(lane+1) K is off it;
// (5) the chunk's 16 outputs from its exact start state are a LINEAR map of its 16 inputs and its 2S start-state components,
//     y = T x + O s0 (T: lower-triangular Toeplitz matrix of the cascade's impulse response, O: its zero-input response per unit
//     state; LY::YM, built in fp64 by the prep kernel): 32 more MFMAs whose B operands are the input granules already in
//     registers and the scan's start states - no recursion runs inside a chunk;
// (6) the result granules go from the matrix cores' D registers straight to memory (full tiles), 1 KiB contiguous per wave instruction.
//     The chunk start states are saved for the backward pass (3 B per sample).
//
// Backward with coefficient gradients (sos_bwd_gram_kernel + sos_gram_finalize_kernel, round 4): inside a chunk every forward signal of
// the cascade is linear in u = (the chunk's 16 inputs, its 2S start-state components) and every adjoint signal linear in v = (its 16
// adjoint inputs, the 2S components of the adjoint state entering from above), so every correlation the coefficient gradients are made
// of is <C, M_kj> with C = sum over chunks of v u^T (28 x 28 for six sections) and M_kj depending on the item's coefficients only. The
// kernel walks the tiles in reverse with lane l on chunk 63 - l (the adjoint lane scan is then an ordinary ascending DPP scan); x, gy and
// the saved chunk states of the next tile arrive by LDS-DMA meanwhile; per tile it scans the adjoint system over the lanes, accumulates C
// with 64 MFMAs straight from the landed images (fp32 inside a tile, fp64 across tiles) and takes gx = T^a gy + O^a lambda from the
// mirror image of the forward output map (28 MFMAs). One 32 x 32 fp64 matrix per row leaves the kernel; the finalize kernel (one
// workgroup per item) forms P = C FW on the fp64 matrix cores, the lag sums and - through the design Jacobian - the 18 control
// gradients. No forward signal is recomputed and no correlation is summed per sample (rounds 2 - 3 did both; deleted in round 5).
// Without coefficient gradients (a fixed filter): sos_bwd_kernel, the adjoint cascade alone.
//
// Few rows (the reference trains with 8 - 32 items): every row is cut into segments that run as workgroups of their own; the segment
// states travel between the workgroups of ONE launch as tagged 64-bit words (decoupled look-back; common.hpp "look-back words and the
// sticky device error" says who may wait for whom and what a time-out does), and the Gram finalize runs inside the backward launch
// (per-workgroup lag sums, the last arriver of an item maps their sum to the gradients: sos_gram_fin.hpp). A segmented EQ step is
// three launches: design, forward, backward. DESIGN.md section 3.1 has the measurements behind each of these choices.
#include "common.hpp"
#include <type_traits>

#include "sos_tile.hpp"
#include <cstdlib>

namespace dasp {

// (p, q) x (p', q') for matrices [[p, -k q], [q, p]] (closed under multiplication for fixed k)
__device__ __forceinline__ void nmul(double k, double p1, double q1, double p2, double q2, double& p, double& q) {
    p = p1 * p2 - k * q1 * q2;
    q = p1 * q2 + q1 * p2;
}
// column-major store of [[p, -k q], [q, p]] (forward) or its transpose (adjoint)
__device__ __forceinline__ void put_blk(float* o, double p, double q, double kap, int adj) {
    o[0] = (float)p;
    o[1] = (float)(adj ? -kap * q : q);
    o[2] = (float)(adj ? q : -kap * q);
    o[3] = (float)p;
}

// ---- forward-mode dual numbers for the RBJ design Jacobian ------------------------------------------
// One partial per thread (thread = (section, control)): three short dependent fp64 chains side by side
// instead of one thread dragging three partials through every operation.
struct D1 {
    double v, d;
};
__device__ __forceinline__ D1 dconst(double c) { return {c, 0.0}; }
__device__ __forceinline__ D1 operator+(D1 a, D1 b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ D1 operator-(D1 a, D1 b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ D1 operator*(D1 a, D1 b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ D1 operator/(D1 a, D1 b) {
    const double iv = 1.0 / b.v, q = a.v * iv;
    return {q, (a.d - q * b.d) * iv};
}
__device__ __forceinline__ D1 operator*(double c, D1 a) { return {c * a.v, c * a.d}; }
__device__ __forceinline__ D1 operator+(double c, D1 a) { return {c + a.v, a.d}; }
__device__ __forceinline__ D1 operator-(double c, D1 a) { return {c - a.v, -a.d}; }
__device__ __forceinline__ D1 operator-(D1 a) { return {-a.v, -a.d}; }
__device__ __forceinline__ D1 dsin(D1 a) { return {sin(a.v), cos(a.v) * a.d}; }
__device__ __forceinline__ D1 dcos(D1 a) { return {cos(a.v), -sin(a.v) * a.d}; }
__device__ __forceinline__ D1 dsqrt(D1 a) { const double r = sqrt(a.v); return {r, 0.5 / r * a.d}; }
__device__ __forceinline__ D1 dexp(D1 a) { const double e = exp(a.v); return {e, e * a.d}; }

struct PeqSpec {
    int types[8];        // 0 peaking, 1 low_shelf, 2 high_shelf, 3 low_pass, 4 high_pass
    double sample_rate;
    const float* rows[24];   // optional: the 3 S controls as separate vectors of Bs values ([3 k + dir]); used when `params` is null
    // normalised controls (Processor.process_normalized, dasp_pytorch/modules.py:25-91): when `norm` is set the packed `params` hold values
    // on [0, 1]; control i of a section row is lo[i] + span[i] * p (modules.py:13-14), the Jacobian columns are taken w.r.t. p, and a value
    // outside [0, 1] sets bit i of *flag (the reference's ValueError, modules.py:83-84, raised by the host after one read-back)
    int norm;
    double lo[24], span[24];
    unsigned* flag;
};

// RBJ cookbook design, same formulas as dasp_pytorch/signal.py:255-304, in fp64 with the Jacobian
// d(b0,b1,b2,a1,a2 normalised)/d(gain_db, cutoff_freq, q_factor).
// c5 = normalised (b0, b1, b2, a1, a2); dc5 = their derivative w.r.t. control `dir` (0 gain_db, 1 cutoff_freq, 2 q_factor)
__device__ void rbj_design(int type, double sample_rate, double gain_db, double fc, double qf, int dir, double* c5, double* dc5) {
    const D1 g = {gain_db, dir == 0 ? 1.0 : 0.0}, f = {fc, dir == 1 ? 1.0 : 0.0}, q = {qf, dir == 2 ? 1.0 : 0.0};
    const D1 A = dexp((2.302585092994045684 / 40.0) * g);
    const D1 w0 = (2.0 * 3.14159265358979323846 / sample_rate) * f;
    double sn, cs;
    sincos(w0.v, &sn, &cs);                       // one evaluation serves sin, cos and both derivatives
    const D1 sw = {sn, cs * w0.d}, cw = {cs, -sn * w0.d};
    const D1 alpha = sw / (2.0 * q);
    const D1 sA = dsqrt(A);
    D1 b0, b1, b2, a0, a1, a2;
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
    const D1 n[5] = {b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
    for (int c = 0; c < 5; ++c) {
        c5[c] = n[c].v;
        dc5[c] = n[c].d;
    }
}

// ------------------------------------------------------------------------------------------------
// Prep kernel: one workgroup per batch item. Builds the realisation and all chunk tables in fp64.
constexpr double OM_MIN = 1e-5;
// A section whose poles are complex and at least this far (imaginary part) from the real axis is flagged "direct" in its coefficient row (rounds 2 - 4's
// recomputation kernels ran such sections in direct form between the exact chunk restarts; the flag is kept in the table layout).
#ifndef DASP_DF_OM_MIN
#define DASP_DF_OM_MIN 0.125
#endif
#ifndef DASP_STATES_CACHED
#define DASP_STATES_CACHED 1   // the backward kernel reads the saved chunk states with the caches' normal policy (not streaming)
#endif

// Normalised coefficients (b0, b1, b2, a1, a2) of section k of `item` and their derivative w.r.t. control `dir` - from the normalised /
// physical control tensors through the RBJ design, or from a row [b0 b1 b2 a0 a1 a2] as given (then a0 is returned and dc5 is zero).
template <int S>
__device__ __forceinline__ void section_coefs(const float* __restrict__ sos, const float* __restrict__ params, const PeqSpec& spec, int item, int k, int dir,
                                              bool report_range, double (&c5)[5], double (&dc5)[5], double& a0) {
    a0 = 1.0;
#pragma unroll
    for (int c = 0; c < 5; ++c) dc5[c] = 0.0;
    if (params && spec.norm) {
        const float* p = params + ((size_t)item * S + k) * 3;
        double v[3];
        for (int c = 0; c < 3; ++c) {
            const double pc = (double)p[c];
            if (report_range && spec.flag && (pc < 0.0 || pc > 1.0) && dir == 0) atomicOr(spec.flag, 1u << (3 * k + c));   // (NaN passes, as in the reference)
            v[c] = spec.lo[3 * k + c] + spec.span[3 * k + c] * pc;
        }
        rbj_design(spec.types[k], spec.sample_rate, v[0], v[1], v[2], dir, c5, dc5);
        for (int c = 0; c < 5; ++c) dc5[c] *= spec.span[3 * k + dir];          // d/d(normalised control)
    } else if (params) {
        const float* p = params + ((size_t)item * S + k) * 3;
        rbj_design(spec.types[k], spec.sample_rate, (double)p[0], (double)p[1], (double)p[2], dir, c5, dc5);
    } else if (!sos) {
        rbj_design(spec.types[k], spec.sample_rate, (double)spec.rows[3 * k][item], (double)spec.rows[3 * k + 1][item],
                   (double)spec.rows[3 * k + 2][item], dir, c5, dc5);
    } else {
        const float* sr = sos + ((size_t)item * S + k) * 6;
        a0 = (double)sr[3];
        c5[0] = sr[0] / a0; c5[1] = sr[1] / a0; c5[2] = sr[2] / a0; c5[3] = sr[4] / a0; c5[4] = sr[5] / a0;
    }
}

// The per-chunk basis responses of an item's cascade (GramFin<S>, below: what the finalize step of the Gram-matrix backward multiplies the
// Gram matrix with). They depend on the coefficients only, and their 96-step fp64 recurrences are half of that step's time - so for
// segmented rows, whose finalize step is the tail of the backward launch, the design kernel computes them here, beside its own chains.
template <int S> struct GramFin;
template <int S, bool AGENT> __device__ __forceinline__ void gram_basis_responses(const double* cf, double* bas, int tid);
template <int S> __device__ __forceinline__ constexpr int basis_doubles();       // GramFin<S>::BASIS (defined with it)

// One workgroup per item. With `basis` (segmented rows whose backward pass will follow) the grid is twice the items: workgroup nitems + i
// designs item i's sections once more and computes its basis responses (one thread per basis vector) - beside the table workgroups, not
// inside them: the 96-step fp64 recurrences are the longest chain of the launch, and as two more waves of the table workgroups (the first
// version) they held those up at the barriers (design launch 14.3 -> 16.0 us at 16 items; the launch has 16 workgroups on 256 CUs).
template <int S, int L>
__global__ void __launch_bounds__(256)
sos_prep_kernel(const float* __restrict__ sos, const float* __restrict__ params, PeqSpec spec,
                float* __restrict__ tab, double* __restrict__ dtab, int nsq_seg = 0, double* __restrict__ segtab = nullptr,
                double* __restrict__ basis = nullptr, int nitems = 0) {
    using LY = SosLayout<S, L>;
    constexpr int S2 = 2 * S, NN = S2 * S2;
    __shared__ double sec[S][10];       // sg, om, kom, g1, g2, d, kappa, b1, b2
    __shared__ double cfb[S][8];        // b0 b1 b2 a1 a2 (normalised), sg, om, 1 / om: gram_fin_coefs' numbers (basis workgroups)
    __shared__ double Phi[2][NN], T1[2][NN], T2[2][NN];
    __shared__ double vv[2][2][S2];
    if (basis && (int)blockIdx.x >= nitems) {
        const int it = blockIdx.x - nitems;
#ifdef DASP_TRACE
        if (it == 0 && threadIdx.x == 0) g_trace[48] = clock64();
#endif
        if (threadIdx.x < S) {
            double c5[5], dc5[5], a0;
            section_coefs<S>(sos, params, spec, it, threadIdx.x, 0, false, c5, dc5, a0);
            const double sg = -0.5 * c5[3];
            double om = sqrt(fabs(sg * sg - c5[4]));
            om = om < OM_MIN ? OM_MIN : om;
            for (int c = 0; c < 5; ++c) cfb[threadIdx.x][c] = c5[c];
            cfb[threadIdx.x][5] = sg; cfb[threadIdx.x][6] = om; cfb[threadIdx.x][7] = 1.0 / om;
        }
        __syncthreads();
#ifdef DASP_TRACE
        if (it == 0 && threadIdx.x == 0) g_trace[49] = clock64();
#endif
        gram_basis_responses<S, false>(&cfb[0][0], basis + (size_t)it * basis_doubles<S>(), threadIdx.x);
#ifdef DASP_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (it == 0 && threadIdx.x == 0) g_trace[50] = clock64();
#endif
        return;
    }
    __shared__ float Gsh[2][L][S2];     // chunk-table columns v_m, written out after the recursion (no global stores inside it)
    __shared__ double Pd[S][7][2];
    __shared__ float hsh[L];            // impulse response of the cascade over one chunk (output map, LY::YM)
    const int tid = threadIdx.x, item = blockIdx.x;
    PTRACE(40, 0);
    float* tb = tab + (size_t)item * LY::TOTAL;
    double* dt = dtab + (size_t)item * S * DT_STRIDE;

    if (tid < 4) tb[LY::CNT + tid] = 0.f;
    if (tid == 4) {     // the tag of this call's look-back words (lookback_publish): odd, different for every launch and item
        const unsigned tg = ((unsigned)wall_clock64() * 2654435761u) ^ ((unsigned)item * 0x9E3779B1u);
        tb[LY::TAG] = __builtin_bit_cast(float, tg | 1u);
    }
    for (int e = tid; e < 2 * L * LY::YMC; e += 256) tb[LY::YM + e] = 0.f;  // (the output maps' non-zero entries are written after the barrier below)
    if (tid < 3 * S) {   // thread = (section k, control dir): values + one Jacobian column each
        const int k = tid / 3, dir = tid % 3;
        double c5[5], dc5[5], a0;
        section_coefs<S>(sos, params, spec, item, k, dir, true, c5, dc5, a0);
        double* d = dt + k * DT_STRIDE;
        for (int c = 0; c < 5; ++c) d[DT_J + c * 3 + dir] = dc5[c];
        if (dir == 0) {
            const double b0 = c5[0], b1 = c5[1], b2 = c5[2], a1 = c5[3], a2 = c5[4];
            const double sg = -0.5 * a1, disc = sg * sg - a2;
            const double kap = disc < 0 ? 1.0 : -1.0;
            double om = sqrt(fabs(disc));
            om = om < OM_MIN ? OM_MIN : om;
            const double g1 = b1 - b0 * a1, g2 = ((b2 - b0 * a2) + g1 * sg) / om;
            sec[k][0] = sg; sec[k][1] = om; sec[k][2] = kap * om; sec[k][3] = g1; sec[k][4] = g2; sec[k][5] = b0; sec[k][6] = kap;
            sec[k][7] = b1; sec[k][8] = b2;
            // (a section that is exactly the identity, e.g. a band at 0 dB, stays in normal form: there g1 = g2 = 0 and it is exact)
            const bool direct = disc < 0 && om >= DASP_DF_OM_MIN && !(b0 == 1.0 && b1 == a1 && b2 == a2);
            float* df = tb + LY::DF + k * 8;
            df[0] = (float)b1; df[1] = (float)b2; df[2] = (float)-a1; df[3] = (float)-a2;
            // entry states of the direct forms from the normal-form chunk start state (s1, s2): transposed form II z1 = g1 s1 + g2 s2,
            // z2 = zc1 s1 + zc2 s2 (= C (A + a1 I) s); form II w[-2] = s2 / om, w[-1] = s1 + (sg / om) s2
            df[4] = (float)(-g1 * sg + g2 * om); df[5] = (float)(-g1 * kap * om - g2 * sg); df[6] = (float)(1.0 / om); df[7] = (float)(sg / om);
            float* cf = tb + LY::COEF + k * 8;
            cf[0] = (float)sg; cf[1] = (float)om; cf[2] = (float)(kap * om); cf[3] = (float)g1; cf[4] = (float)g2;
            cf[5] = (float)b0; cf[6] = (float)kap; cf[7] = direct ? 1.f : 0.f;
            d[DT_OM] = direct ? 1.0 : om;   // the correlations of a direct-form section are taken with w itself, not with om w
            for (int c = 0; c < 5; ++c) d[DT_B0 + c] = c5[c];
            d[DT_A0] = a0; d[7] = kap;
            d[23] = 0.0;
            d[DT_SG32] = (double)(float)sg; d[DT_NF] = direct ? 0.0 : 1.0; d[27] = 0.0;
        }
    }
    PTRACE(41, 0);
    __syncthreads();
    PTRACE(47, 0);
    // Phi for the forward system (sys 0) and the adjoint system (sys 1: sections reversed, A^T, B<->C). Written without branches on
    // purpose: a kernel starts with a cold instruction cache, and in its divergent if / else form this loop's ~50 taken branches
    // each paid an instruction fetch miss (measured 25-50k cycles for ~350 instructions; straight-line code streams through).
    for (int e = tid; e < 2 * NN; e += 256) {
        const int sys = e / NN, i = (e % NN) / S2, j = e % S2;
        const int kk = i / 2, r = i % 2, jj = j / 2, c = j % 2;
        const int fk = sys ? S - 1 - kk : kk, fj = sys ? S - 1 - jj : jj;  // forward section ids
        const double sg = sec[fk][0], om = sec[fk][1], kom = sec[fk][2];
        const int rr = sys ? c : r, cc = sys ? r : c;                      // A^T for the adjoint system
        const double a_el = rr == cc ? sg : (rr == 0 ? -kom : om);        // A = [[sg, -kom], [om, sg]]
        const double Bk = sys ? sec[fk][3 + r] : (r == 0 ? 1.0 : 0.0);   // B of the section at position kk
        const double Cj = sys ? (c == 0 ? 1.0 : 0.0) : sec[fj][3 + c];   // C of the section at position jj
        double gain = 1.0;                                                 // feed-through of the sections strictly between jj and kk
#pragma unroll
        for (int m = 0; m < S; ++m) {
            const double dm = sec[sys ? S - 1 - m : m][5];
            gain *= (m > jj && m < kk) ? dm : 1.0;
        }
        const double v = jj == kk ? a_el : (jj < kk ? Bk * gain * Cj : 0.0);
        Phi[sys][i * S2 + j] = v;
        T1[sys][i * S2 + j] = v;
        T2[sys][i * S2 + j] = 0.0;
    }
    PTRACE(46, 0);
    if (tid < 2 * S2) {
        const int sys = tid / S2, i = tid % S2, kk = i / 2, r = i % 2;
        const int fk = sys ? S - 1 - kk : kk;
        double gain = 1.0;
        for (int m = 0; m < kk; ++m) gain *= sec[sys ? S - 1 - m : m][5];
        vv[0][sys][i] = (sys ? sec[fk][3 + r] : (r == 0 ? 1.0 : 0.0)) * gain;
    }
    PTRACE(42, 0);
    __syncthreads();

    // chunk tables: v_m = Phi^m Bx ; forward GT[k][L-1-m] = v_m[2k..2k+1] ; adjoint (natural order) GAT[i][m] = v_m[2i..2i+1]
    // Two independent dependent chains run side by side in different waves with wave-level synchronisation only
    // (block barriers cost more than the arithmetic here): wave 0 the 16-step chunk-table recursion, wave 1 the
    // log2(L) squarings to M = Phi^L with the coupling blocks and diagonal-block powers that follow from it.
    if (tid < 64) {
        for (int m = 0; m < L; ++m) {
            const int cur = m & 1;
            if (tid < 2 * S2) {
                const int sys = tid / S2, i = tid % S2;
                const double v = vv[cur][sys][i];
                Gsh[sys][m][i] = (float)v;
                double acc = 0.0;
                for (int j = 0; j < S2; ++j) acc += Phi[sys][i * S2 + j] * vv[cur][sys][j];
                vv[cur ^ 1][sys][i] = acc;
            }
            wave_lds_sync();
        }
    } else if (tid >= 128 && tid < 128 + 1 + S2) {
        // Wave 2, beside the two chains above: the output map of a chunk (LY::YM; sos_fwd_kernel's matrix-core output path). Thread 0 runs
        // the cascade on a unit impulse from zero states (h[n] -> row n holds h[n - j] at column j), thread 1 + c on zero input from the unit
        // start state c (section c / 2, component c % 2): 16 samples of the normal-form recursion the forward kernel would run, in fp64.
        const int ex = tid - 128, c0 = ex - 1;
        double s1[S], s2[S];
#pragma unroll
        for (int k = 0; k < S; ++k) { s1[k] = (c0 == 2 * k) ? 1.0 : 0.0; s2[k] = (c0 == 2 * k + 1) ? 1.0 : 0.0; }
        for (int n = 0; n < L; ++n) {
            double u = (ex == 0 && n == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const double sg = sec[k][0], om = sec[k][1], kom = sec[k][2], g1 = sec[k][3], g2 = sec[k][4], d = sec[k][5];
                const double o = d * u + g1 * s1[k] + g2 * s2[k];
                const double t1 = sg * s1[k] - kom * s2[k] + u;
                s2[k] = om * s1[k] + sg * s2[k];
                s1[k] = t1;
                u = o;
            }
            if (ex == 0) hsh[n] = (float)u;                        // h[n]: spread over the n-th sub-diagonal after the barrier below
            else tb[LY::YM + n * LY::YMC + L + state_pos<S>(c0)] = (float)u;
        }
    } else if (tid >= 192 && tid < 192 + S2) {
        // Wave 3: the adjoint cascade's zero-input response per unit adjoint state (LY::YMA's state half): component c of the state that
        // enters the chunk from above (adjoint section i = c / 2 <-> forward section S - 1 - i), samples L - 1 down to 0
        const int c0 = tid - 192;
        double l1[S], l2[S];
#pragma unroll
        for (int i = 0; i < S; ++i) { l1[i] = (c0 == 2 * i) ? 1.0 : 0.0; l2[i] = (c0 == 2 * i + 1) ? 1.0 : 0.0; }
        for (int n = L - 1; n >= 0; --n) {
            double g = 0.0;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const int k = S - 1 - i;
                const double sg = sec[k][0], om = sec[k][1], kom = sec[k][2], g1 = sec[k][3], g2 = sec[k][4], d = sec[k][5];
                const double o = d * g + l1[i];
                const double t1 = sg * l1[i] + om * l2[i] + g1 * g;
                l2[i] = -kom * l1[i] + sg * l2[i] + g2 * g;
                l1[i] = t1;
                g = o;
            }
            tb[LY::YMA + n * LY::YMC + L + state_pos<S>(c0)] = (float)g;
        }
    } else if (tid < 128) {
        const int l = tid - 64;
        double (*src)[NN] = T1;
        double (*dst)[NN] = T2;
        for (int step = 1; step < L; step <<= 1) {
            // Phi and its powers are block lower triangular: only the S (S + 1) / 2 blocks on and below the diagonal are computed
            // (the others stay zero from the initialisation). The inner product keeps its fixed, unrolled length: a data-dependent
            // trip count runs into the cold instruction cache (measured: 2.7x slower).
            constexpr int NTRI = S * (S + 1) / 2 * 4;
            for (int e = l; e < 2 * NTRI; e += 64) {
                const int sys = e / NTRI, q = e % NTRI, blk = q >> 2;
                const int kk = (blk >= 1) + (blk >= 3) + (blk >= 6) + (blk >= 10) + (blk >= 15) + (blk >= 21) + (blk >= 28);
                const int jj = blk - kk * (kk + 1) / 2;
                const int i = 2 * kk + ((q >> 1) & 1), j = 2 * jj + (q & 1);
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < S2; ++m) acc += src[sys][i * S2 + m] * src[sys][m * S2 + j];
                dst[sys][i * S2 + j] = acc;
            }
            wave_lds_sync();
            double (*tmp)[NN] = src; src = dst; dst = tmp;
        }
        // coupling blocks, column-major
        for (int e = l; e < 2 * S * S * 4; e += 64) {
            const int sys = e / (S * S * 4), k = (e / (S * 4)) % S, j = (e / 4) % S, c = e % 4;
            tb[(sys ? LY::MCA : LY::MC) + (k * S + j) * 4 + c] = (float)src[sys][(2 * k + (c & 1)) * S2 + 2 * j + (c >> 1)];
        }
        if (l < S) {  // diagonal-block powers of the *forward* section k: M_kk = [[p, -kap q], [q, p]]
            const int k = l;
            const double kap = sec[k][6];
            double p = src[0][(2 * k) * S2 + 2 * k], q = src[0][(2 * k + 1) * S2 + 2 * k];
            for (int lv = 0; lv < 7; ++lv) {
                Pd[k][lv][0] = p; Pd[k][lv][1] = q;
                if (lv < 4) {
                    put_blk(tb + LY::PL + (k * 4 + lv) * 4, p, q, kap, 0);
                    put_blk(tb + LY::PLA + ((S - 1 - k) * 4 + lv) * 4, p, q, kap, 1);
                }
                if (lv == 6) {
                    put_blk(tb + LY::P64 + k * 4, p, q, kap, 0);
                    put_blk(tb + LY::P64A + (S - 1 - k) * 4, p, q, kap, 1);
                }
                double p2, q2;
                nmul(kap, p, q, p, q, p2, q2);
                p = p2; q = q2;
            }
        }
    }
    PTRACE(43, 0); PTRACE(44, 64);
    __syncthreads();
    for (int e = tid; e < L * L; e += 256) {            // output map, input half: T[n][j] = h[n - j] (the upper triangle stays zero)
        const int n = e / L, j = e % L;
        if (j <= n) tb[LY::YM + n * LY::YMC + j] = hsh[n - j];
        else tb[LY::YMA + n * LY::YMC + j] = hsh[j - n];
        if (j == n) tb[LY::YMA + n * LY::YMC + j] = hsh[0];
    }
    // chunk tables: forward GT[k][L-1-m] = v_m[2k..2k+1] ; adjoint (natural order) GAT[i][m] = v_m[2i..2i+1]
    for (int e = tid; e < 2 * L * S2; e += 256) {
        const int sys = e / (L * S2), m = (e / S2) % L, i = e % S2;
        if (sys == 0) tb[LY::GT + ((i >> 1) * L + (L - 1 - m)) * 2 + (i & 1)] = Gsh[0][m][i];
        else tb[LY::GAT + ((i >> 1) * L + m) * 2 + (i & 1)] = Gsh[1][m][i];
    }
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
        put_blk(tb + LY::PW + (k * 64 + c) * 4, p, q, kap, 0);
        put_blk(tb + LY::PWA + ((S - 1 - k) * 64 + c) * 4, p, q, kap, 1);
    }
    PTRACE(45, 0);
    // Segmented rows (few rows, dasp_hip.h): the per-item segment transition matrices Phi^(samples per segment) of both systems, i.e.
    // nsq_seg more squarings of the Phi^L that wave 1 left behind - what sos_segprep_kernel computes from dtab in a launch of its own
    // (same Phi, same triangular inner products, same order: the same bits), here with all four waves and a barrier per squaring.
    if (segtab) {
        constexpr int NTRI = S * (S + 1) / 2 * 4;
        int nl = 0;
        for (int v = L; v > 1; v >>= 1) ++nl;
        double (*src)[NN] = (nl & 1) ? T2 : T1;         // where the log2(L) squarings above ended
        double (*dst)[NN] = (nl & 1) ? T1 : T2;
        for (int step = 0; step < nsq_seg; ++step) {
            for (int e = tid; e < 2 * NTRI; e += 256) {
                const int sys = e / NTRI, q = e % NTRI, blk = q >> 2;
                const int kk = (blk >= 1) + (blk >= 3) + (blk >= 6) + (blk >= 10) + (blk >= 15) + (blk >= 21) + (blk >= 28);
                const int jj = blk - kk * (kk + 1) / 2;
                const int i = 2 * kk + ((q >> 1) & 1), j = 2 * jj + (q & 1);
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < S2; ++m) acc += src[sys][i * S2 + m] * src[sys][m * S2 + j];
                dst[sys][i * S2 + j] = acc;
            }
            __syncthreads();
            double (*tmp)[NN] = src; src = dst; dst = tmp;
        }
        for (int e = tid; e < 2 * NN; e += 256) segtab[(size_t)item * 2 * NN + e] = src[e / NN][e % NN];
    }
}

// ------------------------------------------------------------------------------------------------
// Segmented rows: the chain start(g + 1) = Phi_seg start(g) + z(g) over a row's segments (upwards for the forward system, downwards for
// the adjoint one), in fp64 - sos_chain_kernel's arithmetic - run by the LAST workgroup of the scan-only pre-pass to finish instead of
// by a launch of its own (a 64-thread kernel between two grid-filling ones costs a launch gap on both sides; the reference's training
// batches are launch-bound). Every workgroup makes its end state visible device-wide (the pre-pass stores nothing else, so
// the agent-scope release fence has next to nothing to write back), bumps the counter of its item - a word of the item's table that the
// prep kernel zeroed - and the one that completes the count chains the item's rows, one wave per row, and resets the counter.
__device__ __forceinline__ void seg_state_store(float* p, f2 v) {      // a segment's end state, visible to the other XCDs (see below)
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int S, int W>
__device__ __forceinline__ void chain_by_last_workgroup(int* __restrict__ cnt, int n_wg, const double* __restrict__ Phi, const float* z,
                                                        float* __restrict__ start, int row0, int nrows, int G, int adjoint,
                                                        float* __restrict__ scratch /* LDS, >= 64 * 2S floats per wave, idle by now */, int scratch_stride) {
    constexpr int S2 = 2 * S, GB = 64;
    __shared__ int s_last;
    __shared__ double s_st[W][2][S2];
    // The workgroups of an item run on different XCDs, whose L2s are not coherent with each other. A release fence (__threadfence)
    // makes every wave write back its XCD's L2 - measured: the pre-pass went from 14 to 60 us. Instead the few values that cross
    // workgroups travel as device-scope relaxed atomics (write-through stores by the writers - seg_state_store below -, cache-bypassing
    // loads here), each wave waits for the acknowledgement of its own stores (vmcnt), the barrier orders that before thread 0's counter
    // increment: the protocol of the fused finalize of sos_bwd_kernel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        s_last = handoff_arrive_is_last(cnt, n_wg);       // (resets the counter: the table can serve another pre-pass)
    }
    __syncthreads();
    if (!s_last) return;
    const int l = lane_id(), w = wave_id();
    float* zs = scratch + w * scratch_stride;
    double prow[S2];
#pragma unroll
    for (int j = 0; j < S2; ++j) prow[j] = l < S2 ? Phi[l * S2 + j] : 0.0;
    const int first = adjoint ? G - 1 : 0, step = adjoint ? -1 : 1;
    for (int row = row0 + w; row < row0 + nrows; row += W) {
        if (l < S2) {
            s_st[w][0][l] = 0.0;
            start[((size_t)row * G + first) * S2 + l] = 0.f;
        }
        int cur = 0;
        // the end states of GB segments at a time into LDS, every load of a block in flight together: they come from other workgroups
        // (other XCDs) and bypass this one's caches - a load inside the dependent chain below cost ~2.5 us per segment
        for (int n0 = 0; n0 < G - 1; n0 += GB) {
            const int nblk = G - 1 - n0 < GB ? G - 1 - n0 : GB;
            wave_lds_sync();
            for (int e = l; e < nblk * S2; e += 64) {
                const int g = first + (n0 + e / S2) * step;
                zs[e] = __hip_atomic_load(z + ((size_t)row * G + g) * S2 + e % S2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_lds_sync();
            for (int i = 0; i < nblk; ++i) {
                const int g = first + (n0 + i) * step;
                if (l < S2) {
                    double acc = (double)zs[i * S2 + l];
#pragma unroll
                    for (int j = 0; j < S2; ++j) acc += prow[j] * s_st[w][cur][j];
                    s_st[w][cur ^ 1][l] = acc;
                    start[((size_t)row * G + g + step) * S2 + l] = (float)acc;
                }
                wave_lds_sync();
                cur ^= 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SEG 0: one workgroup per row. SEG 1 / 2: the segmented scheme for few rows (oracle/chunkscan_model.py forward_row_segmented): one
// workgroup per (row, segment of Tseg tiles); 2 = the scan-only pre-pass from a zero state, which leaves the segment's end state in
// zseg[row][segment][2S]; 1 = the ordinary pass from the segment's start state segstart[row][segment][2S].
// SEG 3 (round 5): both in ONE launch - the workgroup sweeps its segment scan-only, publishes the end state, takes its start state from the
// end states of the row's earlier segments (lookback_start: a decoupled look-back - it only ever waits for workgroups with smaller indices,
// which were dispatched before it) and sweeps again for the outputs (the second read of its 32 KiB of x hits the L2). One launch boundary
// and the last-workgroup hand-off of the pre-pass less per direction.

// A segment's end state on its way to the segments behind it: every component is a 64-bit word (tag << 32 | float bits), one agent-scope
// atomic store - the reader polls the word until the tag is the launch's, so data and "it is there" arrive together and no fence is needed.
// The tag is drawn by the design launch that precedes every use (sos_prep_kernel: the item's table word LY::TAG), so stale words of an
// earlier call - the scratch buffer is whatever the allocator hands out, a graph replay reuses it - never match.
__device__ __forceinline__ void lookback_publish(unsigned long long* w, float v, unsigned tag) {
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (The reader - lookback_start - polls until a word carries the launch's tag, common.hpp lookback_poll: who may wait for whom, and what
// happens to a word that never arrives - the device error word `err[family]` - is written up there.)
// start(seg) = sum_{j < seg} Phi^(seg - 1 - j) z(j) by Horner, fp64 (chain_by_last_workgroup's arithmetic, the same order: the same bits).
// Called by the WHOLE workgroup (it contains barriers): every thread polls one word per round - a lane that walked its three words one
// after the other paid three memory round trips, ~1 us each, on the critical path of the row's last segment - then wave 0 runs the chain.
// z64 = the row's words [segment][2S]; order +1: segments 0 .. seg - 1 ascending (forward system), -1: G - 1 .. seg + 1 descending
// (adjoint system). The result goes into the inbox of the wave that owns the segment's first tile: slots [k][4] = (state, sequence number).
template <int S, int W>
__device__ __forceinline__ void lookback_start(const unsigned long long* z64, int seg, int G, int order, const double* __restrict__ Phi, unsigned tag,
                                               float* inbox, int seq, float* zs /* LDS, >= 64 * 2S floats, no wave's private data */,
                                               double (*st)[2 * S] /* LDS [2][2S] */, unsigned* err, int family) {
    constexpr int S2 = 2 * S, GB = 64;
    const int l = lane_id(), w = wave_id();
    const int npred = order > 0 ? seg : G - 1 - seg, first = order > 0 ? 0 : G - 1;
    double prow[S2];
#pragma unroll
    for (int j = 0; j < S2; ++j) prow[j] = w == 0 && l < S2 && npred > 0 ? Phi[l * S2 + j] : 0.0;
    if (w == 0 && l < S2) st[0][l] = 0.0;
    int cur = 0;
    __syncthreads();                                   // every wave is through its sweep: zs and the mailboxes are free
    for (int n0 = 0; n0 < npred; n0 += GB) {           // (uniform over the workgroup: the barriers below are reached by every thread)
        const int nblk = npred - n0 < GB ? npred - n0 : GB;
        if (n0) __syncthreads();                       // the previous block's chain has read zs
        for (int e = threadIdx.x; e < nblk * S2; e += 64 * W) {
            const unsigned long long* wp = z64 + (size_t)(first + order * (n0 + e / S2)) * S2 + e % S2;
            zs[e] = lookback_poll(wp, tag, err, family);
        }
        __syncthreads();
        if (w == 0) {
            for (int i = 0; i < nblk; ++i) {
                if (l < S2) {
                    double acc = (double)zs[i * S2 + l];
#pragma unroll
                    for (int j = 0; j < S2; ++j) acc += prow[j] * st[cur][j];
                    st[cur ^ 1][l] = acc;
                }
                wave_lds_sync();
                cur ^= 1;
            }
        }
    }
    if (w == 0) {
        wave_lds_sync();
        if (l < S2) inbox[4 * (l >> 1) + (l & 1)] = (float)st[cur][l];
        if (l < S) inbox[4 * l + 2] = __builtin_bit_cast(float, seq);
    }
}

template <int S, int L, int W, int SEG = 0>
__global__ void __launch_bounds__(64 * W, W >= 16 ? 4 : (W * 2 + 3) / 4)   // two workgroups per CU (W = 16, few rows: one)
sos_fwd_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ x, float* __restrict__ y,
               float* __restrict__ carries, int C, int N, int nt, int vec,
               int G = 1, int Tseg = 0, const float* __restrict__ segstart = nullptr, float* __restrict__ zseg = nullptr,
               float* __restrict__ chain_tab = nullptr, const double* __restrict__ segtab = nullptr, float* __restrict__ chain_start = nullptr,
               unsigned* __restrict__ err = nullptr) {
    using LY = SosLayout<S, L>;
    constexpr int S2 = 2 * S, TS = 64 * L, IMG = 64 * L;        // unpadded, swizzled tile images (common.hpp)
    constexpr int LDS_T = W * 2 * IMG, LDS_MB = W * S * 4, LDS_PW = S * 64 * 4;
    __shared__ __attribute__((aligned(16))) float lds[LDS_T + LDS_MB + LDS_PW];
    const int lane = lane_id(), wave = wave_id();
    const int row = SEG ? blockIdx.x / G : blockIdx.x, seg = SEG ? blockIdx.x % G : 0;
    const int t0 = SEG ? seg * Tseg : 0, t1 = SEG ? (t0 + Tseg < nt ? t0 + Tseg : nt) : nt;   // this workgroup's tiles
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : row / C) * LY::TOTAL;
    const float* __restrict__ xr = x + (size_t)row * N;
    float* __restrict__ yr = y + (size_t)row * N;
    // mailboxes and per-lane powers first: below 64 KiB their addresses fold into the 16-bit DS offset field (behind
    // the tile images every slot needed an address register of its own, ~14 VGPRs that ended up spilled on the carry chain)
    const int mb_in = wave * S * 4, mb_out = ((wave + 1) % W) * S * 4;
    float* pw_lds = lds + LDS_MB;
    float* tbx = pw_lds + LDS_PW + wave * 2 * IMG;   // x image: this tile's, then (by LDS-DMA, as soon as it has been read) the next one's
    float* tby = tbx + IMG;                          // y image on its way out
    // mailboxes zeroed; wave 0's inbox holds what its first tile t0 waits for: sequence number t0 and the state the segment starts from
    if (SEG) {
        for (int i = threadIdx.x; i < LDS_MB; i += 64 * W) {
            float v = 0.f;
            if (i < S * 4) {
                const int comp = i & 3;
                if (comp == 2) v = __builtin_bit_cast(float, t0);
                else if (SEG == 1 && comp < 2) v = segstart[((size_t)row * G + seg) * S2 + 2 * (i >> 2) + comp];
            }
            lds[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < LDS_MB; i += 64 * W) lds[i] = 0.f;
    }
    for (int i = threadIdx.x; i < LDS_PW; i += 64 * W) pw_lds[i] = tb[LY::PW + i];
    __syncthreads();
    const f4* pws = reinterpret_cast<const f4*>(pw_lds);

    f2 Kreg[S];
#pragma unroll
    for (int k = 0; k < S; ++k) Kreg[k] = f2{0.f, 0.f};
    const unsigned a_x = __builtin_amdgcn_readfirstlane(lds_addr(tbx));
    float Aop[4];
    chunk_table_operands<S, L>(tb + LY::GT, Aop, lane);
    // The cascade over the chunk on the matrix cores (round 4; sos_tile.hpp cascade_outputs_mfma: y = T x + O s0, LY::YM) - 32
    // v_mfma_f32_16x16x4_f32 beside the 16 of the chunk products where rounds 1 - 3 ran 768 vector instructions of per-lane recursion.
    static_assert(L == 16 && S2 <= 16, "one 16 x 16 output block per 16 chunks");
    float AT[4], AO[4];
    if (SEG != 2) cascade_map_operands<S, L>(tb + LY::YM, LY::YMC, AT, AO, lane);      // (SEG 2: scan only, no outputs)
    // look-back launches: the launch's tag and the row's words
    const unsigned tag = SEG == 3 ? __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, tb[LY::TAG])) : 0u;
    unsigned long long* z64 = reinterpret_cast<unsigned long long*>(zseg) + (size_t)row * G * S2;

    // One sweep over the workgroup's tiles. SCAN: the lane scans only (a pre-pass: nothing is stored but the segment's end state);
    // seq0: added to the mailboxes' sequence numbers (a second sweep must not mistake the first one's entries for its own).
    auto sweep = [&](auto scan_tag, const int seq0, const bool first_tile_requested = false) {
        constexpr bool SCAN = decltype(scan_tag)::value;
        if (!first_tile_requested && t0 + wave < t1 && tile_full<L>((long)(t0 + wave) * TS, N, vec)) tile_dma_issue_swz(xr + (size_t)(t0 + wave) * TS, a_x, lane);
        int stores_in_flight = 0;
        for (int t = t0 + wave; t < t1; t += W) {
            int toff = 0;
            asm volatile("" : "+s"(toff));   // opaque uniform 0: keeps the scalar table loads inside the tile loop (no SGPR spills)
            const float* __restrict__ tbl = tb + toff;
            const bool full = tile_full<L>((long)t * TS, N, vec);
            WIDE_PRIO(DASP_SCAN_PRIO);
            TRACE(0);
            // The x image of this tile was requested one tile ago (LDS-DMA, no staging registers); with the loads exposed at the top
            // of every tile the kernel ran 24 % above its compute-only time. vmcnt is in order: the previous tile's state and y stores,
            // issued after that request, may stay in flight.
            if (full) wait_vmcnt(stores_in_flight);
            else tile_global_to_swz_guarded(tbx, xr, (long)t * TS, N);
            f4 Bop[4], zacc[4];
            chunk_products_load(tbx, Bop, lane);
            pin(Bop);
            if (t + W < t1 && tile_full<L>((long)(t + W) * TS, N, vec)) tile_dma_issue_swz(xr + (size_t)(t + W) * TS, a_x, lane);
            TRACE(1);
            float Z[L];
            chunk_products_issue(Bop, Aop, zacc);
            chunk_products_collect<L>(tby, zacc, Z, lane, lane);   // the y image is idle until the end of the tile
            pin(Z); TRACE(5);

            f2 st[S];
            MboxPeek pk;
            SCAN_PRIO(DASP_SCAN_PRIO);
            tile_scan<S, L>(Z, [](f2 v) { return v; }, st, tbl + LY::MC, tbl + LY::PL, tbl + LY::P64, pws, lane,
                [&](int k) { if (W > 1) pk = mbox_peek(lds, mb_in + 4 * k); },   // waited for with the per-lane powers (same lgkmcnt(0))
                [&](int k, f2& K) {
                    if (W == 1) K = Kreg[k];
                    else if (pk.seq == t + seq0) K = f2{pk.a, pk.b};   // tile 0 finds the zero-initialised inbox: sequence 0, carry 0
                    else { float a, b; mbox_wait(lds, mb_in + 4 * k, t + seq0, a, b); K = f2{a, b}; }
                },
                [&](int k, f2 Kn) {
                    if (W == 1) Kreg[k] = f2{read_lane(Kn.x, 63), read_lane(Kn.y, 63)};
                    else if (t + 1 < t1) mbox_publish<63>(lds, mb_out + 4 * k, Kn.x, Kn.y, t + 1 + seq0);
                    else if (SCAN && lane == 63) {       // the segment's end state
                        if (SEG == 3) { lookback_publish(z64 + (size_t)seg * S2 + 2 * k, Kn.x, tag); lookback_publish(z64 + (size_t)seg * S2 + 2 * k + 1, Kn.y, tag); }
                        else seg_state_store(zseg + ((size_t)row * G + seg) * S2 + 2 * k, Kn);
                    }
                }
#ifdef DASP_TRACE
                , blockIdx.x == 7 && threadIdx.x == 64 && t >= 40 && t < 40 + W
#endif
                );
            SCAN_PRIO(0);
            TRACE(2);
            if (SCAN) {              // scan-only: the carries are all this sweep is for; nothing was stored
                stores_in_flight = 0;
                continue;
            }
            if (carries) {   // chunk start states for the backward pass: [row][tile][section pair][lane] f4, 1 KiB per wave store
                static_assert(S % 2 == 0, "states are stored in section pairs");
                f4* cs = reinterpret_cast<f4*>(carries) + ((size_t)row * nt + t) * (S / 2) * 64 + lane;
#pragma unroll
                for (int m = 0; m < S / 2; ++m) {
                    const f4 v = f4{st[2 * m].x, st[2 * m].y, st[2 * m + 1].x, st[2 * m + 1].y};
                    if (DASP_FWD_NT & 4) st_stream(cs + m * 64, v); else cs[m * 64] = v;
                }
            }

            f4 yacc[4];
            cascade_outputs_mfma_acc<S, L>(tby, st, Bop, AT, AO, lane, yacc);
            TRACE(3);
            WIDE_PRIO(DASP_SCAN_PRIO);
            if (DASP_DIRECT_OUT && full) {       // the output granules straight from the matrix cores' result registers to memory
                mfma_granules_to_global(yr + (size_t)t * TS, yacc, DASP_FWD_NT & 2, lane);
            } else {
                mfma_granules_to_image(tby, yacc, lane);
                if (full) tile_swz_to_global_full(tby, yr, (long)t * TS, DASP_FWD_NT & 2, lane);
                else tile_swz_to_global_guarded(tby, yr, (long)t * TS, N);
            }
            stores_in_flight = full ? (carries ? S / 2 : 0) + L / 4 : -1;
            TRACE(4);
        }
    };
    if (SEG == 3) {
        sweep(std::true_type{}, 0);
        // the start state from the end states of the segments in front of this one -> wave 0's inbox, sequence t0 + SEQ2
        constexpr int SEQ2 = 1 << 28;
        __shared__ double lb_st[2][S2];
        // the output sweep's first x tile is on its way while the look-back runs (the look-back stages its words in wave 0's y image, idle
        // between the sweeps; the x images are the waves' own)
        if (t0 + wave < t1 && tile_full<L>((long)(t0 + wave) * TS, N, vec)) tile_dma_issue_swz(xr + (size_t)(t0 + wave) * TS, a_x, lane);
        lookback_start<S, W>(z64, seg, G, 1, segtab + (size_t)(tab_bcast ? 0 : row / C) * 2 * S2 * S2, tag, lds, t0 + SEQ2, pw_lds + LDS_PW + IMG, lb_st, err, DASP_DEVERR_SOS_FWD);
        __syncthreads();
        sweep(std::false_type{}, SEQ2, true);
    } else if (SEG == 2) {
        sweep(std::true_type{}, 0);
    } else {
        sweep(std::false_type{}, 0);
    }
    if (SEG == 2 && chain_tab) {      // scan-only pre-pass: the last workgroup of the item (of the call, with a shared table) chains its rows
        const int item = tab_bcast ? 0 : row / C;
        chain_by_last_workgroup<S, W>(reinterpret_cast<int*>(chain_tab + (size_t)item * LY::TOTAL + LY::CNT) + 1, tab_bcast ? (int)gridDim.x : C * G,
                                      segtab + (size_t)item * 2 * S2 * S2, zseg, chain_start, tab_bcast ? 0 : item * C,
                                      tab_bcast ? (int)gridDim.x / G : C, G, 0, pw_lds + LDS_PW, 2 * IMG);      // (the tile images are idle now)
    }
}

#include "sos_gram_fin.hpp"      // EmitCoef / emit_section_grads, GramFin ... gram_fused_tail

// ------------------------------------------------------------------------------------------------
// The adjoint cascade on its own (one workgroup of W waves per row; lane l on chunk 63 - l so that the adjoint lane scan, which runs from
// the last chunk to the first, is an ordinary ascending DPP scan). Two uses:
//   SEG 0 / 1  the backward pass of a FIXED filter (no coefficient gradients asked for: x and the saved chunk states are not read): gx by the
//              per-lane cascade, every section in transposed direct form II from the exact chunk costate (o = b0 g + z1; z1 = b1 g - a1 o +
//              z2; z2 = b2 g - a2 o: 5 ops against 7 in normal form; direct forms lose digits over long horizons, not over the 16 samples
//              between two exact restarts: the chunk's entry costate comes from the normal-form lane scan and is mapped once per chunk,
//              z1 = l1, z2 = -sg l1 + om l2). 80 % of 8 TB/s at (256, 2, 131072). SEG 1: one workgroup per (row, segment of Tseg tiles),
//              the adjoint state entering the segment from above in segstart[row][segment][2S].
//   SEG 2      the adjoint scan-only pre-pass of segmented rows (oracle/chunkscan_model.py backward_row_segmented): from a zero adjoint state,
//              leaving the state below the segment in zseg[row][segment][2S]; its last workgroup per item chains the segments
//              (chain_by_last_workgroup). The pass that follows is sos_bwd_gram_kernel<SEG = 1> (gradients) or this kernel's SEG 1.
// With coefficient gradients the backward pass is sos_bwd_gram_kernel below; rounds 2 - 4's recomputation kernels (all six sections
// recomputed from the saved states, five correlations per section in every lane) are in the history of this file.
template <int S, int L, int W, int SEG = 0>
__global__ void __launch_bounds__(64 * W, (W * 2 + 3) / 4)   // two workgroups per CU
sos_bwd_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ gy, float* __restrict__ gx, int C, int N, int nt, int vec,
               int G = 1, int Tseg = 0, const float* __restrict__ segstart = nullptr, float* __restrict__ zseg = nullptr,
               float* __restrict__ chain_tab = nullptr, const double* __restrict__ segtab = nullptr, float* __restrict__ chain_start = nullptr) {
    using LY = SosLayout<S, L>;
    constexpr bool GX = SEG != 2;
    constexpr int S2 = 2 * S, TS = 64 * L, IMG = 64 * L, REGION = 2 * IMG;      // per wave: gy landing image, gx staging image (unpadded, swizzled: common.hpp)
    constexpr int LDS_T = W * REGION, LDS_MB = W * S * 4, LDS_PW = S * 64 * 4, LDS_CF = S * 16;   // COEF rows, DF rows
    __shared__ __attribute__((aligned(16))) float lds[LDS_T + LDS_MB + LDS_PW + LDS_CF];
    const int lane = lane_id(), wave = wave_id();
    const int row = SEG ? blockIdx.x / G : blockIdx.x, seg = SEG ? blockIdx.x % G : 0;
    const int t0 = SEG ? seg * Tseg : 0, t1 = SEG ? (t0 + Tseg < nt ? t0 + Tseg : nt) : nt, nr = t1 - t0;   // this workgroup's tiles, walked t1 - 1 .. t0
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : row / C) * LY::TOTAL;
    const float* __restrict__ gr = gy + (size_t)row * N;
    float* __restrict__ gxr = gx + (size_t)row * N;
    const int mb_in = wave * S * 4, mb_out = ((wave + 1) % W) * S * 4;   // small regions first (16-bit DS offsets), as in the forward kernel
    float* cf_lds = lds + LDS_MB;
    float* pw_lds = cf_lds + LDS_CF;
    float* tbg = pw_lds + LDS_PW + wave * REGION;  // gy image of this tile; receives the next tile's as soon as it has been read
    float* tbo = tbg + IMG;                        // gx image on its way out
    // mailboxes zeroed; wave 0's inbox carries the sequence number its first tile (t1 - 1: the row's or the segment's last) waits for,
    // with the adjoint state that enters from above (zero at the end of the row)
    for (int i = threadIdx.x; i < LDS_MB; i += 64 * W) {
        float v = 0.f;
        if (i < S * 4) {
            const int comp = i & 3;
            if (comp == 2) v = __builtin_bit_cast(float, t1);
            else if (SEG == 1 && comp < 2) v = segstart[((size_t)row * G + seg) * S2 + 2 * (i >> 2) + comp];
        }
        lds[i] = v;
    }
    for (int i = threadIdx.x; i < LDS_PW; i += 64 * W) pw_lds[i] = tb[LY::PWA + i];
    for (int i = threadIdx.x; i < S * 8; i += 64 * W) cf_lds[i] = tb[LY::COEF + i];
    for (int i = threadIdx.x; i < S * 8; i += 64 * W) cf_lds[S * 8 + i] = tb[LY::DF + i];
    __syncthreads();
    const f4* pwa = reinterpret_cast<const f4*>(pw_lds);
    f2 Kreg[S];
#pragma unroll
    for (int k = 0; k < S; ++k) Kreg[k] = f2{0.f, 0.f};

    // LDS-DMA prefetch of the next tile: issued as soon as this tile's image has been read into registers, i.e. a whole tile time before it
    // is needed, with no staging registers. vmcnt is in order on gfx9, so the wait at the top of a tile lets the previous tile's L/4 gx
    // stores stay in flight.
    const unsigned a_g = __builtin_amdgcn_readfirstlane(lds_addr(tbg));
    if (wave < nr && tile_full<L>((long)(t1 - 1 - wave) * TS, N, vec)) tile_dma_issue_swz(gr + (size_t)(t1 - 1 - wave) * TS, a_g, lane);
    int stores_in_flight = 0;
    float Aop[4];
    chunk_table_operands<S, L>(tb + LY::GAT, Aop, lane);

    for (int r = wave; r < nr; r += W) {
        const int t = t1 - 1 - r;
        int toff = 0;
        asm volatile("" : "+s"(toff));   // opaque uniform 0: keeps the scalar table loads inside the tile loop
        const float* __restrict__ tbl = tb + toff;
        const bool full = tile_full<L>((long)t * TS, N, vec);
        float GY[L];
        WIDE_PRIO(DASP_SCAN_PRIO);
        TRACE(16);
        if (full) {
            if (GX && stores_in_flight == L / 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(L / 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            tile_global_to_swz_guarded(tbg, gr, (long)t * TS, N);
        }
        const int cl = 63 - lane;
        lds_to_chunks_swz<L>(tbg, GY, cl);
        f4 Bop[4], zacc[4];
        chunk_products_load(tbg, Bop, lane);
        pin(GY); pin(Bop); TRACE(17);
        if (r + W < nr) tile_dma_issue_swz(gr + (size_t)(t - W) * TS, a_g, lane);   // the image is in registers now; tiles below a row's last one are always full
        float Z[L];
        chunk_products_issue(Bop, Aop, zacc);
        chunk_products_collect<L>(tbo, zacc, Z, lane, cl);   // the gx image is idle until the end of the tile
        pin(Z); TRACE(25);
        // ---- adjoint chunk end states: the scan runs from the last chunk to the first = ascending lanes (chunk 63 - lane) ----
        f2 lam[S];  // adjoint section order: i <-> forward section S-1-i
        {
            MboxPeek pk;
            SCAN_PRIO(DASP_SCAN_PRIO);
            tile_scan<S, L>(Z, [](f2 v) { return v; }, lam,
                tbl + LY::MCA, tbl + LY::PLA, tbl + LY::P64A, pwa, lane,
                [&](int i) { if (W > 1) pk = mbox_peek(lds, mb_in + 4 * i); },
                [&](int i, f2& K) {
                    if (W == 1) K = Kreg[i];
                    else if (pk.seq == t + 1) K = f2{pk.a, pk.b};   // the last tile finds wave 0's inbox as initialised: sequence t1, carry 0 / the segment's state
                    else { float a, b; mbox_wait(lds, mb_in + 4 * i, t + 1, a, b); K = f2{a, b}; }
                },
                [&](int i, f2 Kn) {
                    if (W == 1) Kreg[i] = f2{read_lane(Kn.x, 63), read_lane(Kn.y, 63)};
                    else if (t > t0) mbox_publish<63>(lds, mb_out + 4 * i, Kn.x, Kn.y, t);
                    else if (SEG == 2 && lane == 63) seg_state_store(zseg + ((size_t)row * G + seg) * S2 + 2 * i, Kn);   // state below the segment
                });
        }
        SCAN_PRIO(0);
        if (SEG == 2) {          // adjoint scan-only pre-pass
            stores_in_flight = 0;
            continue;
        }
        pin(GY); pin(lam);   // scans done before the cascade starts
        TRACE(19);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {      // adjoint section k (descending time), in place over GY; the coefficient loads are addressed with
            const int oz = opaque_zero_after(GY[0]);      // an opaque zero chained behind the previous section: VGPR operands, one section's set live
            const int i = S - 1 - k;
            const f4 ca = *reinterpret_cast<const f4*>(cf_lds + k * 8 + oz);           // sg, om, kom, g1
            const float d = cf_lds[k * 8 + 5 + oz];                                     // b0
            const f4 cd = *reinterpret_cast<const f4*>(cf_lds + S * 8 + k * 8 + oz);   // b1, b2, -a1, -a2
            float z1 = lam[i].x, z2 = fmaf(ca.y, lam[i].y, -ca.x * lam[i].x);
#pragma unroll
            for (int n = L - 1; n >= 0; --n) {
                const float g = GY[n];
                const float o = fmaf(d, g, z1);
                z1 = fmaf(cd.x, g, fmaf(cd.z, o, z2));
                z2 = fmaf(cd.y, g, cd.w * o);
                GY[n] = o;
            }
        }
        pin(GY);
        TRACE(23);
        WIDE_PRIO(DASP_SCAN_PRIO);
        __builtin_amdgcn_sched_barrier(0);
        chunks_to_lds_swz<L>(tbo, GY, cl);
        if (full) tile_swz_to_global_full(tbo, gxr, (long)t * TS, true, lane);
        else tile_swz_to_global_guarded(tbo, gxr, (long)t * TS, N);
        stores_in_flight = full ? L / 4 : 0;
        TRACE(24);
    }
    if (SEG == 2 && chain_tab) {      // adjoint scan-only pre-pass: chain downwards with the adjoint system's segment matrix (word 2 of the counters)
        const int item = tab_bcast ? 0 : row / C;
        chain_by_last_workgroup<S, W>(reinterpret_cast<int*>(chain_tab + (size_t)item * LY::TOTAL + LY::CNT) + 2, tab_bcast ? (int)gridDim.x : C * G,
                                      segtab + ((size_t)item * 2 + 1) * S2 * S2, zseg, chain_start, tab_bcast ? 0 : item * C,
                                      tab_bcast ? (int)gridDim.x / G : C, G, 1, pw_lds + LDS_PW, REGION);       // (the tile images are idle now)
    }
}

// ------------------------------------------------------------------------------------------------
// Segmented rows (few rows: B*C workgroups do not fill the chip). Per item: Phi^(samples per segment) of the forward and the adjoint
// cascade, in fp64 from the fp64 design in dtab (same realisation and Phi as the prep kernel), by repeated squaring. One workgroup of four
// waves per item; Phi and its powers are block lower triangular, so a squaring is 2 x 84 inner products - one per thread - and a
// barrier (one wave walking all 2 x 144 elements of every squaring took 14 us for the 14 squarings of a 16-tile segment; this: ~5).
template <int S>
__global__ void __launch_bounds__(256)
sos_segprep_kernel(const double* __restrict__ dtab, int nsq, double* __restrict__ segtab) {
    constexpr int S2 = 2 * S, NN = S2 * S2;
    __shared__ double sec[S][8];
    __shared__ double T1[2][NN], T2[2][NN];
    const int l = threadIdx.x, item = blockIdx.x;
    if (l < S) {
        const double* d = dtab + ((size_t)item * S + l) * DT_STRIDE;
        const double b0 = d[DT_B0], b1 = d[DT_B0 + 1], b2 = d[DT_B0 + 2], a1 = d[DT_B0 + 3], a2 = d[DT_B0 + 4];
        const double sg = -0.5 * a1, disc = sg * sg - a2, kap = disc < 0 ? 1.0 : -1.0;
        double om = sqrt(fabs(disc));
        om = om < OM_MIN ? OM_MIN : om;
        const double g1 = b1 - b0 * a1, g2 = ((b2 - b0 * a2) + g1 * sg) / om;
        sec[l][0] = sg; sec[l][1] = om; sec[l][2] = kap * om; sec[l][3] = g1; sec[l][4] = g2; sec[l][5] = b0;
    }
    __syncthreads();
    for (int e = l; e < 2 * NN; e += 256) {   // as in sos_prep_kernel
        const int sys = e / NN, i = (e % NN) / S2, j = e % S2;
        const int kk = i / 2, r = i % 2, jj = j / 2, c = j % 2;
        const int fk = sys ? S - 1 - kk : kk, fj = sys ? S - 1 - jj : jj;
        const double sg = sec[fk][0], om = sec[fk][1], kom = sec[fk][2];
        const int rr = sys ? c : r, cc = sys ? r : c;
        const double a_el = rr == cc ? sg : (rr == 0 ? -kom : om);
        const double Bk = sys ? sec[fk][3 + r] : (r == 0 ? 1.0 : 0.0);
        const double Cj = sys ? (c == 0 ? 1.0 : 0.0) : sec[fj][3 + c];
        double gain = 1.0;
#pragma unroll
        for (int m = 0; m < S; ++m) {
            const double dm = sec[sys ? S - 1 - m : m][5];
            gain *= (m > jj && m < kk) ? dm : 1.0;
        }
        T1[sys][i * S2 + j] = jj == kk ? a_el : (jj < kk ? Bk * gain * Cj : 0.0);
        T2[sys][i * S2 + j] = 0.0;                                  // the blocks above the diagonal stay zero in every power
    }
    __syncthreads();
    double (*src)[NN] = T1;
    double (*dst)[NN] = T2;
    constexpr int NTRI = S * (S + 1) / 2 * 4;                       // elements of the blocks on and below the diagonal, per system
    for (int step = 0; step < nsq; ++step) {
        for (int e = l; e < 2 * NTRI; e += 256) {
            const int sys = e / NTRI, q = e % NTRI, blk = q >> 2;
            const int kk = (blk >= 1) + (blk >= 3) + (blk >= 6) + (blk >= 10) + (blk >= 15) + (blk >= 21) + (blk >= 28);
            const int jj = blk - kk * (kk + 1) / 2;
            const int i = 2 * kk + ((q >> 1) & 1), j = 2 * jj + (q & 1);
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < S2; ++m) acc += src[sys][i * S2 + m] * src[sys][m * S2 + j];
            dst[sys][i * S2 + j] = acc;
        }
        __syncthreads();
        double (*tmp)[NN] = src; src = dst; dst = tmp;
    }
    for (int e = l; e < 2 * NN; e += 256) segtab[(size_t)item * 2 * NN + e] = src[e / NN][e % NN];
}

// Chains the segments of a row: start(g + 1) = Phi_seg start(g) + z(g) upwards for the forward system (adjoint = 0), downwards
// aend(g - 1) = Phia_seg aend(g) + za(g) for the adjoint system (adjoint = 1). z and the result are [row][G][2S] fp32; one wave per row.
template <int S>
__global__ void __launch_bounds__(64)
sos_chain_kernel(const double* __restrict__ segtab, int tab_bcast, int C, const float* __restrict__ z, float* __restrict__ start, int G,
                 int adjoint) {
    constexpr int S2 = 2 * S, NN = S2 * S2;
    __shared__ double st[2][S2];
    const int l = threadIdx.x, row = blockIdx.x;
    const double* Phi = segtab + ((size_t)(tab_bcast ? 0 : row / C) * 2 + (adjoint ? 1 : 0)) * NN;
    double prow[S2];
#pragma unroll
    for (int j = 0; j < S2; ++j) prow[j] = l < S2 ? Phi[l * S2 + j] : 0.0;
    const int first = adjoint ? G - 1 : 0, step = adjoint ? -1 : 1;
    if (l < S2) {
        st[0][l] = 0.0;
        start[((size_t)row * G + first) * S2 + l] = 0.f;
    }
    wave_lds_sync();
    int cur = 0;
    for (int n = 0, g = first; n < G - 1; ++n, g += step) {
        if (l < S2) {
            double acc = (double)z[((size_t)row * G + g) * S2 + l];
#pragma unroll
            for (int j = 0; j < S2; ++j) acc += prow[j] * st[cur][j];
            st[cur ^ 1][l] = acc;
            start[((size_t)row * G + g + step) * S2 + l] = (float)acc;
        }
        wave_lds_sync();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward by Gram matrix (round 4; one workgroup per row, coefficient gradients wanted). No recomputation of the cascade at all.
// Within a chunk every signal of the cascade is a LINEAR function of u = (the chunk's L inputs x, its 2S forward start-state components)
// and every adjoint signal a linear function of v = (the chunk's L adjoint inputs gy, the 2S components of the adjoint state that enters
// it from above). Every correlation the coefficient gradients are made of - sum_n g_k[n] w_k[n - j], sum_n o_k[n] w_k[n - j] over all
// samples of the row - is therefore a bilinear form in (v, u) summed over the chunks, i.e. <C, M_kj> with
//     C = sum over chunks of v u^T          ((L + 2S) x (L + 2S): 28 x 28 for the EQ)
// and M_kj built from the per-chunk basis responses of the item's cascade, which depend on the item's coefficients only. The kernel
// accumulates C on the matrix cores: the contraction index of the product is the chunk, and with the tile images read row-major (lane
// 16 k + i: entry i of chunk 4 m + k, one ds_read_b32 per step m) the A / B operands of v_mfma_f32_16x16x4_f32 are the images as they
// are - 64 MFMAs per tile for the four 16 x 16 blocks of C (padded to 32 x 32), fp32 within the tile, folded into fp64 sums per tile
// (the part that needs it: C's entries are sums over the whole row of products of O(1) signals, and the gradients are differences of
// nearly equal combinations of them at the low-frequency corner of the EQ's ranges). What is left per tile besides that:
//   - the adjoint lane scan (as in sos_bwd_kernel: Kogge-Stone over the lanes, mailboxes between waves; its table products on the VALU),
//   - gx = TA gy + OA lam, the adjoint cascade's outputs as a linear map of v (LY::YMA, 28 MFMAs: the forward kernel's output path mirrored;
//     the adjoint states three to a granule so that the all-zero contraction step is skipped, sos_tile.hpp state_pos),
//   - no forward scan: the forward chunk start states are the ones the forward kernel saved; they land by LDS-DMA directly in the
//     [chunk][16] image the products read (the DMA's global addresses do the transposition: lane = (chunk, section pair)).
// The 30 dependent VALU instructions per sample of sos_bwd_kernel (recomputation + adjoint + correlations) become 92 MFMAs per tile
// next to the scan. (They do not run beside it: on gfx950 an fp32 MFMA occupies its SIMD's vector issue for its 32 cycles - tools/ubench5.hip -
// so the kernel is bound by the sum of both; the products that do not depend on the scan are still issued from hook points
// inside it, which measured 3 % better than one block.) sos_gram_finalize_kernel turns sum-over-rows(C) into the gradients (fp64 basis
// responses). Executable specification and error budget: oracle/chunkscan_model.py (gram_backward_row), tests/test_oracle_cpu.py.
// gram: [row][16 registers][64 lanes] doubles - register 4 (2 bv + bu) + e of lane l = C[16 bv + 4 (l / 16) + e][16 bu + l % 16].
__device__ __forceinline__ void gram_operands_load(const float* img, float (&R)[16], int lane) {
    const int k = lane >> 4, i = lane & 15;
#pragma unroll
    for (int m = 0; m < 16; ++m) R[m] = img[64 * m + 16 * k + 4 * ((i >> 2) ^ (m & 3)) + (i & 3)];      // (swz_slot(4 m + k, i / 4), entry i % 4)
}

// FLAGS & BWD_NOGX: no gradient for x is wanted (the EQ is the first effect of the reference's chain: examples/style_transfer.py:150) - no
// output map, 4 B per sample less.
constexpr int BWD_NOGX = 2;
// SEG = 1: segmented rows (few rows): workgroup = (row, segment of Tseg tiles), the adjoint state entering the segment from above comes from
// segstart[row][segment][2S] (the scan-only pre-pass of sos_bwd_kernel<SEG = 2> and its chain); one matrix per (row, segment).
template <int S, int L, int W, int FLAGS, int SEG = 0>
__global__ void __launch_bounds__(64 * W, 2)   // two waves per SIMD: two workgroups of 4 waves per CU, or one of 8 (few rows)
sos_bwd_gram_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ x,
                    const float* __restrict__ gy, const float* __restrict__ carries, float* __restrict__ gx,
                    double* __restrict__ gram, int C, int N, int nt, int vec, int G = 1, int Tseg = 0, const float* __restrict__ segstart = nullptr,
                    GramFuse fz = GramFuse{}) {
    using LY = SosLayout<S, L>;
    static_assert(L == 16 && 2 * S <= 16, "one 16-wide block of state components");
    constexpr bool GX = !(FLAGS & BWD_NOGX);
    constexpr int TS = 64 * L, IMG = 64 * L, REGION = 4 * IMG;      // per wave: gy, x, states, scratch (chunk products / adjoint states / gx on its way out)
    constexpr int LDS_T = W * REGION, LDS_MB = W * S * 4, LDS_PW = S * 64 * 4;
    static_assert(REGION >= 2 * 1024, "a wave's region holds its 1024 fp64 sums at the end");
    __shared__ __attribute__((aligned(16))) float lds[LDS_T + LDS_MB + LDS_PW];
    const int lane = lane_id(), wave = wave_id();
    const int wg = blockIdx.x;
    if constexpr (SEG != 0) {
        static_assert(W == 4, "the finalize steps are written for 256 threads");
        static_assert(GramFin<S>::WORK * 2 <= LDS_T + LDS_MB + LDS_PW, "the finalize work area fits the tile images");
    }
    // SEG 3 (round 5, with fz.on): the adjoint scan-only pre-pass inside this launch - the workgroup sweeps its segment's gy scan-only,
    // publishes the adjoint state below the segment, takes the state entering from above from the segments above it (lookback_start, the
    // forward kernel's scheme mirrored) and then runs the pass. segstart then points at the row-major 64-bit words. The workgroup -> segment
    // map (common.hpp lookback_bwd_segment, round 6): groups of eight segments, the highest group first - the segments above a workgroup's
    // own belong to workgroups with smaller indices or to the up to seven right behind it, and with G a multiple of eight a (row, seg) sits
    // on XCD (row G + seg) % 8 as in the forward launch, so the x tiles and saved states it reads were last touched through the same L2
    // (profiles/r05/bwd_lookback_ab.log: a plainly reversed row cost this kernel AND the next step's forward kernel ~3 us each; round 5
    // kept the forward map and waited for ALL later workgroups of the row, which needs the row resident at once).
    const int row = SEG ? wg / G : wg, seg = SEG == 3 ? lookback_bwd_segment(wg % G, G) : SEG ? wg % G : 0;
    const int t0 = SEG ? seg * Tseg : 0, t1 = SEG ? (t0 + Tseg < nt ? t0 + Tseg : nt) : nt, nr = t1 - t0;   // this workgroup's tiles, walked t1 - 1 .. t0
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : row / C) * LY::TOTAL;
    const float* __restrict__ xr = x + (size_t)row * N;
    const float* __restrict__ gr = gy + (size_t)row * N;
    float* __restrict__ gxr = gx + (size_t)row * N;
    const int mb_in = wave * S * 4, mb_out = ((wave + 1) % W) * S * 4;
    float* pw_lds = lds + LDS_MB;
    float* tbg = pw_lds + LDS_PW + wave * REGION;
    float* tbx = tbg + IMG;
    float* tsi = tbx + IMG;
    float* tbo = tsi + IMG;
    // mailboxes zeroed; wave 0's inbox carries the sequence number its first tile (t1 - 1) waits for, with the adjoint state that enters
    // from above (zero at the end of the row)
    for (int i = threadIdx.x; i < LDS_MB; i += 64 * W) {
        float v = 0.f;
        if (i < S * 4) {
            const int comp = i & 3;
            if (comp == 2) v = __builtin_bit_cast(float, t1);
            else if (SEG == 1 && comp < 2) v = segstart[((size_t)row * G + seg) * (2 * S) + 2 * (i >> 2) + comp];
        }
        lds[i] = v;
    }
    for (int i = threadIdx.x; i < LDS_PW; i += 64 * W) pw_lds[i] = tb[LY::PWA + i];
    __syncthreads();
    const f4* pwa = reinterpret_cast<const f4*>(pw_lds);
    f2 Kreg[S];
#pragma unroll
    for (int k = 0; k < S; ++k) Kreg[k] = f2{0.f, 0.f};

    const unsigned a_x = __builtin_amdgcn_readfirstlane(lds_addr(tbx)), a_g = __builtin_amdgcn_readfirstlane(lds_addr(tbg)),
                   a_s = __builtin_amdgcn_readfirstlane(lds_addr(tsi));
    constexpr int SEQ2 = SEG == 3 ? 1 << 28 : 0;        // added to the mailboxes' sequence numbers of the pass (the pre-pass sweep used the plain ones)
    // states: image slot 64 q + lane is granule g = (lane % 4) ^ (lane / 16) of chunk 16 q + lane / 4 (swz_slot), granule = section pair;
    // the pad granules (g >= S / 2) fetch pair 0 again: finite numbers in columns of C that nobody reads
    const int sg_ = (lane & 3) ^ (lane >> 4), soff = ((sg_ < S / 2 ? sg_ : 0) * 64 + (lane >> 2)) * 4;
    auto issue_dma = [&](int tt, bool full, int parts = 7) {     // parts: 1 x, 2 gy, 4 saved states
        if (full && (parts & 1)) tile_dma_issue_swz(xr + (size_t)tt * TS, a_x, lane);
        if (full && (parts & 2)) tile_dma_issue_swz(gr + (size_t)tt * TS, a_g, lane);
        if (parts & 4) {
            const float* cs = carries + ((size_t)row * nt + tt) * (S * 128) + soff;
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16<!DASP_STATES_CACHED>(cs + 64 * q, a_s + 1024 * q);
        }
    };
    if constexpr (SEG == 3) {
        // (the pass's first x tile and saved states are on their way while the sweep and the look-back run: their images are not the sweep's)
        if (wave < nr) issue_dma(t1 - 1 - wave, tile_full<L>((long)(t1 - 1 - wave) * TS, N, vec), 5);
        // ---- the adjoint scan-only sweep over the segment's gy (sos_bwd_kernel<SEG = 2>'s loop with this kernel's table products), the
        //      look-back, and wave 0's inbox for the pass ----
        const unsigned tag = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, tb[LY::TAG])) ^ 0x5A5A0000u;     // (the forward pass's words carry the plain tag)
        unsigned long long* z64 = reinterpret_cast<unsigned long long*>(const_cast<float*>(segstart)) + (size_t)row * G * (2 * S);
        if (wave < nr && tile_full<L>((long)(t1 - 1 - wave) * TS, N, vec)) tile_dma_issue_swz(gr + (size_t)(t1 - 1 - wave) * TS, a_g, lane);
        for (int r = wave; r < nr; r += W) {
            const int t = t1 - 1 - r;
            int toff = 0;
            asm volatile("" : "+s"(toff));
            const float* __restrict__ tbl = tb + toff;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!tile_full<L>((long)t * TS, N, vec)) tile_global_to_swz_guarded(tbg, gr, (long)t * TS, N);
            float GYc[L], Z[L];
            lds_to_chunks_swz<L>(tbg, GYc, 63 - lane);
            pin(GYc);
            if (r + W < nr) tile_dma_issue_swz(gr + (size_t)(t - W) * TS, a_g, lane);      // (tiles below a row's last one are always full)
#pragma unroll
            for (int k = 0; k < S; ++k) {
                f2 z0 = f2{0.f, 0.f}, z1 = f2{0.f, 0.f};
#pragma unroll
                for (int n = 0; n < L; n += 2) {
                    const f2 xy = f2{GYc[n], GYc[n + 1]};
                    z0 = fma2_bcast<0>(TLD2(tbl + LY::GAT + (k * L + n) * 2), xy, z0);
                    z1 = fma2_bcast<1>(TLD2(tbl + LY::GAT + (k * L + n + 1) * 2), xy, z1);
                }
                const f2 z = z0 + z1;
                Z[2 * k] = z.x; Z[2 * k + 1] = z.y;
            }
#pragma unroll
            for (int c = 2 * S; c < L; ++c) Z[c] = 0.f;
            pin(Z);
            f2 lam0[S];
            MboxPeek pk;
            SCAN_PRIO(DASP_SCAN_PRIO);
            tile_scan<S, L>(Z, [](f2 v) { return v; }, lam0, tbl + LY::MCA, tbl + LY::PLA, tbl + LY::P64A, pwa, lane,
                [&](int i) { pk = mbox_peek(lds, mb_in + 4 * i); },
                [&](int i, f2& K) {
                    if (pk.seq == t + 1) K = f2{pk.a, pk.b};
                    else { float a, b; mbox_wait(lds, mb_in + 4 * i, t + 1, a, b); K = f2{a, b}; }
                },
                [&](int i, f2 Kn) {
                    if (t > t0) mbox_publish<63>(lds, mb_out + 4 * i, Kn.x, Kn.y, t);
                    else if (lane == 63) { lookback_publish(z64 + (size_t)seg * (2 * S) + 2 * i, Kn.x, tag); lookback_publish(z64 + (size_t)seg * (2 * S) + 2 * i + 1, Kn.y, tag); }
                });
            SCAN_PRIO(0);
        }
        __shared__ double lb_st[2][2 * S];
        // the pass's first gy tile is on its way as well now (x and the saved states since before the sweep); the look-back stages its words
        // in wave 0's scratch image, idle between tiles
        if (wave < nr) issue_dma(t1 - 1 - wave, tile_full<L>((long)(t1 - 1 - wave) * TS, N, vec), 2);
        lookback_start<S, W>(z64, seg, G, -1, fz.segtab_adj + (size_t)(row / C) * 2 * (2 * S) * (2 * S), tag, lds, t1 + SEQ2, pw_lds + LDS_PW + 3 * IMG, lb_st, fz.err, DASP_DEVERR_SOS_BWD);
        __syncthreads();
    }
    if (SEG != 3 && wave < nr) issue_dma(t1 - 1 - wave, tile_full<L>((long)(t1 - 1 - wave) * TS, N, vec));
    int stores_in_flight = 0;
    float Aop[4], AT[4], AO[4];
    chunk_table_operands<S, L>(tb + LY::GAT, Aop, lane);
    if (GX) cascade_map_operands<S, L>(tb + LY::YMA, LY::YMC, AT, AO, lane);
    double gsum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) gsum[i] = 0.0;

    for (int r = wave; r < nr; r += W) {
        const int t = t1 - 1 - r;
        int toff = 0;
        asm volatile("" : "+s"(toff));   // opaque uniform 0: keeps the scalar table loads inside the tile loop
        const float* __restrict__ tbl = tb + toff;
        const bool full = tile_full<L>((long)t * TS, N, vec);
        WIDE_PRIO(DASP_SCAN_PRIO);
        TRACE(16);
        if (full && GX && stores_in_flight == L / 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(L / 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!full) {
            tile_global_to_swz_guarded(tbx, xr, (long)t * TS, N);
            tile_global_to_swz_guarded(tbg, gr, (long)t * TS, N);
        }
        TRACE(17);
        const int cl = 63 - lane;        // lane l scans chunk 63 - l (sos_bwd_kernel)
        f4 Bg[4];
        float Rg[16], Rx[16], Rs[16];
        chunk_products_load(tbg, Bg, lane);
        gram_operands_load(tbg, Rg, lane);
        gram_operands_load(tbx, Rx, lane);
        gram_operands_load(tsi, Rs, lane);
        float GYc[L];
        lds_to_chunks_swz<L>(tbg, GYc, cl);
        pin(GYc);
        pin(Bg); pin(Rg); pin(Rx); pin(Rs);
        if (r + W < nr) issue_dma(t - W, tile_full<L>((long)(t - W) * TS, N, vec));     // the three images are in registers now
        TRACE(18);
        float Z[L];
        {   // zero-state chunk end states on the VALU (packed FMAs against the wave-uniform table in SGPRs, one section's 16 entries at a time,
            // the next section's loads in flight meanwhile): the 16 matrix-core products this replaces queued behind the other wave's bulk
            // products and came back through an LDS round trip - the longest phase of the tile after the scan
            f2 G[L];
#pragma unroll
            for (int n = 0; n < L; ++n) G[n] = TLD2(tbl + LY::GAT + 2 * n);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                f2 z0 = f2{0.f, 0.f}, z1 = f2{0.f, 0.f};
#pragma unroll
                for (int n = 0; n < L; n += 2) {
                    const f2 xy = f2{GYc[n], GYc[n + 1]};
                    z0 = fma2_bcast<0>(G[n], xy, z0);
                    z1 = fma2_bcast<1>(G[n + 1], xy, z1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < S) {
#pragma unroll
                    for (int n = 0; n < L; ++n) G[n] = TLD2(tbl + LY::GAT + ((k + 1) * L + n) * 2);
                }
                const f2 z = z0 + z1;
                Z[2 * k] = z.x; Z[2 * k + 1] = z.y;
            }
#pragma unroll
            for (int c = 2 * S; c < L; ++c) Z[c] = 0.f;
        }
        pin(Z); TRACE(25);
        // the half of the tile's products that does not need the scan: (gy x) and (gy states) blocks of C, TA gy of gx - they run on the
        // matrix cores while the VALU scans
        f4 cacc[4], oacc[4];
        const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};       // (the first product of every accumulator takes the constant: no zeroing moves)
        // product number idx = 3 j + type of the 48: (gy x) step j, (gy states) step j, TA gy term j
        auto early = [&](int idx) {
            const int j = idx / 3, ty = idx % 3;
            if (ty == 0) { cacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rg[j], Rx[j], j ? cacc[0] : zero4, 0, 0, 0); }
            else if (ty == 1) { cacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rg[j], Rs[j], j ? cacc[1] : zero4, 0, 0, 0); }
            else if (GX) oacc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(AT[j >> 2], Bg[j & 3][j >> 2], (j >> 2) ? oacc[j & 3] : zero4, 0, 0, 0);
        };
        TRACE(26);
        // ---- adjoint chunk end states: the scan runs from the last chunk to the first = ascending lanes (chunk 63 - lane) ----
        f2 lam[S];  // adjoint section order: i <-> forward section S-1-i
        {
            MboxPeek pk;
            SCAN_PRIO(DASP_SCAN_PRIO);
            tile_scan_h<S, L>(Z, [](f2 v) { return v; }, lam,
                tbl + LY::MCA, tbl + LY::PLA, tbl + LY::P64A, pwa, lane,
                [&](int i) { if (W > 1) pk = mbox_peek(lds, mb_in + 4 * i); },
                [&](int i, f2& K) {
                    if (W == 1) K = Kreg[i];
                    else if (pk.seq == t + 1 + SEQ2) K = f2{pk.a, pk.b};   // the last tile finds wave 0's inbox as initialised: sequence nt, carry 0
                    else { float a, b; mbox_wait(lds, mb_in + 4 * i, t + 1 + SEQ2, a, b); K = f2{a, b}; }
                },
                [&](int i, f2 Kn) {
                    if (W == 1) Kreg[i] = f2{read_lane(Kn.x, 63), read_lane(Kn.y, 63)};
                    else if (t > t0) mbox_publish<63>(lds, mb_out + 4 * i, Kn.x, Kn.y, t + SEQ2);
                },
                [&](int k, int p) {      // the 48 products above, dealt out over the 4 S hook points of the scan
                    constexpr int NPS = (48 + S - 1) / S, base = NPS / 4, extra = NPS % 4;
                    const int start = k * NPS + p * base + (p < extra ? p : extra), cnt = base + (p < extra ? 1 : 0);
#pragma unroll
                    for (int i = 0; i < NPS; ++i)
                        if (i < cnt && start + i < 48) early(start + i);
                });
        }
        SCAN_PRIO(0);
        pin(lam); TRACE(19);
        // ---- the other half: the adjoint states as one more [chunk][16] image (components 2 i + c; zeros beyond 2S) ----
        {
            float sc[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int c = state_comp_at<S>(p);
                sc[p] = c >= 0 ? ((c & 1) ? lam[c >> 1].y : lam[c >> 1].x) : 0.f;
            }
            chunks_to_lds_swz<L>(tbo, sc, cl);
        }
        {
            float Rl[16];
            gram_operands_load(tbo, Rl, lane);
            pin(Rl); TRACE(27);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                cacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rl[m], Rx[m], m ? cacc[2] : zero4, 0, 0, 0);
                cacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rl[m], Rs[m], m ? cacc[3] : zero4, 0, 0, 0);
            }
        }
        WIDE_PRIO(DASP_SCAN_PRIO);
        TRACE(28);
        if (GX) {
            f4 Bl[4];
            chunk_products_load(tbo, Bl, lane);
            pin(Bl);
#pragma unroll
            for (int q = 0; q < state_steps<S>(); ++q)          // (entries 4 k + 3 of the adjoint-state image are zeros for 2S <= 12: sos_tile.hpp state_pos)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(AO[q], Bl[c][q], oacc[c], 0, 0, 0);
            // (segmented rows: the launch ends with a hand-off - gx goes through the L2 so that its release finds nothing to write back)
            if (DASP_DIRECT_OUT && full) {
                mfma_granules_to_global(gxr + (size_t)t * TS, oacc, true, lane, SEG != 0);
            } else {
                mfma_granules_to_image(tbo, oacc, lane);
                if (full) tile_swz_to_global_full(tbo, gxr, (long)t * TS, true, lane, SEG != 0);
                else tile_swz_to_global_guarded(tbo, gxr, (long)t * TS, N, SEG != 0);
            }
            stores_in_flight = full ? L / 4 : 0;
        }
        TRACE(23);
        // the tile's fp32 block sums -> the row's fp64 sums
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) gsum[4 * b + e] += (double)cacc[b][e];
        TRACE(24);
    }
    // the W waves' sums -> one matrix per row (per (row, segment))
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (LDS-DMA never targets an image after the last tile, but the regions change hands below)
    __syncthreads();
    double* red = reinterpret_cast<double*>(pw_lds + LDS_PW);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave * (REGION / 2) + i * 64 + lane] = gsum[i];
    __syncthreads();
    if constexpr (SEG != 0) {
        if (fz.on) {        // the finalize step inside the launch: this workgroup's lag sums from its own matrix (gram_fused_tail)
            const int item = row / C;
            gram_fused_tail<S>(fz, item, C * G, (row - item * C) * G + seg, gram + (size_t)item * C * G * 32, red, REGION / 2, reinterpret_cast<double*>(lds));
            return;
        }
    }
    for (int e = threadIdx.x; e < 1024; e += 64 * W) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) s += red[w * (REGION / 2) + e];
        gram[(SEG ? (size_t)row * G + seg : (size_t)row) * 1024 + e] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// signal.biquad as a call of its own (dasp_pytorch/signal.py:242-306): the same fp64 design the prep kernel runs, one thread per
// (item, control): thread (item, dir) writes its Jacobian column, dir 0 also the coefficients [b0 b1 b2 1 a1 a2] (normalised by a0).
__global__ void biquad_design_kernel(const double* __restrict__ gain_db, const double* __restrict__ cutoff, const double* __restrict__ qf,
                                     int n, int type, double sample_rate, double* __restrict__ ba, double* __restrict__ jac) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * n) return;
    const int item = idx / 3, dir = idx % 3;
    double c5[5], dc5[5];
    rbj_design(type, sample_rate, gain_db[item], cutoff[item], qf[item], dir, c5, dc5);
    for (int c = 0; c < 5; ++c) jac[(size_t)item * 15 + c * 3 + dir] = dc5[c];
    if (dir == 0) {
        double* o = ba + (size_t)item * 6;
        o[0] = c5[0]; o[1] = c5[1]; o[2] = c5[2]; o[3] = 1.0; o[4] = c5[3]; o[5] = c5[4];
    }
}
// gparams[item][dir] = sum_c gba[item][c'] jac[item][c][dir], c over b0 b1 b2 a1 a2 (a0 is the constant 1)
__global__ void biquad_backward_kernel(const double* __restrict__ jac, const double* __restrict__ gba, int n, double* __restrict__ gparams) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * n) return;
    const int item = idx / 3, dir = idx % 3;
    const double* g = gba + (size_t)item * 6;
    const double g5[5] = {g[0], g[1], g[2], g[4], g[5]};
    double v = 0.0;
    for (int c = 0; c < 5; ++c) v += g5[c] * jac[(size_t)item * 15 + c * 3 + dir];
    gparams[(size_t)item * 3 + dir] = v;
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
constexpr int kL = 16;    // samples per lane chunk
#ifndef DASP_FWD_W
#define DASP_FWD_W 8
#endif
constexpr int kWF = DASP_FWD_W;    // waves per row, forward (2 rows per CU -> 4 waves per SIMD, <= 128 VGPRs; measured best)
#ifndef DASP_BWD_W
#define DASP_BWD_W 4
#endif
constexpr int kWB = DASP_BWD_W;    // waves per row, Gram-matrix backward (2 rows per CU -> 2 waves per SIMD)
#ifndef DASP_BWD_W_ADJ
#define DASP_BWD_W_ADJ 8
#endif
constexpr int kWBA = DASP_BWD_W_ADJ;   // ... of the adjoint-only kernel (~100 registers, 8 KiB of LDS per wave - the forward kernel's shape; with kWB waves it
                                       // ran at 2 waves per SIMD and was latency-bound)

// At most one row per CU (B * C <= 256 rows, one workgroup each): twice the waves per row - the same waves per CU as two rows of the
// ordinary width, on one row (one workgroup per CU fits: LDS 134 / 139 KiB). Rows are latency-bound there: (64..128, 2, 131072) EQ steps
// took the same 0.23 ms as a 256-row batch; cutting rows into segments pays up to 128 rows (dasp_sos_segment_tiles), not above.
inline bool wide_rows(long rows) { return rows <= 256; }

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// squarings that take Phi^L (what the prep kernel has after its chunk tables) to Phi^(64 L Tseg), one segment of Tseg tiles
inline int seg_extra_squarings(long Tseg) {
    int n = 0;
    for (long v = 64L * Tseg; v > 1; v >>= 1) ++n;
    return n;
}

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

#ifdef DASP_TRACE
int dasp_debug_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dasp::g_trace), sizeof(long long) * 64); }
#endif

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
long dasp_sos_carry_floats(long rows, long N, int S) { return rows * dasp_sos_num_tiles(N) * 2 * S * 64; }
long dasp_sos_partial_floats(long rows, int S) { (void)S; return rows * 2048; }     // one 32 x 32 fp64 Gram matrix per row (per (row, segment))

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

/* The same design from 3 S separate control vectors (host array of device pointers, [3 k + dir] -> Bs values): what
 * functional.parametric_eq receives (functional.py:118-139), without packing them first. */
// want_basis (with Tseg > 0): a backward pass will follow - two more waves leave the basis responses of the Gram finalize step behind the
// segment matrices (dasp_sos_segtab_doubles)
static int peq_prepare_rows_impl(const float* const* rows, int Bs, int S, const int* types, double sample_rate, float* tab, double* dtab,
                                 long Tseg, double* segtab, int want_basis, void* stream) {
    if (!rows || !types || !tab || !dtab || Bs <= 0 || S > 8 || S <= 0 || (Tseg > 0 && (!segtab || (Tseg & (Tseg - 1))))) return DASP_ERR_ARG;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        PeqSpec spec = {};
        for (int i = 0; i < S; ++i) {
            if (types[i] < 0 || types[i] > 4) return DASP_ERR_ARG;
            spec.types[i] = types[i];
        }
        for (int i = 0; i < 3 * S; ++i) {
            if (!rows[i]) return DASP_ERR_ARG;
            spec.rows[i] = rows[i];
        }
        spec.sample_rate = sample_rate;
        double* basis = Tseg > 0 && want_basis ? segtab + (size_t)Bs * 2 * (2 * SS) * (2 * SS) : nullptr;
        hipLaunchKernelGGL((sos_prep_kernel<SS, kL>), dim3(basis ? 2 * Bs : Bs), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, spec, tab, dtab,
                           Tseg > 0 ? seg_extra_squarings(Tseg) : 0, Tseg > 0 ? segtab : nullptr, basis, Bs);
        return check_launch();
    });
}
int dasp_peq_prepare_rows(const float* const* rows, int Bs, int S, const int* types, double sample_rate, float* tab, double* dtab,
                          void* stream) {
    return peq_prepare_rows_impl(rows, Bs, S, types, sample_rate, tab, dtab, 0, nullptr, 0, stream);
}

int dasp_sosfilt_forward(const float* tab, int Bs, const float* x, float* y, float* carries, int B, int C, long N,
                         int S, void* stream) {
    if (!tab || !x || !y || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B)) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N);
    const int vec = (N % 4 == 0) && aligned16(x) && aligned16(y);
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        if (wide_rows(B * C))
            hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, 2 * kWF>), dim3(B * C), dim3(128 * kWF), 0, (hipStream_t)stream, tab,
                               Bs == 1 && B != 1, x, y, carries, C, (int)N, nt, vec);
        else
            hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF>), dim3(B * C), dim3(64 * kWF), 0, (hipStream_t)stream, tab,
                               Bs == 1 && B != 1, x, y, carries, C, (int)N, nt, vec);
        return check_launch();
    });
}

// gx == null: no input gradient; partials == null: no coefficient gradients (the adjoint-only kernel: x and carries are not read).
// With coefficient gradients the pass is sos_bwd_gram_kernel and leaves one Gram matrix per row in `partials`.
int dasp_sosfilt_backward_ex(const float* tab, int Bs, const float* x, const float* gy, const float* carries, float* gx,
                             float* partials, int B, int C, long N, int S, void* stream) {
    if (!tab || !gy || (!gx && !partials) || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B)) return DASP_ERR_ARG;
    if (partials && (!x || !carries || !aligned16(partials))) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && aligned16(gy) && (!x || aligned16(x)) && (!gx || aligned16(gx));
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipStream_t st = (hipStream_t)stream;
        if (!partials) {
            hipLaunchKernelGGL((sos_bwd_kernel<SS, kL, kWBA, 0>), dim3(B * C), dim3(64 * kWBA), 0, st, tab, bc, gy, gx, C, (int)N, nt, vec);
            return check_launch();
        }
        const dim3 g(B * C), b(64 * kWB), b2(128 * kWB);
        double* gm = reinterpret_cast<double*>(partials);
        const bool wide = wide_rows(B * C);
        if (wide && !gx)
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, 2 * kWB, BWD_NOGX>), g, b2, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec);
        else if (wide)
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, 2 * kWB, 0>), g, b2, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec);
        else if (!gx)
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, BWD_NOGX>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec);
        else
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, 0>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec);
        return check_launch();
    });
}

int dasp_sosfilt_backward(const float* tab, int Bs, const float* x, const float* gy, const float* carries, float* gx,
                          float* partials, int B, int C, long N, int S, void* stream) {
    if (!x || !carries || !gx || !partials) return DASP_ERR_ARG;
    return dasp_sosfilt_backward_ex(tab, Bs, x, gy, carries, gx, partials, B, C, N, S, stream);
}

// mode 0: gout (B,S,6) = dL/dsos ; mode 1: gout (B,S,3) = dL/d(gain_db, cutoff_freq, q_factor); mode 2: the same as (3S, B) rows.
// partials: one Gram matrix per (row, segment) - `segments` of them per row as written by the *_seg entry points, 1 for the plain ones.
static int grad_finalize_impl(const double* dtab, int Bs, const float* partials, int B, int C, int S, int segments, int mode, float* gout, void* stream) {
    if (!dtab || !partials || !gout || B <= 0 || C <= 0 || (Bs != 1 && Bs != B) || mode < 0 || mode > 2 || segments <= 0)
        return DASP_ERR_ARG;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipLaunchKernelGGL((sos_gram_finalize_kernel<SS>), dim3(B), dim3(256), 0, (hipStream_t)stream, dtab, Bs == 1 && B != 1,
                           reinterpret_cast<const double*>(partials), B, C * segments, mode, gout);
        return check_launch();
    });
}

int dasp_sos_grad_finalize_ex(const double* dtab, int Bs, const float* partials, int B, int C, int S, int segments, int mode,
                              float* gout, void* stream) {
    return grad_finalize_impl(dtab, Bs, partials, B, C, S, segments, mode, gout, stream);
}

int dasp_sos_grad_finalize(const double* dtab, int Bs, const float* partials, int B, int C, int S, int mode,
                           float* gout, void* stream) {
    return grad_finalize_impl(dtab, Bs, partials, B, C, S, 1, mode, gout, stream);
}

// dasp_sosfilt_backward followed by dasp_sos_grad_finalize (same mode / gout) as one call (two launches: fusing the finalize step into the
// backward kernel of one-workgroup-per-row launches was measured at 2 - 3 us of 425 and is not built; segmented rows - dasp_peq_backward -
// do finalize inside their launch).
int dasp_sosfilt_backward_grads_ex(float* tab, const double* dtab, int Bs, const float* x, const float* gy, const float* carries,
                                   float* gx, float* partials, int mode, float* gout, int B, int C, long N, int S, void* stream) {
    if (mode < 0 || mode > 2 || (partials && (!dtab || !gout))) return DASP_ERR_ARG;
    const int rc = dasp_sosfilt_backward_ex(tab, Bs, x, gy, carries, gx, partials, B, C, N, S, stream);
    if (rc != DASP_OK || !partials) return rc;
    return grad_finalize_impl(dtab, Bs, partials, B, C, S, 1, mode, gout, stream);
}

int dasp_sosfilt_backward_grads(float* tab, const double* dtab, int Bs, const float* x, const float* gy, const float* carries,
                                float* gx, float* partials, int mode, float* gout, int B, int C, long N, int S, void* stream) {
    if (!gx || !partials) return DASP_ERR_ARG;
    return dasp_sosfilt_backward_grads_ex(tab, dtab, Bs, x, gy, carries, gx, partials, mode, gout, B, C, N, S, stream);
}

// ---- segmented rows -------------------------------------------------------------------------------------------------------------
// For few rows (a row is one workgroup) every row can be cut into segments of Tseg tiles that run as independent workgroups:
// dasp_sos_segment_tiles proposes Tseg (a power of two; 0 = do not segment), dasp_sos_segment_prepare builds the per-item segment
// transition matrices from dtab, and the *_seg entry points run scan-only pre-pass, chain kernel and the ordinary pass. segbuf:
// dasp_sos_seg_floats(rows, N, S, Tseg) floats of scratch; partials: dasp_sos_partial_floats(rows * segments, S).
long dasp_sos_segment_tiles(long rows, long N) {
    const long nt = dasp_sos_num_tiles(N);
    // (more than 128 rows: one workgroup per row at twice the waves - wide_rows - is faster. Round 4 drew the line at 64 rows, when a
    // segmented step was five launches; with three (round 5) segments pay up to 128 rows: (40 / 48 / 64, 2, 131072) 0.173 -> 0.158 /
    // 0.159 / 0.170 ms, and lose beyond: (96, 2, 131072) 0.180 -> 0.259 - profiles/r05/seg_crossover.log)
    if (rows <= 0 || rows > 128 || nt < 16) return 0;
    long T = 8;                                        // at least one tile per forward wave
    // two workgroups per CU: measured at (8 / 16 / 32, 2, 131072) forward + backward 0.107 / 0.114 / 0.142 ms with this rule against 0.106 /
    // 0.114 / 0.162 ms with up to four per CU and 0.12 / 0.12 / 0.167 with one (profiles/r02/segment_length_sweep.log)
    while (rows * ((nt + T - 1) / T) > 512 && T < nt) T *= 2;
    return (nt + T - 1) / T > 1 ? T : 0;
}
long dasp_sos_segments(long N, long Tseg) { return Tseg > 0 ? (dasp_sos_num_tiles(N) + Tseg - 1) / Tseg : 1; }
// per item: the segment transition matrices of both systems; behind the Bs items' matrices, per item, the basis responses of the Gram
// finalize step (GramFin<S>::BASIS doubles: written by the design launch of dasp_peq_forward*, read by dasp_peq_backward)
static long sos_basis_doubles(int S) {      // = GramFin<S>::BASIS: FW in whole 16-row blocks, FG, FO
    const long D = 16 + 2 * S, NPB = (S * 18 + 15) / 16;
    return (D / 4) * NPB * 64 + 2 * D * S * 16;
}
long dasp_sos_segtab_doubles(int S) { return 2L * (2 * S) * (2 * S) + sos_basis_doubles(S); }
long dasp_sos_seg_floats(long rows, long N, int S, long Tseg) { return 2 * rows * dasp_sos_segments(N, Tseg) * 2 * S; }

int dasp_sos_segment_prepare(const double* dtab, int Bs, int S, long Tseg, double* segtab, void* stream) {
    if (!dtab || !segtab || Bs <= 0 || Tseg <= 0 || (Tseg & (Tseg - 1))) return DASP_ERR_ARG;
    int nsq = 0;
    for (long n = 64L * kL * Tseg; n > 1; n >>= 1) ++nsq;   // Phi^(64 L Tseg): log2 squarings
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipLaunchKernelGGL((sos_segprep_kernel<SS>), dim3(Bs), dim3(256), 0, (hipStream_t)stream, dtab, nsq, segtab);
        return check_launch();
    });
}

int dasp_sosfilt_forward_seg(const float* tab, const double* segtab, int Bs, const float* x, float* y, float* carries, float* segbuf,
                             int B, int C, long N, int S, long Tseg, void* stream) {
    if (!tab || !segtab || !x || !y || !segbuf || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || Tseg <= 0) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), G = (int)dasp_sos_segments(N, Tseg), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && aligned16(x) && aligned16(y);
    float* z = segbuf;
    float* start = segbuf + (size_t)B * C * G * 2 * S;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipStream_t st = (hipStream_t)stream;
        // scan-only pre-pass; its last workgroup per item chains the segments (chain_by_last_workgroup: the counter word lives in the table)
        hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF, 2>), dim3(B * C * G), dim3(64 * kWF), 0, st, tab, bc, x, (float*)nullptr,
                           (float*)nullptr, C, (int)N, nt, vec, G, (int)Tseg, (const float*)nullptr, z, const_cast<float*>(tab), segtab, start);
        hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF, 1>), dim3(B * C * G), dim3(64 * kWF), 0, st, tab, bc, x, y, carries, C, (int)N, nt,
                           vec, G, (int)Tseg, (const float*)start, (float*)nullptr);
        return check_launch();
    });
}

// The segmented forward pass of dasp_peq_forward* as ONE launch (sos_fwd_kernel<SEG = 3>: scan-only sweep, look-back over the row's earlier
// segments, output sweep). Only behind a design launch of the same call - that launch draws the tag the look-back words are validated
// with (LY::TAG); dasp_sosfilt_forward_seg, whose tables may serve many calls, keeps the two launches. segbuf: dasp_sos_seg_floats
// floats, used as rows x segments x 2S 64-bit words. -DDASP_FWD_LOOKBACK=0 builds the two-launch path here as well (developer A/B).
#ifndef DASP_FWD_LOOKBACK
#define DASP_FWD_LOOKBACK 1
#endif
#ifndef DASP_BWD_LOOKBACK
#define DASP_BWD_LOOKBACK 1      // the same for the backward pass of dasp_peq_backward (sos_bwd_gram_kernel<SEG = 3>); 0: pre-pass launch + pass
#endif
static int sosfilt_forward_lookback(const float* tab, const double* segtab, int Bs, const float* x, float* y, float* carries, float* segbuf,
                                    int B, int C, long N, int S, long Tseg, void* stream) {
    if (!DASP_FWD_LOOKBACK || !lookback_enabled()) return dasp_sosfilt_forward_seg(tab, segtab, Bs, x, y, carries, segbuf, B, C, N, S, Tseg, stream);
    if (error_pending()) return DASP_ERR_DEVICE;
    unsigned* err = error_words_device();
    if (!tab || !segtab || !x || !y || !segbuf || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || Tseg <= 0 || (reinterpret_cast<uintptr_t>(segbuf) & 7)) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), G = (int)dasp_sos_segments(N, Tseg), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && aligned16(x) && aligned16(y);
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF, 3>), dim3(B * C * G), dim3(64 * kWF), 0, (hipStream_t)stream, tab, bc, x, y, carries, C, (int)N, nt,
                           vec, G, (int)Tseg, (const float*)nullptr, segbuf, (float*)nullptr, segtab, (float*)nullptr, err);
        return check_launch();
    });
}

// The first two launches of dasp_sosfilt_forward_seg on their own: scan-only pre-pass + chain. segbuf (dasp_sos_seg_floats floats) then
// holds, in its second half, the state every (row, segment) starts from, [row][segment][2S] - for callers that run their own per-segment
// pass from those states (chainfwd.hip: the fused EQ -> compressor forward).
int dasp_sos_segment_starts(const float* tab, const double* segtab, int Bs, const float* x, float* segbuf, int B, int C, long N, int S,
                            long Tseg, void* stream) {
    if (!tab || !segtab || !x || !segbuf || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || Tseg <= 0) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), G = (int)dasp_sos_segments(N, Tseg), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && aligned16(x);
    float* z = segbuf;
    float* start = segbuf + (size_t)B * C * G * 2 * S;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL((sos_fwd_kernel<SS, kL, kWF, 2>), dim3(B * C * G), dim3(64 * kWF), 0, st, tab, bc, x, (float*)nullptr,
                           (float*)nullptr, C, (int)N, nt, vec, G, (int)Tseg, (const float*)nullptr, z, const_cast<float*>(tab), segtab, start);
        return check_launch();
    });
}

// gx / partials as in dasp_sosfilt_backward_ex
// fin_dtab != null (with one table per item, the basis responses behind segtab's matrices - dasp_peq_forward* - and coefficient gradients
// asked for): every (row, segment) workgroup turns its Gram matrix into lag sums and the one that completes an item's count maps their sum
// to the gradients gout (mode as in dasp_sos_grad_finalize_ex) - no finalize launch (gram_fused_tail)
static int sosfilt_backward_seg_impl(const float* tab, const double* segtab, int Bs, const float* x, const float* gy, const float* carries,
                                     float* gx, float* partials, float* segbuf, int B, int C, long N, int S, long Tseg,
                                     const double* fin_dtab, int fin_mode, float* fin_gout, void* stream) {
    if (!tab || !segtab || !gy || (!gx && !partials) || !segbuf || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || Tseg <= 0) return DASP_ERR_ARG;
    if (partials && (!x || !carries || !aligned16(partials))) return DASP_ERR_ARG;
    if (N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), G = (int)dasp_sos_segments(N, Tseg), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && aligned16(gy) && (!x || aligned16(x)) && (!gx || aligned16(gx));
    float* z = segbuf;
    float* start = segbuf + (size_t)B * C * G * 2 * S;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        hipStream_t st = (hipStream_t)stream;
        const bool fuse = partials && fin_dtab && fin_gout && !bc;
        const void* lb_kernel = !gx ? reinterpret_cast<const void*>(sos_bwd_gram_kernel<SS, kL, kWB, BWD_NOGX, 3>) : reinterpret_cast<const void*>(sos_bwd_gram_kernel<SS, kL, kWB, 0, 3>);
        if (fuse && DASP_BWD_LOOKBACK && lookback_enabled() && lookback_has_room(lb_kernel, 64 * kWB)) {
            if (error_pending()) return DASP_ERR_DEVICE;
            // one launch: adjoint scan-only sweep, look-back over the segments above, Gram pass, finalize (sos_bwd_gram_kernel<SEG = 3>)
            if (GramFin<SS>::BASIS != sos_basis_doubles(SS) || (reinterpret_cast<uintptr_t>(segbuf) & 7)) return DASP_ERR_UNSUPPORTED;
            GramFuse fz = {};
            fz.on = 1; fz.B = B; fz.mode = fin_mode; fz.dtab = fin_dtab; fz.gout = fin_gout;
            fz.basis = segtab + (size_t)Bs * 2 * (2 * SS) * (2 * SS);
            fz.cnt_tab = const_cast<float*>(tab);
            fz.segtab_adj = segtab + (2 * SS) * (2 * SS);
            fz.lb_words = reinterpret_cast<unsigned long long*>(segbuf);
            fz.err = error_words_device();
            const dim3 g(B * C * G), b(64 * kWB);
            double* gm = reinterpret_cast<double*>(partials);
            if (!gx)
                hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, BWD_NOGX, 3>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec, G, (int)Tseg, (const float*)segbuf, fz);
            else
                hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, 0, 3>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec, G, (int)Tseg, (const float*)segbuf, fz);
            return check_launch();
        }
        // adjoint scan-only pre-pass; its last workgroup per item chains the segments (chain_by_last_workgroup: the counter word lives in the table)
        // (kWBA waves: a segment of eight tiles is one tile per wave - the scan-only pass is a latency chain, not a throughput loop)
        hipLaunchKernelGGL((sos_bwd_kernel<SS, kL, kWBA, 2>), dim3(B * C * G), dim3(64 * kWBA), 0, st, tab, bc, gy, (float*)nullptr, C, (int)N, nt, vec, G, (int)Tseg,
                           (const float*)nullptr, z, const_cast<float*>(tab), segtab, start);
        if (!partials) {
            hipLaunchKernelGGL((sos_bwd_kernel<SS, kL, kWBA, 1>), dim3(B * C * G), dim3(64 * kWBA), 0, st, tab, bc, gy, gx, C, (int)N, nt, vec, G, (int)Tseg,
                               (const float*)start, (float*)nullptr);
            return check_launch();
        }
        if (GramFin<SS>::BASIS != sos_basis_doubles(SS)) return DASP_ERR_UNSUPPORTED;        // (the size query and the layout are two statements of one number)
        double* gm = reinterpret_cast<double*>(partials);
        GramFuse fz = {};
        if (fuse) {
            fz.on = 1; fz.B = B; fz.mode = fin_mode; fz.dtab = fin_dtab; fz.gout = fin_gout;
            fz.basis = segtab + (size_t)Bs * 2 * (2 * SS) * (2 * SS);
            fz.cnt_tab = const_cast<float*>(tab);
        }
        const dim3 g(B * C * G), b(64 * kWB);
        if (!gx)
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, BWD_NOGX, 1>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec, G, (int)Tseg, (const float*)start, fz);
        else
            hipLaunchKernelGGL((sos_bwd_gram_kernel<SS, kL, kWB, 0, 1>), g, b, 0, st, tab, bc, x, gy, carries, gx, gm, C, (int)N, nt, vec, G, (int)Tseg, (const float*)start, fz);
        return check_launch();
    });
}
int dasp_sosfilt_backward_seg_ex(const float* tab, const double* segtab, int Bs, const float* x, const float* gy, const float* carries,
                                 float* gx, float* partials, float* segbuf, int B, int C, long N, int S, long Tseg, void* stream) {
    return sosfilt_backward_seg_impl(tab, segtab, Bs, x, gy, carries, gx, partials, segbuf, B, C, N, S, Tseg, nullptr, 0, nullptr, stream);
}

int dasp_sosfilt_backward_seg(const float* tab, const double* segtab, int Bs, const float* x, const float* gy, const float* carries,
                              float* gx, float* partials, float* segbuf, int B, int C, long N, int S, long Tseg, void* stream) {
    if (!x || !carries || !gx || !partials) return DASP_ERR_ARG;
    return dasp_sosfilt_backward_seg_ex(tab, segtab, Bs, x, gy, carries, gx, partials, segbuf, B, C, N, S, Tseg, stream);
}

// dasp_sos_grad_finalize for partial sums produced by dasp_sosfilt_backward_seg with `segments` segments per row
int dasp_sos_grad_finalize_seg(const double* dtab, int Bs, const float* partials, int B, int C, int S, int segments, int mode,
                               float* gout, void* stream) {
    return grad_finalize_impl(dtab, Bs, partials, B, C, S, segments, mode, gout, stream);
}

// ---- one call per direction for functional.parametric_eq (functional.py:118-272) ------------------------------------------------------
// Forward: RBJ design from the 3 S control vectors (dasp_peq_prepare_rows) + the cascade. Tseg > 0 takes the segmented-row path
// (segtab / segbuf as for dasp_sosfilt_forward_seg; Tseg = dasp_sos_segment_tiles(B * C, N) or 0).
int dasp_peq_forward(const float* const* rows, int Bp, int S, const int* types, double sample_rate, float* tab, double* dtab,
                     const float* x, float* y, float* carries, int B, int C, long N, long Tseg, double* segtab, float* segbuf,
                     void* stream) {
    // (segmented rows: the segment transition matrices come out of the design launch - no dasp_sos_segment_prepare launch)
    const int rc = peq_prepare_rows_impl(rows, Bp, S, types, sample_rate, tab, dtab, Tseg > 0 ? Tseg : 0, segtab, carries != nullptr, stream);
    if (rc != DASP_OK) return rc;
    if (Tseg <= 0) return dasp_sosfilt_forward(tab, Bp, x, y, carries, B, C, N, S, stream);
    return sosfilt_forward_lookback(tab, segtab, Bp, x, y, carries, segbuf, B, C, N, S, Tseg, stream);
}

// The same from the normalised (Bp, 3 S) parameter tensor of Processor.process_normalized (dasp_pytorch/modules.py:25-91): de-normalisation
// (lo / span: host arrays of 3 S doubles, min and max - min per column, modules.py:136-155), range check (flag: one device word, bit i set
// when column i leaves [0, 1]; zero it before the call, read it back to raise the reference's ValueError; NULL = no check), design and
// cascade in the two launches of dasp_peq_forward. dasp_peq_backward with mode 1 then returns the gradient w.r.t. the normalised tensor.
// dasp_peq_prepare_norm is the design step on its own (tables only).
static int peq_prepare_norm_impl(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                                 unsigned* flag, float* tab, double* dtab, long Tseg, double* segtab, int want_basis, void* stream) {
    if (!pn || !types || !lo || !span || !tab || !dtab || Bp <= 0 || S > 8 || S <= 0 || (Tseg > 0 && (!segtab || (Tseg & (Tseg - 1))))) return DASP_ERR_ARG;
    return dispatch_S(S, [&](auto s) {
        constexpr int SS = decltype(s)::value;
        PeqSpec spec = {};
        for (int i = 0; i < S; ++i) {
            if (types[i] < 0 || types[i] > 4) return DASP_ERR_ARG;
            spec.types[i] = types[i];
        }
        for (int i = 0; i < 3 * S; ++i) { spec.lo[i] = lo[i]; spec.span[i] = span[i]; }
        spec.sample_rate = sample_rate;
        spec.norm = 1;
        spec.flag = flag;
        double* basis = Tseg > 0 && want_basis ? segtab + (size_t)Bp * 2 * (2 * SS) * (2 * SS) : nullptr;
        hipLaunchKernelGGL((sos_prep_kernel<SS, kL>), dim3(basis ? 2 * Bp : Bp), dim3(256), 0, (hipStream_t)stream, nullptr, pn, spec, tab, dtab,
                           Tseg > 0 ? seg_extra_squarings(Tseg) : 0, Tseg > 0 ? segtab : nullptr, basis, Bp);
        return check_launch();
    });
}
/* Tseg > 0 (a power of two) with segtab: the design launch also leaves the segment transition matrices of dasp_sos_segment_prepare. */
int dasp_peq_prepare_norm_seg(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                              unsigned* flag, float* tab, double* dtab, long Tseg, double* segtab, void* stream) {
    return peq_prepare_norm_impl(pn, Bp, S, types, sample_rate, lo, span, flag, tab, dtab, Tseg, segtab, 0, stream);
}
int dasp_peq_prepare_norm(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                          unsigned* flag, float* tab, double* dtab, void* stream) {
    return peq_prepare_norm_impl(pn, Bp, S, types, sample_rate, lo, span, flag, tab, dtab, 0, nullptr, 0, stream);
}
int dasp_peq_forward_norm(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                          unsigned* flag, float* tab, double* dtab, const float* x, float* y, float* carries, int B, int C, long N, long Tseg,
                          double* segtab, float* segbuf, void* stream) {
    const int rc = peq_prepare_norm_impl(pn, Bp, S, types, sample_rate, lo, span, flag, tab, dtab, Tseg > 0 ? Tseg : 0, segtab, carries != nullptr, stream);
    if (rc != DASP_OK) return rc;
    if (Tseg <= 0) return dasp_sosfilt_forward(tab, Bp, x, y, carries, B, C, N, S, stream);
    return sosfilt_forward_lookback(tab, segtab, Bp, x, y, carries, segbuf, B, C, N, S, Tseg, stream);
}

// Backward of the same call: adjoint cascade + control gradients (mode as in dasp_sos_grad_finalize), the tables being the ones
// dasp_peq_forward filled (designed cascade: the monic / identity kernel). gx == null: no input gradient; partials == null: none for
// the controls. Tseg / segtab / segbuf as in the forward call (partials then hold dasp_sos_partial_floats(rows * segments, S) floats).
int dasp_peq_backward(float* tab, const double* dtab, int Bp, const float* x, const float* gy, const float* carries, float* gx,
                      float* partials, int mode, float* gout, int B, int C, long N, int S, long Tseg, const double* segtab,
                      float* segbuf, void* stream) {
    if (Tseg <= 0) return dasp_sosfilt_backward_grads_ex(tab, dtab, Bp, x, gy, carries, gx, partials, mode, gout, B, C, N, S, stream);
    // one table per item: the finalize step runs inside the Gram pass (gram_fused_tail); a shared table (Bp == 1 < B) takes the separate launch
    const bool fuse = partials && dtab && gout && Bp == B && mode >= 0 && mode <= 2;
    const int rc = sosfilt_backward_seg_impl(tab, segtab, Bp, x, gy, carries, gx, partials, segbuf, B, C, N, S, Tseg, fuse ? dtab : nullptr, mode,
                                             fuse ? gout : nullptr, stream);
    if (rc != DASP_OK || !partials || fuse) return rc;
    return grad_finalize_impl(dtab, Bp, partials, B, C, S, (int)dasp_sos_segments(N, Tseg), mode, gout, stream);
}

// ---- signal.biquad (dasp_pytorch/signal.py:242-306) ------------------------------------------------------------------------------
// gain_db, cutoff_freq, q_factor: n fp64 values each; type as in dasp_peq_prepare. ba: (n, 6) fp64 rows [b0 b1 b2 1 a1 a2];
// jac: (n, 15) fp64, d(b0 b1 b2 a1 a2)/d(gain_db, cutoff_freq, q_factor), kept for dasp_biquad_backward.
int dasp_biquad_design(const double* gain_db, const double* cutoff_freq, const double* q_factor, int n, int type, double sample_rate,
                       double* ba, double* jac, void* stream) {
    if (!gain_db || !cutoff_freq || !q_factor || !ba || !jac || n <= 0 || type < 0 || type > 4) return DASP_ERR_ARG;
    hipLaunchKernelGGL(biquad_design_kernel, dim3((3 * n + 127) / 128), dim3(128), 0, (hipStream_t)stream, gain_db, cutoff_freq, q_factor, n,
                       type, sample_rate, ba, jac);
    return check_launch();
}
// gba: (n, 6) fp64 gradient w.r.t. the rows of ba -> gparams (n, 3) fp64 = gradient w.r.t. [gain_db, cutoff_freq, q_factor]
int dasp_biquad_backward(const double* jac, const double* gba, int n, double* gparams, void* stream) {
    if (!jac || !gba || !gparams || n <= 0) return DASP_ERR_ARG;
    hipLaunchKernelGGL(biquad_backward_kernel, dim3((3 * n + 127) / 128), dim3(128), 0, (hipStream_t)stream, jac, gba, n, gparams);
    return check_launch();
}

}  // extern "C"
