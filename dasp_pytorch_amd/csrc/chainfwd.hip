// Fused forward of the first two stages of the reference's effect chain: parametric EQ -> compressor (+ the chain's final gain,
// which the caller folds into the make-up gain), for gfx950.
//
// The reference runs `equalizer -> compressor -> reverb -> gain` with gradients for the prediction (examples/style_transfer.py:150-154)
// and, every training step, the same chain once more without gradients to synthesise the target (examples/style_transfer.py:293-299).
// Unfused, the EQ writes its output y1 (4 B per channel-sample) and the compressor reads it back twice (side chain, output stage):
// 16 B per channel-sample for the two forwards. Here one workgroup owns a batch item - both channels, because the compressor's side
// chain is their sum (functional.py:328) - and a tile of the EQ's output never leaves the chip: per 1024-sample tile a wave runs the
// cascade of channel 0, then of channel 1 (sos_tile.hpp: LDS-DMA tile image, chunk products on the matrix cores, lane scan, carries
// from wave to wave through LDS mailboxes), forms the side chain in registers, runs the gain computer (dyn_common.hpp), the one-pole
// smoothing as a per-lane recursion + lane scan in the EQ's chunk layout (16 consecutive samples per lane: factor alpha^16 between lanes,
// alpha^1024 between tiles, a third mailbox chain), multiplies both channels and stores: 8 B per channel-sample, no saved states
// (the no-gradient pass: the chunk states and tile carries the backward kernels need are not written; SAVE = true - dasp_chain_forward_saving,
// the training pass from ~200 items on - writes them and the EQ's output: 15 B per channel-sample).
// Few batch items: every item is cut into segments of Tseg tiles that run as independent workgroups, as in sosfilt.hip / dynamics.hip;
// the compressor's segment start state depends on the EQ's output, so the order is: EQ scan-only pre-pass + chain (sosfilt.hip,
// dasp_sos_segment_starts), this kernel with the compressor scan-only (SEG 2: EQ from its segment start state, nothing stored), the
// scalar chain of the smoothing state, this kernel again from both start states (SEG 1).
#include "sos_tile.hpp"
#include "dyn_common.hpp"

extern "C" long dasp_sos_num_tiles(long N);
extern "C" long dasp_sos_segments(long N, long Tseg);
extern "C" long dasp_sos_seg_floats(long rows, long N, int S, long Tseg);
extern "C" int dasp_sos_segment_starts(const float* tab, const double* segtab, int Bs, const float* x, float* segbuf, int B, int C, long N, int S,
                                       long Tseg, void* stream);

#ifndef DASP_CHAIN_MFMA_OUT
#define DASP_CHAIN_MFMA_OUT 1      // the EQ half's per-chunk cascade on the matrix cores (0: the per-lane recursion)
#endif

namespace dasp {

// alpha^(16 m) for a per-lane m < 128, fp64 repeated squaring
__device__ __forceinline__ float alpha_pow16(double alpha, int m) {
    double p = 1.0, s = alpha;
    s = s * s; s = s * s; s = s * s; s = s * s;   // alpha^16
    for (int bit = 0; bit < 7; ++bit) {
        if (m & (1 << bit)) p *= s;
        s *= s;
    }
    return (float)p;
}

// MODE: 0 compressor, 1 expander (dyn_common.hpp). SEG: 0 = one workgroup per item; 1 = one workgroup per (item, segment), EQ from
// segstart_eq[row][segment][2S], smoothing state from segstart_dyn[item][segment]; 2 = the same EQ pass with the compressor scan-only:
// nothing stored, the zero-state end of the segment's smoothing state goes to zseg_dyn[item][segment].
template <int S, int L, int W, int MODE, int SEG, bool SAVE = false>
__global__ void __launch_bounds__(64 * W, (W + 3) / 4)   // one workgroup of W = 16 waves per CU: four waves per SIMD, <= 128 registers
chain_fwd_kernel(const float* __restrict__ tab, int tab_bcast, const float* __restrict__ x, const float* __restrict__ ctl, float* __restrict__ y,
                 int C, int N, int nt, int vec, double sample_rate, float eps, int G, int Tseg, const float* __restrict__ segstart_eq,
                 const float* __restrict__ segstart_dyn, float* __restrict__ zseg_dyn, float* __restrict__ chain_tab = nullptr,
                 float* __restrict__ chain_start = nullptr, float* __restrict__ yeq = nullptr, float* __restrict__ carries = nullptr,
                 float* __restrict__ dyn_carries = nullptr, int nt_dyn = 0) {
    using LY = SosLayout<S, L>;
    static_assert(L == 16, "chunk layout of the smoothing scan");
    constexpr int S2 = 2 * S, TS = 64 * L, IMG = 64 * L, CH = 2;
    constexpr int LDS_MB = W * CH * S * 4, LDS_MD = W * 4, LDS_PW = S * 64 * 4, LDS_CF = S * 8, LDS_T = W * 2 * IMG;
    __shared__ __attribute__((aligned(16))) float lds[LDS_MB + LDS_MD + LDS_PW + LDS_CF + LDS_T];
    const int lane = lane_id(), wave = wave_id();
    const int b = SEG ? blockIdx.x / G : blockIdx.x, seg = SEG ? blockIdx.x % G : 0;
    const int t0 = SEG ? seg * Tseg : 0, t1 = SEG ? (t0 + Tseg < nt ? t0 + Tseg : nt) : nt;
    const float* __restrict__ tb = tab + (size_t)(tab_bcast ? 0 : b) * LY::TOTAL;
    const float* __restrict__ xb = x + (size_t)b * C * N;
    float* __restrict__ yb = y + (size_t)b * C * N;
    float* md_lds = lds + LDS_MB;
    float* pw_lds = md_lds + LDS_MD;
    float* cf_lds = pw_lds + LDS_PW;
    float* tbx = cf_lds + LDS_CF + wave * 2 * IMG;     // x image of the current pass, then (LDS-DMA) of the next one
    float* tby = tbx + IMG;                             // scratch of the chunk products, then the y image on its way out
    // EQ mailboxes [wave][channel][section]{v0, v1, seq, -}: zero, except that wave 0's inboxes hold what its first tile waits for
    for (int i = threadIdx.x; i < LDS_MB + LDS_MD; i += 64 * W) {
        float v = 0.f;
        if (SEG && i < CH * S * 4) {
            const int c = i / (S * 4), k = (i >> 2) % S, comp = i & 3;
            if (comp == 2) v = __builtin_bit_cast(float, t0);
            else if (comp < 2 && c < C) v = segstart_eq[(((size_t)b * C + c) * G + seg) * S2 + 2 * k + comp];
        }
        lds[i] = v;
    }
    for (int i = threadIdx.x; i < LDS_PW; i += 64 * W) pw_lds[i] = tb[LY::PW + i];
    for (int i = threadIdx.x; i < S * 8; i += 64 * W) cf_lds[i] = tb[LY::COEF + i];
    __syncthreads();
    const f4* pws = reinterpret_cast<const f4*>(pw_lds);

    // per-item constants of the compressor; the powers of alpha the chunk layout needs, in fp64 once per thread
    const DynItem it = load_item(ctl, b, sample_rate, eps);
    double a_d;
    {
        const double nat = sample_rate * ((double)ctl[(size_t)b * 5 + 2] / 1e3);
        a_d = exp(-2.1972245773362196 / nat);
    }
    const float q16 = alpha_pow16(a_d, 1), q32 = alpha_pow16(a_d, 2), q64 = alpha_pow16(a_d, 4), q128 = alpha_pow16(a_d, 8), q1024 = alpha_pow16(a_d, 64);
    const float dpw16 = alpha_pow16(a_d, (lane & 15) + 1), dpw32 = alpha_pow16(a_d, (lane & 31) + 1), dpws = alpha_pow16(a_d, lane);
    float Kdyn = SEG == 1 ? segstart_dyn[(size_t)b * G + seg] : 0.f;

    const int mb_in = wave * CH * S * 4, mb_out = ((wave + 1) % W) * CH * S * 4;
    const int md_in = wave * 4, md_out = ((wave + 1) % W) * 4;
    const unsigned a_x = __builtin_amdgcn_readfirstlane(lds_addr(tbx));
    if (t0 + wave < t1 && tile_full<L>((long)(t0 + wave) * TS, N, vec)) tile_dma_issue_swz(xb + (size_t)(t0 + wave) * TS, a_x, lane);
    int stores_in_flight = 0;
    float Aop[4];
    chunk_table_operands<S, L>(tb + LY::GT, Aop, lane);
    // the EQ's cascade over every chunk on the matrix cores (sos_tile.hpp cascade_outputs_mfma, as sos_fwd_kernel); the outputs come back
    // into the chunk layout the compressor half works in
    // (one workgroup per item only: with segmented items - few items, every launch latency-bound - the two more LDS round trips per channel
    // cost more than the recursion they replace: (16,1,262144) 0.0647 -> 0.0708 ms, against 0.1747 -> 0.1595 ms at (256,2,131072))
    constexpr bool MO = DASP_CHAIN_MFMA_OUT && SEG == 0 && L == 16 && S2 <= 16;
    float AT[4], AO[4];
    if (MO) cascade_map_operands<S, L>(tb + LY::YM, LY::YMC, AT, AO, lane);

    for (int t = t0 + wave; t < t1; t += W) {
        int toff = 0;
        asm volatile("" : "+s"(toff));   // opaque uniform 0: keeps the scalar table loads inside the tile loop (no SGPR spills)
        const float* __restrict__ tbl = tb + toff;
        const bool full = tile_full<L>((long)t * TS, N, vec);
        float Y[CH][L];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c >= C) {
#pragma unroll
                for (int n = 0; n < L; ++n) Y[c][n] = 0.f;
                continue;
            }
            const float* __restrict__ xr = xb + (size_t)c * N;
            float X[L];
            WIDE_PRIO(DASP_SCAN_PRIO);
            // the image of this pass was requested one pass ago; vmcnt is in order: the previous tile's y stores were issued before the
            // request of channel 0's image and may stay in flight, the request of channel 1's image came after them
            // (SAVE: the previous pass's seven state / EQ-output stores were issued behind this image's request as well)
            if (full) wait_vmcnt(SAVE ? (c == 0 ? (stores_in_flight < 0 ? 0 : stores_in_flight) : 7) : c == 0 ? stores_in_flight : 0);
            else tile_global_to_swz_guarded(tbx, xr, (long)t * TS, N);
            if (!MO) lds_to_chunks_swz<L>(tbx, X, lane);
            f4 Bop[4], zacc[4];
            chunk_products_load(tbx, Bop, lane);
            if (!MO) pin(X);
            pin(Bop);
            {   // next pass: the other channel of this tile, or channel 0 of this wave's next tile
                const int tn = c + 1 < C ? t : t + W, cn = c + 1 < C ? c + 1 : 0;
                if (tn < t1 && tile_full<L>((long)tn * TS, N, vec)) tile_dma_issue_swz(xb + (size_t)cn * N + (size_t)tn * TS, a_x, lane);
            }
            float Z[L];
            chunk_products_issue(Bop, Aop, zacc);
            chunk_products_collect<L>(tby, zacc, Z, lane, lane);
            pin(Z);
            f2 st[S];
            MboxPeek pk;
            SCAN_PRIO(DASP_SCAN_PRIO);
            tile_scan<S, L>(Z, [](f2 v) { return v; }, st, tbl + LY::MC, tbl + LY::PL, tbl + LY::P64, pws, lane,
                [&](int k) { pk = mbox_peek(lds, mb_in + (c * S + k) * 4); },
                [&](int k, f2& K) {
                    if (pk.seq == t) K = f2{pk.a, pk.b};   // tile 0 finds the zero-initialised inbox: sequence 0, carry 0
                    else { float a, bb; mbox_wait(lds, mb_in + (c * S + k) * 4, t, a, bb); K = f2{a, bb}; }
                },
                [&](int k, f2 Kn) { if (t + 1 < t1) mbox_publish<63>(lds, mb_out + (c * S + k) * 4, Kn.x, Kn.y, t + 1); });
            SCAN_PRIO(0);
            if constexpr (SAVE) {        // the chunk start states as sos_fwd_kernel saves them for the EQ's backward pass
                f4* cs = reinterpret_cast<f4*>(carries) + (((size_t)b * C + c) * nt + t) * (S / 2) * 64 + lane;
#pragma unroll
                for (int m = 0; m < S / 2; ++m) st_stream(cs + m * 64, f4{st[2 * m].x, st[2 * m].y, st[2 * m + 1].x, st[2 * m + 1].y});
            }
            if constexpr (MO) {
                cascade_outputs_mfma<S, L>(tby, st, Bop, AT, AO, lane);
                lds_to_chunks_swz<L>(tby, X, lane);
            } else
            // the cascade, one section at a time in place over the chunk (sos_fwd_kernel; normal form, coefficients in VGPRs)
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int oz = opaque_zero_after(X[0]);
                const f4 ca = *reinterpret_cast<const f4*>(cf_lds + k * 8 + oz);       // sg, om, kom, g1
                const f4 cb = *reinterpret_cast<const f4*>(cf_lds + k * 8 + 4 + oz);   // g2, d, kappa, -
                float s1 = st[k].x, s2 = st[k].y;
                const float nk = -ca.z;
#pragma unroll
                for (int n = 0; n < L; ++n) {
                    const float u = X[n];
                    X[n] = fmaf(ca.w, s1, fmaf(cb.x, s2, cb.y * u));
                    const float t1_ = fmaf(ca.x, s1, fmaf(nk, s2, u));
                    s2 = fmaf(ca.y, s1, ca.x * s2);
                    s1 = t1_;
                }
            }
#pragma unroll
            for (int n = 0; n < L; ++n) Y[c][n] = X[n];
            pin(Y[c]);
            if constexpr (SAVE) {        // ... and the EQ's output, which the compressor's backward pass recomputes its gain from
                float* __restrict__ er = yeq + ((size_t)b * C + c) * N;
                chunks_to_lds_swz<L>(tby, Y[c], lane);
                if (full) tile_swz_to_global_full(tby, er, (long)t * TS, true, lane);
                else tile_swz_to_global_guarded(tby, er, (long)t * TS, N);
            }
        }
        // ---- compressor on the tile while it is in registers (functional.py:325-399) ----
        WIDE_PRIO(DASP_SCAN_PRIO);
        float gc[L];
        float e = 0.f;
#pragma unroll
        for (int n = 0; n < L; ++n) {
            float d0, d1, d2, d3;
            const float s = Y[0][n] + Y[1][n];                                             // side chain: sum over channels (:328)
            const float x_db = DB_PER_LOG2 * log2f(fmaxf(fabsf(s), it.eps));               // :347
            gc[n] = it.beta * gain_computer<MODE, false>(x_db, it, d0, d1, d2, d3);        // (1 - alpha) g_c[n]
            e = fmaf(it.alpha, e, gc[n]);                                                  // zero-state end of the chunk
        }
        // inclusive scan over the 64 chunks: E_i = e_i + alpha^16 E_{i-1}
        e = fmaf(q16, dpp0<0x111, 0xf>(e), e);
        e = fmaf(q32, dpp0<0x112, 0xf>(e), e);
        e = fmaf(q64, dpp0<0x114, 0xf>(e), e);
        e = fmaf(q128, dpp0<0x118, 0xf>(e), e);
        e = fmaf(dpw16, dpp0<0x142, 0xa>(e), e);
        e = fmaf(dpw32, dpp0<0x143, 0xc>(e), e);
        float K;
        if (t == t0) K = Kdyn;                   // (only wave 0 sees t == t0: the state the item / segment starts from)
        else { float dummy; mbox_wait(lds, LDS_MB + md_in, t, K, dummy); }
        {
            const float Kn = fmaf(q1024, K, read_lane(e, 63));       // the only work on the cross-wave chain of the smoothing state
            if (t + 1 < t1) mbox_publish(lds, LDS_MB + md_out, Kn, 0.f, t + 1);
            else if (SEG == 2 && lane == 0) __hip_atomic_store(zseg_dyn + (size_t)b * G + seg, Kn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (SEG == 2) { WIDE_PRIO(0); continue; }
        float g = fmaf(dpws, K, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x138, 0xf, 0xf, true)));
        if constexpr (SAVE) {
            // the smoothing state entering the compressor kernels' own tiles (512 samples: lanes 0 and 32 of this 1024-sample tile), which
            // dyn_bwd_kernel recomputes the gain curve from (dynamics.hip: carries[item][tile])
            if ((lane & 31) == 0 && 2 * t + (lane >> 5) < nt_dyn) dyn_carries[(size_t)b * nt_dyn + 2 * t + (lane >> 5)] = g;
        }
#pragma unroll
        for (int n = 0; n < L; ++n) {
            g = fmaf(it.alpha, g, gc[n]);                                                  // :372-380 as a recursion
            const float lin = exp2f((g + it.makeup) * LOG2_PER_DB);                        // :388-391
            Y[0][n] *= lin;
            Y[1][n] *= lin;
        }
        int nst = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c >= C) continue;
            float* __restrict__ yr = yb + (size_t)c * N;
            chunks_to_lds_swz<L>(tby, Y[c], lane);
            if (full) tile_swz_to_global_full(tby, yr, (long)t * TS, true, lane);
            else tile_swz_to_global_guarded(tby, yr, (long)t * TS, N);
            nst += L / 4;
        }
        stores_in_flight = full ? nst + (SAVE ? 8 : 0) : -1;      // (SAVE: + the last channel's seven saves and the compressor carry)
        WIDE_PRIO(0);
    }
    if (SEG == 2 && chain_tab) {
        // The chain of the smoothing state over the item's segments, start(g + 1) = alpha^(samples per segment) start(g) + z(g) in fp64, by
        // the last of the item's workgroups to finish (as sosfilt.hip's chain_by_last_workgroup; the counter is word 3 of the item's table,
        // zeroed by the prep kernel, reset here) - no launch of its own.
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (device-scope relaxed atomics instead of fences: see chain_by_last_workgroup, sosfilt.hip)
        __syncthreads();
        const int item = tab_bcast ? 0 : b;
        int* cnt = reinterpret_cast<int*>(chain_tab + (size_t)item * LY::TOTAL + LY::CNT) + 3;
        if (threadIdx.x == 0) {
            s_last = handoff_arrive_is_last(cnt, tab_bcast ? (int)gridDim.x : G);
        }
        __syncthreads();
        if (s_last) {
            const int b0 = tab_bcast ? 0 : b, nb_ = tab_bcast ? (int)gridDim.x / G : 1;
            // z of the item(s) into LDS first (the tile images are idle), all loads in flight together: they bypass this XCD's caches,
            // and one inside the dependent chain cost ~2.5 us per segment
            float* zs = cf_lds + LDS_CF;
            constexpr int ZMAX = W * 2 * IMG;
            for (int bi0 = b0; bi0 < b0 + nb_; bi0 += ZMAX / G) {
                const int nbi = b0 + nb_ - bi0 < ZMAX / G ? b0 + nb_ - bi0 : ZMAX / G;
                __syncthreads();
                for (int e = threadIdx.x; e < nbi * G; e += 64 * W)
                    zs[e] = __hip_atomic_load(zseg_dyn + (size_t)bi0 * G + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                for (int i = threadIdx.x; i < nbi; i += 64 * W) {
                    const int bi = bi0 + i;
                    const double nat = sample_rate * ((double)ctl[(size_t)bi * 5 + 2] / 1e3);
                    const double a = exp(-2.1972245773362196 / nat * (double)Tseg * (double)TS);
                    double s = 0.0;
                    for (int g = 0; g < G; ++g) {
                        chain_start[(size_t)bi * G + g] = (float)s;
                        s = a * s + (double)zs[i * G + g];
                    }
                }
            }
        }
    }
}

}  // namespace dasp

// ================================================================================================
// C-ABI (include/dasp_hip.h)
using namespace dasp;

namespace {
constexpr int kL = 16, kWC = 16, kS = 6;
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int chk() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DASP_OK : (int)e;
}
}  // namespace

extern "C" {

/* Tiles per segment for dasp_chain_forward, 0 = one workgroup per item. A workgroup is 16 waves and fills a CU; below 128 items the items are
 * cut until about 256 workgroups exist (every wave at least one tile). */
long dasp_chain_segment_tiles(long B, long N) {
    const long nt = dasp_sos_num_tiles(N);
    if (B <= 0 || B >= 128 || nt < 2 * kWC) return 0;
    long T = kWC;
    while (B * ((nt + T - 1) / T) > 256 && T < nt) T *= 2;
    return (nt + T - 1) / T > 1 ? T : 0;
}
/* floats of segbuf for dasp_chain_forward with Tseg > 0 */
long dasp_chain_seg_floats(long B, long C, long N, int S, long Tseg) {
    if (Tseg <= 0) return 0;
    return dasp_sos_seg_floats(B * C, N, S, Tseg) + 2 * B * dasp_sos_segments(N, Tseg);
}

/* y = compressor(parametric_eq(x)) (mode 0; 1 = expander) in one pass over x, forward only.
 *   tab    : the EQ's tables as dasp_peq_prepare / dasp_peq_prepare_rows / dasp_peq_prepare_norm fill them (Bs = 1 or B items, S = 6 sections)
 *   ctl    : (B, 5) rows [threshold_db, ratio, attack_ms, knee_db, makeup_gain_db] as for dasp_dynamics_forward (no look-ahead)
 *   x, y   : (B, C, N), C = 1 or 2
 *   Tseg   : 0, or dasp_chain_segment_tiles(B, N) with segtab from dasp_sos_segment_prepare(dtab, Bs, S, Tseg, ...) and
 *            segbuf of dasp_chain_seg_floats(B, C, N, S, Tseg) floats */
int dasp_chain_forward(const float* tab, int Bs, const float* x, const float* ctl, float* y, int B, int C, long N, int S, int mode,
                       double sample_rate, float eps, long Tseg, const double* segtab, float* segbuf, void* stream) {
    if (!tab || !x || !ctl || !y || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || (mode != 0 && mode != 1)) return DASP_ERR_ARG;
    if (C > 2 || S != kS || N > 0x7fffffffL) return DASP_ERR_UNSUPPORTED;
    if (Tseg > 0 && (!segtab || !segbuf)) return DASP_ERR_ARG;
    const int nt = (int)dasp_sos_num_tiles(N), bc = Bs == 1 && B != 1;
    const int vec = (N % 4 == 0) && al16(x) && al16(y);
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(64 * kWC);
#define DASP_CHAIN_LAUNCH(MODE_, SEG_, grid, se, sd, zd)                                                                                    \
    hipLaunchKernelGGL((chain_fwd_kernel<kS, kL, kWC, MODE_, SEG_>), dim3(grid), blk, 0, st, tab, bc, x, ctl, y, C, (int)N, nt, vec, sample_rate, \
                       eps, G, (int)Tseg, se, sd, zd)
    if (Tseg <= 0) {
        const int G = 1;
        if (mode == 0) DASP_CHAIN_LAUNCH(0, 0, B, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
        else DASP_CHAIN_LAUNCH(1, 0, B, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
        return chk();
    }
    const int G = (int)dasp_sos_segments(N, Tseg);
    if (G > 4096) return DASP_ERR_UNSUPPORTED;        // (the chain of the smoothing state stages an item's G segment states in LDS; the planner proposes <= 256)
    const int rc = dasp_sos_segment_starts(tab, segtab, Bs, x, segbuf, B, C, N, S, Tseg, stream);      // EQ scan-only pre-pass + chain
    if (rc != DASP_OK) return rc;
    const float* start_eq = segbuf + (size_t)B * C * G * 2 * S;
    float* zd = segbuf + dasp_sos_seg_floats((long)B * C, N, S, Tseg);
    float* start_dyn = zd + (size_t)B * G;
    // compressor scan-only pass; its last workgroup per item chains the smoothing state over the segments
#define DASP_CHAIN_LAUNCH_PRE(MODE_)                                                                                                        \
    hipLaunchKernelGGL((chain_fwd_kernel<kS, kL, kWC, MODE_, 2>), dim3(B * G), blk, 0, st, tab, bc, x, ctl, y, C, (int)N, nt, vec, sample_rate,  \
                       eps, G, (int)Tseg, start_eq, (const float*)nullptr, zd, const_cast<float*>(tab), start_dyn)
    if (mode == 0) DASP_CHAIN_LAUNCH_PRE(0); else DASP_CHAIN_LAUNCH_PRE(1);
#undef DASP_CHAIN_LAUNCH_PRE
    if (mode == 0) DASP_CHAIN_LAUNCH(0, 1, B * G, start_eq, (const float*)start_dyn, (float*)nullptr);
    else DASP_CHAIN_LAUNCH(1, 1, B * G, start_eq, (const float*)start_dyn, (float*)nullptr);
#undef DASP_CHAIN_LAUNCH
    return chk();
}


/* dasp_chain_forward for the pass that carries gradients (examples/style_transfer.py:150-154): the same ONE pass over x that also writes
 * what the two backward passes read - the EQ's output yeq (B, C, N), which dasp_dynamics_backward takes as its input and recomputes the gain
 * curve from; the EQ's chunk start states eq_carries (dasp_sos_carry_floats(B * C, N, S) floats, as dasp_sosfilt_forward saves them) for
 * dasp_peq_backward; and the smoothing state entering every 512-sample compressor tile, dyn_carries (dasp_dyn_carry_floats(B, N) floats).
 * 15 B per channel-sample against 19 for the two forward calls. One workgroup per item (no segments): it pays from ~200 items on
 * ((256,2,131072) 0.272 -> 0.207 ms, (128,2,131072) 0.176 -> 0.183: profiles/r06/chain_fwd_saving_ab.log); the callers take it from 384 rows (inside the whole chain step the gain is 0.4 %: chain_step_ab.log). */
int dasp_chain_forward_saving(const float* tab, int Bs, const float* x, const float* ctl, float* y, float* yeq, float* eq_carries, float* dyn_carries,
                              int B, int C, long N, int S, int mode, double sample_rate, float eps, void* stream) {
    if (!tab || !x || !ctl || !y || !yeq || !eq_carries || !dyn_carries || B <= 0 || C <= 0 || N <= 0 || (Bs != 1 && Bs != B) || (mode != 0 && mode != 1))
        return DASP_ERR_ARG;
    if (C > 2 || S != kS || N > 0x7fffffffL - 1024) return DASP_ERR_UNSUPPORTED;
    const int nt = (int)dasp_sos_num_tiles(N), bc = Bs == 1 && B != 1, ntd = (int)((N + 511) / 512);
    const int vec = (N % 4 == 0) && al16(x) && al16(y) && al16(yeq);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0)
        hipLaunchKernelGGL((chain_fwd_kernel<kS, kL, kWC, 0, 0, true>), dim3(B), dim3(64 * kWC), 0, st, tab, bc, x, ctl, y, C, (int)N, nt, vec, sample_rate, eps, 1, 0,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, yeq, eq_carries, dyn_carries, ntd);
    else
        hipLaunchKernelGGL((chain_fwd_kernel<kS, kL, kWC, 1, 0, true>), dim3(B), dim3(64 * kWC), 0, st, tab, bc, x, ctl, y, C, (int)N, nt, vec, sample_rate, eps, 1, 0,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, yeq, eq_carries, dyn_carries, ntd);
    return chk();
}

}  // extern "C"
