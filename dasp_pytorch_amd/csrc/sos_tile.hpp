// Tile machinery of the cascaded-biquad scans, shared by sosfilt.hip (forward / backward cascades) and chainfwd.hip (the fused
// EQ -> compressor forward kernel): table layout, chunk products on the matrix cores, the wave-level lane scan with its LDS mailboxes.
// See the header of sosfilt.hip for the algorithm; executable fp64 specification: oracle/chunkscan_model.py.
#pragma once
#include "common.hpp"
#include <type_traits>

#ifndef DASP_FWD_NT
// forward kernel: streaming hints on 1 = x loads, 2 = y stores, 4 = state stores. The kernels run back to back (forward, backward,
// forward, ...) and share the 256 MB MALL, so only the pair can be judged. The signals (x, y: touched once per kernel) stream; the
// 201 MB of chunk states do not, on either side (DASP_STATES_CACHED: the backward kernel reads them with the normal policy too):
// they are read back last-written-first by the backward pass and overwritten in place by the next step's forward pass, and keeping
// them cache-resident took the forward kernel from 0.156 to 0.143 ms (same box; either half of the change alone: nothing).
#define DASP_FWD_NT 3
#endif

// Issue priority (s_setprio) of a wave while it is in a latency-bound phase of its tile - the lane scan (dependent DPP chains, few
// instructions) and, with DASP_PRIO_WIDE, the LDS round trips around it: it gets the issue slot whenever it is ready and the waves
// in their cascade phase (long runs of independent FMAs) fill the gaps. Same box, fwd + bwd: 0.432 ms without, 0.424 scan only,
// 0.420 wide.
#ifndef DASP_SCAN_PRIO
#define DASP_SCAN_PRIO 1
#endif
#define SCAN_PRIO(p) do { if (DASP_SCAN_PRIO) __builtin_amdgcn_s_setprio(p); } while (0)
#ifndef DASP_PRIO_WIDE
#define DASP_PRIO_WIDE 1     // 1: also the load / transposition and store phases of a tile, i.e. everything but the cascade
#endif
#define WIDE_PRIO(p) do { if (DASP_SCAN_PRIO && DASP_PRIO_WIDE) __builtin_amdgcn_s_setprio(p); } while (0)
#ifndef DASP_DIRECT_OUT
#define DASP_DIRECT_OUT 1         // forward / Gram backward kernels, full tiles: the matrix-core output granules go straight to memory (no staging image)
#endif

namespace dasp {

#ifdef DASP_TRACE   // developer builds only: cycle stamps of one wave's phases (tools/sosbench prints them)
static __device__ long long g_trace[64];
#define TRACE(i) do { if (blockIdx.x == 7 && threadIdx.x == 64 && t >= 40 && t < 40 + W) g_trace[i] = clock64(); } while (0)
#define TRACE2(i) do { if (trace_on && k == 3) g_trace[i] = clock64(); } while (0)
#define PTRACE(i, t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                          if (blockIdx.x == 7 && threadIdx.x == (t)) g_trace[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TRACE(i)
#define TRACE2(i)
#define PTRACE(i, t)
#endif


// ------------------------------------------------------------------------------------------------
// Per-item fp32 table layout (floats). S sections, chunk length L. Every 2x2 matrix is stored
// column-major (c00, c10, c01, c11) so that  f += col0 * t1 + col1 * t2  is two packed FMAs.
// "A" suffix = the adjoint system (sections in reverse order, transposed state matrices).
template <int S, int L>
struct SosLayout {
    static constexpr int S2 = 2 * S;
    static constexpr int COEF = 0;                   // [S][8]: sg, om, kom, g1, g2, d, kappa, direct-form flag (0 / 1)
    static constexpr int GT = COEF + S * 8;          // [S][L][2] forward chunk table, section-major
    static constexpr int MC = GT + L * S2;           // [S][S][4] blocks of Phi^L (j < k used)
    static constexpr int PL = MC + 4 * S * S;        // [S][4][4] M_kk^(2^l), l = 0..3
    static constexpr int P64 = PL + 16 * S;          // [S][4]    M_kk^64
    static constexpr int PW = P64 + 4 * S;           // [S][64][4] M_kk^(c+1), c = 0..63
    static constexpr int SYS = PW + 256 * S - GT;    // size of one system's block (GT..PW)
    static constexpr int GAT = GT + SYS;             // adjoint chunk table, natural sample order
    static constexpr int MCA = MC + SYS;
    static constexpr int PLA = PL + SYS;
    static constexpr int P64A = P64 + SYS;
    static constexpr int PWA = PW + SYS;
    static constexpr int DF = GT + 2 * SYS;          // [S][8]: b1, b2, -a1, -a2 (normalised), zc1, zc2, 1/om, sg/om: direct-form sections
    static constexpr int YMC = L + 16;               // columns of the output map below: L input samples, then up to 16 start-state components
    static constexpr int YM = DF + 8 * S;            // [L][YMC]: the chunk's L outputs as a linear map of (its L inputs, its 2S start-state
                                                     //      components): row n = (h[n], h[n-1], .., h[0], 0, .. | O[n][0..2S), 0, ..), h = impulse response
                                                     //      of the whole cascade, O = its zero-input response per unit state (forward kernel, MFMA output path)
    static constexpr int YMA = YM + L * YMC;         // [L][YMC]: the same for the adjoint cascade, natural sample order: row n = (0, .., h[0], h[1], .., h[L-1-n] |
                                                     //      OA[n][0..2S), 0, ..): the adjoint outputs (gx) of a chunk from its L adjoint inputs (gy) and
                                                     //      the 2S components of the adjoint state entering it from above (sos_bwd_gram_kernel)
    static constexpr int CNT = YMA + L * YMC;        // [8]: the item's counter words (ints; zeroed by the prep kernel, returned to zero by whoever completes a count):
                                                     //      0 backward finalize (gram_fused_tail), 1 forward chain, 2 adjoint chain (chain_by_last_workgroup), 3 the
                                                     //      fused forward chain's (chainfwd.hip); word TAG: the tag of this call's look-back words (sosfilt.hip
                                                     //      lookback_publish), drawn by the prep kernel
    static constexpr int TAG = CNT + 4;
    static constexpr int TOTAL = CNT + 8;
};
// fp64 side table for the finalize kernel, per (item, section)
// [DT_OM] om (1 for a direct-form section: its correlations are taken with w itself), [DT_B0..+4] b0 b1 b2 a1 a2 (normalised), [DT_A0] a0 as
// given, [7] kappa, [DT_J..+14] design Jacobian, [DT_SG32] sg as the kernels hold it (fp32), [DT_PK] product of the b0 of the sections before
// this one (scale of its signals in the monic recomputation), [DT_NF] 1 = the section's kept signal is om w (normal form), 0 = w (direct form)
constexpr int DT_OM = 0, DT_B0 = 1, DT_A1 = 4, DT_A0 = 6, DT_J = 8, DT_SG32 = 24, DT_PK = 25, DT_NF = 26, DT_STRIDE = 28;

// ------------------------------------------------------------------------------------------------
// shared pieces of the forward / backward tile code

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
// acc += g * (x, x) with x = one half of `xy`, selected by the instruction's op_sel bits: the table product needs every sample
// broadcast to both halves of a packed operand, and building that operand with two v_mov per sample costs as many issue slots as a
// sixth of the product itself. g is wave-uniform (SGPR pair).
template <int HALF> __device__ __forceinline__ f2 fma2_bcast(f2 g, f2 xy, f2 acc) {
    if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(g), "v"(xy));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(g), "v"(xy));
    return acc;
}
// the same with a per-lane (VGPR) g
template <int HALF> __device__ __forceinline__ f2 fma2_bcast_v(f2 g, f2 xy, f2 acc) {
    if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(g), "v"(xy));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g), "v"(xy));
    return acc;
}

// DPP move: value of the source lane selected by CTRL, 0 where there is none / the row is masked off
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
// the same, rows outside ROW_MASK keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
// lane i <- v[i-1]; lane 0 <- first
__device__ __forceinline__ float wave_shr1(float first, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// lane l <- v[63 - l]
__device__ __forceinline__ float wave_mirror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((63 - lane_id()) * 4, __builtin_bit_cast(int, v)));
}
// an integer 0 the compiler cannot prove uniform: loads addressed with it stay in VGPRs
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
// zero-cost scheduling edge: `a` is not available before `dep` has been computed
__device__ __forceinline__ void order_after(float& a, float dep) { asm volatile("" : "+v"(a) : "v"(dep)); }
// Phase fence: volatile asm statements keep their program order, so pinning every live value of a
// phase makes everything computed from them start after everything that produced them. Without
// it the compiler overlaps independent phases of a tile and their register live ranges add up.
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(f2& a) { float x = a.x, y = a.y; pin(x); pin(y); a = f2{x, y}; }
__device__ __forceinline__ void pin(f4& a) { asm volatile("" : "+v"(a)); }
template <typename T, int N>
__device__ __forceinline__ void pin(T (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) pin(a[i]);
}
template <typename T, int N, int M>
__device__ __forceinline__ void pin(T (&a)[N][M]) {
#pragma unroll
    for (int i = 0; i < N; ++i) pin(a[i]);
}
// the same, but ordered after `dep` has been computed (serialises otherwise independent phases so
// that their register live ranges do not overlap)
__device__ __forceinline__ int opaque_zero_after(float dep) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(dep));
    return z;
}

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate); anything unexpected waits for everything
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// z = sum_n T[n] * X[n] for one section: T = [L] f2 (wave-uniform -> scalar loads, packed FMAs);
// two accumulators halve the dependent chain.
#define TLD2(p) (*reinterpret_cast<const f2*>(p))
#define TLD4(p) (*reinterpret_cast<const f4*>(p))

// f += [c.x c.z; c.y c.w] * (t1, t2)   (column-major 2x2 block, packed FMAs)
// f + [[c.x, c.z], [c.y, c.w]] * t with t.x / t.y broadcast by op_sel: two issue slots per packed FMA and nothing else (the splat
// form below costs two extra v_mov per application). _s: wave-uniform block (SGPRs), _v: per-lane block.
__device__ __forceinline__ f2 blk_apply_s(f4 c, f2 t, f2 f) {
    return fma2_bcast<0>(f2{c.x, c.y}, t, fma2_bcast<1>(f2{c.z, c.w}, t, f));
}
__device__ __forceinline__ f2 blk_apply_v(f4 c, f2 t, f2 f) {
    return fma2_bcast_v<0>(f2{c.x, c.y}, t, fma2_bcast_v<1>(f2{c.z, c.w}, t, f));
}
__device__ __forceinline__ f2 blk_apply(f4 c, float t1, float t2, f2 f) {
    return fma2(f2{c.x, c.y}, splat(t1), fma2(f2{c.z, c.w}, splat(t2), f));
}

// bit k set: section k runs in direct form (prep kernel's decision, COEF[k][7]); wave-uniform
template <int S>
__device__ __forceinline__ unsigned direct_form_mask(const float* __restrict__ coef) {
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) m |= (coef[k * 8 + 7] != 0.f ? 1u : 0u) << k;
    return __builtin_amdgcn_readfirstlane(m);
}

// Zero-state chunk end states of all sections for the whole tile, on the matrix cores: z[2S x 64 chunks] = G[2S x L] * X[L x 64] is
// a small dense product whose left factor is the same for every tile of a row. As packed VALU FMAs it was L issue-slot pairs per
// section and tile (19 % of the forward kernel's VALU work, 10 % of the backward's); as 16 v_mfma_f32_16x16x4_f32 it runs beside
// the VALU. Operand layouts (16x16x4 f32): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[i][j] in lane 16 (i / 4) + j,
// register i % 4. With contraction slot (q, k) <-> sample 4 k + q, lane l's B operands of column block c are one granule of the
// swizzled tile image (chunk 16 c + l % 16, granule l / 16) and its D registers are one granule (rows 4 (l / 16) ..) of a [chunk][16
// rows] image with the same geometry, so both sides use the tile-image helpers.
//   chunk_table_operands: A registers, once per kernel. Row 2 k + comp of G is Gs[(k L + n) 2 + comp]; rows >= 2 S are zero.
//   chunk_products_load / _issue / _collect: img = this tile's swizzled input image (the operands are in registers after _load, so
//   the image can be handed to the next DMA), scratch = a free image of the wave; Z[2 k + comp] for chunk `chunk`.
template <int S, int L>
__device__ __forceinline__ void chunk_table_operands(const float* __restrict__ Gs, float (&A)[4], int lane) {
    static_assert(L == 16 && 2 * S <= 16, "one 16x16x4 block row");
    const int i = lane & 15, k = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) A[q] = i < 2 * S ? Gs[((i >> 1) * L + 4 * k + q) * 2 + (i & 1)] : 0.f;
}
__device__ __forceinline__ void chunk_products_load(const float* img, f4 (&Bv)[4], int lane) {
#pragma unroll
    for (int c = 0; c < 4; ++c) Bv[c] = *reinterpret_cast<const f4*>(img + 4 * swz_slot(16 * c + (lane & 15), lane >> 4));
}
__device__ __forceinline__ void chunk_products_issue(const f4 (&Bv)[4], const float (&A)[4], f4 (&acc)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q], Bv[c][q], acc[c], 0, 0, 0);
}
template <int L>
__device__ __forceinline__ void chunk_products_collect(float* scratch, const f4 (&acc)[4], float (&Z)[L], int lane, int chunk) {
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<f4*>(scratch + 4 * swz_slot(16 * c + (lane & 15), lane >> 4)) = acc[c];
    wave_lds_sync();
    lds_to_chunks_swz<L>(scratch, Z, chunk);
}

// The cascade over every chunk of a tile on the matrix cores (round 4). From its exact start state a chunk's L = 16 outputs are a linear
// map of its 16 inputs and its 2S start-state components, y = T x + O s0 (LY::YM: row n = (h[n], .., h[0], 0, .. | O[n][0 .. 2S), 0, ..),
// fp64 in the prep kernel), so the per-lane recursion - 768 vector instructions per tile for six sections - is 32 v_mfma_f32_16x16x4_f32
// with the operand geometry of the chunk products above: the B operands of T x are the input granules the chunk products already hold
// (Bx), those of O s0 the scan's start states written as one more [chunk][16] image (component c at entry state_pos(c); the other entries
// zeros) and read back the same way, and the D registers are granules of the output image: on return `img` holds the tile's outputs in the
// swizzled image layout (what chunks_to_lds_swz would have written). AT / AO: cascade_map_operands, once per kernel.
// Position of state component c in the [chunk][16] state images of the matrix-core maps. The contraction slot (step q, lane group k) of a
// 16x16x4 product is entry 4 k + q of the image row, so zero entries only save a step when they fill one q for every k: with 2S <= 12
// the components go three to a granule (entries 4 g .. 4 g + 2, entry 4 g + 3 zero) and step q = 3 of the state products is skipped.
template <int S> __host__ __device__ constexpr bool state_packed3() { return 2 * S <= 12; }
template <int S> __host__ __device__ constexpr int state_pos(int c) { return state_packed3<S>() ? 4 * (c / 3) + c % 3 : c; }
template <int S> __host__ __device__ constexpr int state_comp_at(int p) {     // component at entry p of the image row, -1 = a zero
    return state_packed3<S>() ? ((p % 4 == 3 || 3 * (p / 4) + p % 4 >= 2 * S) ? -1 : 3 * (p / 4) + p % 4) : (p < 2 * S ? p : -1);
}
template <int S> __host__ __device__ constexpr int state_steps() { return state_packed3<S>() ? 3 : 4; }

template <int S, int L>
__device__ __forceinline__ void cascade_map_operands(const float* __restrict__ ym, int ymc, float (&AT)[4], float (&AO)[4], int lane) {
    static_assert(L == 16 && 2 * S <= 16, "one 16x16 output block per 16 chunks");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        AT[q] = ym[(lane & 15) * ymc + 4 * (lane >> 4) + q];
        AO[q] = ym[(lane & 15) * ymc + L + 4 * (lane >> 4) + q];
    }
}
// _acc: the products only - yacc[c] = rows 4 (lane / 16) .. + 3 of chunk 16 c + lane % 16, i.e. one 16-byte granule of the tile per
// register quad (`img` is used for the state image). cascade_outputs_mfma: the same, then written to `img` as the output image.
// mfma_granules_to_global: the granules straight to memory - per instruction the wave writes 16 chunks x 64 B = 1 KiB contiguous, every
// 64-byte chunk by four lanes - no staging image, no LDS round trip (full tiles only; ragged tiles go through the image).
template <int S, int L>
__device__ __forceinline__ void cascade_outputs_mfma_acc(float* img, const f2 (&st)[S], const f4 (&Bx)[4], const float (&AT)[4], const float (&AO)[4], int lane,
                                                         f4 (&yacc)[4]);
__device__ __forceinline__ void mfma_granules_to_image(float* img, const f4 (&acc)[4], int lane) {
    wave_lds_sync();              // every lane has its operands before the image is overwritten with the outputs
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<f4*>(img + 4 * swz_slot(16 * c + (lane & 15), lane >> 4)) = acc[c];
    wave_lds_sync();
}
__device__ __forceinline__ void mfma_granules_to_global(float* __restrict__ tile, const f4 (&acc)[4], bool stream, int lane, bool through = false) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f4* p = reinterpret_cast<f4*>(tile + (16 * c + (lane & 15)) * 16 + 4 * (lane >> 4));
        if (through) st_through(p, acc[c]); else if (stream) st_stream(p, acc[c]); else *p = acc[c];
    }
}
template <int S, int L>
__device__ __forceinline__ void cascade_outputs_mfma(float* img, const f2 (&st)[S], const f4 (&Bx)[4], const float (&AT)[4], const float (&AO)[4], int lane) {
    f4 yacc[4];
    cascade_outputs_mfma_acc<S, L>(img, st, Bx, AT, AO, lane, yacc);
    mfma_granules_to_image(img, yacc, lane);
}
template <int S, int L>
__device__ __forceinline__ void cascade_outputs_mfma_acc(float* img, const f2 (&st)[S], const f4 (&Bx)[4], const float (&AT)[4], const float (&AO)[4], int lane,
                                                         f4 (&yacc)[4]) {
    float sc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int c = state_comp_at<S>(p);
        sc[p] = c >= 0 ? ((c & 1) ? st[c >> 1].y : st[c >> 1].x) : 0.f;
    }
    chunks_to_lds_swz<L>(img, sc, lane);
    f4 Bs[4];
    chunk_products_load(img, Bs, lane);
    pin(Bs);
#pragma unroll
    for (int c = 0; c < 4; ++c) yacc[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) yacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(AT[q], Bx[c][q], yacc[c], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < state_steps<S>(); ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) yacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(AO[q], Bs[c][q], yacc[c], 0, 0, 0);
}

// Whole-tile scan for one system (forward or adjoint tables): lane chunks X -> chunk start states st.
//   Gs   : [S][L][2] chunk table (zero-state end state of section k = sum_n Gs[k][n] X[n]); zmap is
//          applied to that per-lane value before the scan (identity, or the lane mirror for the adjoint)
//   MCs  : [S][S][4] coupling blocks, PLs : [S][4][4] M^(1,2,4,8), P64s : [S][4] M^64  (global, scalar loads)
//   pws  : [S][64] f4 per-lane powers M^(c+1) in LDS
//   carry_in(k, K)  : obtain the tile carry-in of section k (uniform)
//   carry_out(k, K) : hand the carry for the next tile on
// Per section: f = z_k + sum_{j<k} M_kj st_j; inclusive scan of f over the 64 lanes (four Kogge-Stone
// levels inside each 16-lane row on DPP row_shr, then row_bcast:15 / row_bcast:31 with per-lane
// powers); E = f + M^(lane+1) K; st = E shifted by one lane.
// The wave-uniform tables go through SGPRs. Scalar loads return after hundreds of cycles and a wave
// has nothing else to issue meanwhile, so they are software-pipelined by hand: each table buffer is
// refilled for section k+1 right after its last use in section k (the scheduling barriers pin the
// issue points), which keeps at most one section's worth (~72 SGPRs) live.
//   hook(k, p)      : p = 0..3, four points of section k's code (start, before the coupling sum, before the in-row levels, end) where
//          the caller may issue work that is independent of the scan - the backward kernel's matrix-core products, which then run
//          under the scan's dependent chains instead of in a phase of their own (tile_scan_h; tile_scan = no hook)
template <int S, int L, typename FMap, typename FPre, typename FIn, typename FOut, typename FHook>
__device__ __forceinline__ void tile_scan_h(const float (&Z)[L], FMap&& zmap, f2 (&st)[S],
                                            const float* __restrict__ MCs, const float* __restrict__ PLs,
                                            const float* __restrict__ P64s, const f4* __restrict__ pws, int lane,
                                            FPre&& carry_prefetch, FIn&& carry_in, FOut&& carry_out, FHook&& hook, bool trace_on = false) {
    (void)P64s;   // M^64 is lane 63's per-lane power
    f4 MC[S], PL[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) PL[l] = TLD4(PLs + 4 * l);
    // destinations of the two row-broadcast moves: their masked-off rows are never written and stay 0, so one zeroing per tile serves
    // all sections (a fresh zero "old" value per move cost four v_mov per section)
    float b16x = 0.f, b16y = 0.f, b32x = 0.f, b32y = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        hook(k, 0);
        // LDS reads of the section (per-lane powers, carry mailbox) are issued first and *waited for* right after the
        // table product, before any scalar refill is in flight: LDS and SMEM share lgkmcnt and SMEM returns out of
        // order, so a later LDS wait would be an lgkmcnt(0) that also waits for the refills just issued.
        TRACE2(8);
        f4 pw16 = pws[k * 64 + (lane & 15)], pw32 = pws[k * 64 + (lane & 31)], pw64 = pws[k * 64 + lane];
        carry_prefetch(k);
        f2 f = zmap(f2{Z[2 * k], Z[2 * k + 1]});
        { float a = pw16.x, b = pw32.x, c = pw64.x; pin(a); pin(b); pin(c); pw16.x = a; pw32.x = b; pw64.x = c; }
        TRACE2(9);
        __builtin_amdgcn_sched_barrier(0);
        hook(k, 1);
        // (the coupling sum on two accumulators - a shorter dependent chain, one more packed add - measured nothing: profiles/r02/ab_micro_variants.log)
#pragma unroll
        for (int j = 0; j < k; ++j) f = blk_apply_s(MC[j], st[j], f);
        pin(f); TRACE2(10);
        __builtin_amdgcn_sched_barrier(0);
        hook(k, 2);
        if (k + 1 < S) {
#pragma unroll
            for (int j = 0; j <= k; ++j) MC[j] = TLD4(MCs + ((k + 1) * S + j) * 4);
        }
        f = blk_apply_s(PL[0], f2{dpp0<0x111, 0xf>(f.x), dpp0<0x111, 0xf>(f.y)}, f);
        f = blk_apply_s(PL[1], f2{dpp0<0x112, 0xf>(f.x), dpp0<0x112, 0xf>(f.y)}, f);
        f = blk_apply_s(PL[2], f2{dpp0<0x114, 0xf>(f.x), dpp0<0x114, 0xf>(f.y)}, f);
        f = blk_apply_s(PL[3], f2{dpp0<0x118, 0xf>(f.x), dpp0<0x118, 0xf>(f.y)}, f);
        pin(f); TRACE2(11);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < S) {
#pragma unroll
            for (int l = 0; l < 4; ++l) PL[l] = TLD4(PLs + (k + 1) * 16 + 4 * l);
        }
        // rows 1, 3 += M^(j+1) * (last lane of the previous row); rows 2, 3 += M^((lane % 32) + 1) * lane 31
        b16x = dpp_keep<0x142, 0xa>(b16x, f.x); b16y = dpp_keep<0x142, 0xa>(b16y, f.y);
        f = blk_apply_v(pw16, f2{b16x, b16y}, f);
        b32x = dpp_keep<0x143, 0xc>(b32x, f.x); b32y = dpp_keep<0x143, 0xc>(b32y, f.y);
        f = blk_apply_v(pw32, f2{b32x, b32y}, f);
        pin(f); TRACE2(12);
        f2 K;
        carry_in(k, K);
        pin(K); TRACE2(13);
        // E = f + M^(lane+1) K: the chunk end states; lane 63's is the carry for the next tile and that lane hands it on (the only
        // work on the cross-wave serial chain; no readlane / SGPR round trip, no separate M^64 product)
        const f2 E = blk_apply_v(pw64, K, f);
        carry_out(k, E);
        st[k] = f2{wave_shr1(K.x, E.x), wave_shr1(K.y, E.y)};
        pin(st[k]); TRACE2(14);
        hook(k, 3);
    }
}
template <int S, int L, typename FMap, typename FPre, typename FIn, typename FOut>
__device__ __forceinline__ void tile_scan(const float (&Z)[L], FMap&& zmap, f2 (&st)[S],
                                          const float* __restrict__ MCs, const float* __restrict__ PLs,
                                          const float* __restrict__ P64s, const f4* __restrict__ pws, int lane,
                                          FPre&& carry_prefetch, FIn&& carry_in, FOut&& carry_out, bool trace_on = false) {
    tile_scan_h<S, L>(Z, zmap, st, MCs, PLs, P64s, pws, lane, carry_prefetch, carry_in, carry_out, [](int, int) {}, trace_on);
}

}  // namespace dasp
