// The Gram-matrix finalize step of the cascaded-biquad backward pass (csrc/sosfilt.hip): a row's (or an item's) 32 x 32 fp64 Gram matrix C
// -> coefficient / control gradients. Device code shared by sos_gram_finalize_kernel (a launch of its own), the segmented Gram pass
// (every workgroup's lag sums + the item's last workgroup, gram_fused_tail) and the design launch (basis responses). Included by
// sosfilt.hip inside namespace dasp, after sos_tile.hpp and the design-table constants (DT_STRIDE, DT_JAC).
#pragma once

// ------------------------------------------------------------------------------------------------
// g5 = dL/d(b0, b1, b2, a1, a2) of section k of the item (normalised coefficients) -> the requested gradients (mode as in dasp_sos_grad_finalize)
struct EmitCoef { double a0, b[5], J[15]; };      // of one (item, section): a0 as given, the normalised coefficients, the design Jacobian (dtab)
__device__ __forceinline__ EmitCoef load_emit_coef(const double* __restrict__ d) {
    EmitCoef e;
    e.a0 = d[DT_A0];
#pragma unroll
    for (int i = 0; i < 5; ++i) e.b[i] = d[DT_B0 + i];
#pragma unroll
    for (int i = 0; i < 15; ++i) e.J[i] = d[DT_J + i];
    return e;
}
__device__ __forceinline__ void emit_section_grads(const EmitCoef& e, const double (&g5)[5], int B, int S, int mode,
                                                   float* __restrict__ gout, int item, int k) {
    const int idx = item * S + k;
    if (mode == 0) {
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) dot += g5[i] * e.b[i];
        float* o = gout + (size_t)idx * 6;
        o[0] = (float)(g5[0] / e.a0); o[1] = (float)(g5[1] / e.a0); o[2] = (float)(g5[2] / e.a0);
        o[3] = (float)(-dot / e.a0);
        o[4] = (float)(g5[3] / e.a0); o[5] = (float)(g5[4] / e.a0);
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < 5; ++c) v += g5[c] * e.J[c * 3 + i];
            if (mode == 1) gout[(size_t)idx * 3 + i] = (float)v;
            else gout[(size_t)(3 * k + i) * B + item] = (float)v;
        }
    }
}
__device__ __forceinline__ void emit_section_grads(const double* __restrict__ d, const double (&g5)[5], int B, int S, int mode,
                                                   float* __restrict__ gout, int item, int k) {
    emit_section_grads(load_emit_coef(d), g5, B, S, mode, gout, item, k);
}


// ------------------------------------------------------------------------------------------------
// C (summed over the item's rows, or over its (row, segment) workgroups) -> the gradients of the item's sections. The steps are shared by
// sos_gram_finalize_kernel (one workgroup per item, a launch of its own: one workgroup per row, shared tables, the two-call C entry points)
// and by the last workgroup of an item in the segmented Gram pass (sos_bwd_gram_kernel<SEG = 1>, gram_fused_tail):
//   1. the per-chunk basis responses of the item's cascade in fp64, one thread per basis vector - FW[k][n][j] = w_k[n - 2] (the all-pole
//      signal of section k, n = 0 .. L + 1) for u = e_j; FG / FO[k][n][j] = adjoint input / output of section k at sample n for v = e_j.
//      Start states enter as the kernels define them: w[-2] = s2 / om, w[-1] = s1 + (sg / om) s2 (normal-form chunk start state);
//      z1 = l1, z2 = -sg l1 + om l2 (adjoint state, transposed direct form II). They depend on the item's coefficients only: in the
//      segmented pass a workgroup of its own computes them BESIDE the tile workgroups (gram_basis_responses<AGENT = true> into global
//      scratch), so that the 96-step fp64 chain - half of the finalize step's time - is not on the tail of the launch;
//   2. C summed over the matrices (gram_sum), P[k][m] = C FW[k][m] for the S (L + 2) signal rows on the fp64 matrix cores
//      (v_mfma_f64_16x16x4_f64: D[i][j] in lane 16 (i % 4) + j, register i / 4 - not the f32 instruction's row order;
//      tools/mfma64_probe.hip), then thread (which, k, n): the products of F[k][n] with P[k][n + 2 - j]: the lag sums sum g w[n],
//      sum g w[n - 1], sum g w[n - 2], sum o w[n - 1], sum o w[n - 2] after a 16-lane reduction over n;
//   3. thread k: dL/d(b0, b1, b2, a1, a2) = (the three g sums, minus the two o sums) -> emit_section_grads.
// AGENT: the operands were written by other workgroups of the running launch (relaxed agent-scope accesses, common.hpp hand-off).
// developer builds (-DDASP_TRACE): cycle stamps of the finalize steps of item 0 (tools/sosbench, scripts/seg_tail_trace.py)
#ifdef DASP_TRACE
#define GTRACE(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                       if (item == 0 && threadIdx.x == 0) g_trace[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GTRACE(i)
#endif
template <int S>
struct GramFin {
    static constexpr int L = 16, D = L + 2 * S, NW = L + 2, NP = S * NW, NPB = (NP + 15) / 16;
    // Basis responses (doubles), laid out as the finalize step reads them, so that every load of a wave is one contiguous 512 bytes (the
    // natural [row][column] layouts are strided gathers for it: 84 such loads took a wave 5k cycles just to issue,
    // profiles/r05/seg_tail_trace.log):
    //   FW[p][u], p = k (L + 2) + n < NP, u < D  at fw(p, u):  [u / 4][p / 16][u % 4][p % 16] - the B operands of one matrix-core step
    //                                                          (lane = 16 (u % 4) + p % 16) of one 16-row block are 64 consecutive doubles
    //   FG / FO[k][n][v]                          at fg(k, n, v), fo(..): [v][k L + n] - thread (k, n) of the lag sums reads column v
    static constexpr int FW = 0, FG = (D / 4) * NPB * 64, FO = FG + D * S * L, BASIS = FO + D * S * L;
    static_assert(D % 4 == 0, "whole matrix-core steps");
    __host__ __device__ static constexpr int fw(int p, int u) { return FW + (((u >> 2) * NPB + (p >> 4)) * 4 + (u & 3)) * 16 + (p & 15); }
    __host__ __device__ static constexpr int fg(int k, int n, int v) { return FG + v * (S * L) + k * L + n; }
    __host__ __device__ static constexpr int fo(int k, int n, int v) { return FO + v * (S * L) + k * L + n; }
    // work area in LDS (doubles): C[32][33] (C[v][u], zero beyond D), P[NP][33] (P[k (L + 2) + m][v] = sum_u C[v][u] FW[k][m][u]),
    // lag sums [S][5], coefficients [S][8] (b0 b1 b2 a1 a2 normalised, sg, om, 1 / om)
    static constexpr int CM = 0, P = 32 * 33, LAG = P + NP * 33, CF = LAG + S * 5 + (S * 5) % 2, WORK = CF + S * 8;
};
template <int S> __device__ __forceinline__ constexpr int basis_doubles() { return GramFin<S>::BASIS; }
template <bool AGENT> __device__ __forceinline__ double fin_ld(const double* p) {
    if (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool AGENT> __device__ __forceinline__ void fin_st(double* p, double v) {
    if (AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int S>
__device__ __forceinline__ void gram_fin_coefs(const double* __restrict__ d0, double* cf, int tid) {
    if (tid < S) {
        const double* d = d0 + tid * DT_STRIDE;
        const double a1 = d[DT_B0 + 3], a2 = d[DT_B0 + 4], sg = -0.5 * a1;
        double om = sqrt(fabs(sg * sg - a2));
        om = om < OM_MIN ? OM_MIN : om;
        double* c = cf + tid * 8;
        c[0] = d[DT_B0]; c[1] = d[DT_B0 + 1]; c[2] = d[DT_B0 + 2]; c[3] = a1; c[4] = a2;
        c[5] = sg; c[6] = om; c[7] = 1.0 / om;
    }
}
// threads 0 .. 127 (forward basis vectors: 0 .. D - 1, adjoint ones: 64 .. 64 + D - 1); cf must be visible (barrier) before the call
template <int S, bool AGENT>
__device__ __forceinline__ void gram_basis_responses(const double* cf, double* bas, int tid) {
    using GF = GramFin<S>;
    constexpr int L = GF::L, NW = GF::NW, D = GF::D;
    if (tid >= 128) return;
    const int j = tid & 63, adj = tid >> 6;
    if (j >= D) return;
    double sig[L];
#pragma unroll
    for (int n = 0; n < L; ++n) sig[n] = (j == n) ? 1.0 : 0.0;
    for (int i = 0; i < S; ++i) {
        const int k = adj ? S - 1 - i : i;
        const double* c = cf + k * 8;
        const double b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[3], a2 = c[4], sg = c[5], om = c[6], iom = c[7];
        const double c1 = (j == L + 2 * i) ? 1.0 : 0.0, c2 = (j == L + 2 * i + 1) ? 1.0 : 0.0;   // the unit state component, if it is this section's
        if (!adj) {
            double w2 = c2 * iom, w1 = c1 + sg * iom * c2;
            fin_st<AGENT>(bas + GF::fw(k * NW, j), w2); fin_st<AGENT>(bas + GF::fw(k * NW + 1, j), w1);
#pragma unroll
            for (int n = 0; n < L; ++n) {
                const double w = fma(-a1, w1, fma(-a2, w2, sig[n]));      // (one FMA on the recurrence's dependent chain)
                sig[n] = fma(b0, w, fma(b1, w1, b2 * w2));
                fin_st<AGENT>(bas + GF::fw(k * NW + n + 2, j), w);
                w2 = w1; w1 = w;
            }
        } else {
            double z1 = c1, z2 = -sg * c1 + om * c2;
#pragma unroll
            for (int n = L - 1; n >= 0; --n) {
                const double g = sig[n];
                fin_st<AGENT>(bas + GF::fg(k, n, j), g);
                const double o = fma(b0, g, z1);
                z1 = fma(-a1, o, fma(b1, g, z2));
                z2 = fma(-a2, o, b2 * g);
                fin_st<AGENT>(bas + GF::fo(k, n, j), o);
                sig[n] = o;
            }
        }
    }
}
// C[v][u] (v, u < 32; zero beyond D) = sum of the nmat matrices at g0, gathered from the kernel's register layout; rows 16.. are the
// entries of the adjoint-state image, component c at entry state_pos(c). 256 threads, four entries each; the matrices are fetched
// sixteen at a time (64 independent loads per thread in flight: one trip to memory per sixteen matrices), summed in index order.
template <int S, bool AGENT>
__device__ __forceinline__ void gram_sum(const double* __restrict__ g0, int nmat, double* Cm, int tid) {
    constexpr int D = GramFin<S>::D, NBATCH = AGENT ? 16 : 4;
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    int src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, dv = e >> 5, du = e & 31;
        const int sv = dv < 16 ? dv : 16 + state_pos<S>(dv - 16 < 2 * S ? dv - 16 : 0);
        src[q] = (dv < D && du < D) ? (((sv >> 4) * 2 + (du >> 4)) * 4 + (sv & 3)) * 64 + ((sv & 15) >> 2) * 16 + (du & 15) : -1;
    }
    for (int c0 = 0; c0 < nmat; c0 += NBATCH) {
        double v[NBATCH][4];
#pragma unroll
        for (int j = 0; j < NBATCH; ++j) {
            const double* g = g0 + (size_t)(c0 + j < nmat ? c0 + j : c0) * 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[j][q] = src[q] >= 0 ? fin_ld<AGENT>(g + src[q]) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NBATCH; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) cs[q] += c0 + j < nmat ? v[j][q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        Cm[(e >> 5) * 33 + (e & 31)] = cs[q];
    }
}
// A workgroup's own matrix (segmented pass with the finalize step inside the launch): the four sums of thread tid - entries tid + 256 i of
// the kernel's register layout (register r = entry / 64 = 4 (2 bv + bu) + e of lane entry % 64 is C[16 bv + 4 (lane / 16) + e][16 bu +
// lane % 16]) - scattered into the zero-filled C of the work area; 256 threads.
template <int S>
__device__ __forceinline__ void gram_scatter(const double (&s4)[4], double* Cm, int tid) {
    constexpr int D = GramFin<S>::D;
    for (int e = tid; e < 32 * 33; e += 256) Cm[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, reg = idx >> 6, lane = idx & 63;
        const int sv = 16 * (reg >> 3) + 4 * (lane >> 4) + (reg & 3), du = 16 * ((reg >> 2) & 1) + (lane & 15);
        int dv = sv;
        if (sv >= 16) {
            const int comp = state_comp_at<S>(sv - 16);
            dv = comp >= 0 ? 16 + comp : -1;
        }
        if (dv >= 0 && dv < D && du < D) Cm[dv * 33 + du] = s4[i];
    }
}
// step 2 (products, lag sums). The operands that come from the basis responses are a thread's own registers, fetched by gram_prefetch -
// in the segmented pass before the workgroup's matrix is even reduced, so that their trip to L2 / memory (half of this step's time when
// they were loaded where they are used: profiles/r05/seg_tail_trace.log) runs under that reduction.
template <int S>
struct GramOps {
    static constexpr int NST = (GramFin<S>::D + 3) / 4, NB = (2 * GramFin<S>::NPB + 3) / 4;
    double bop[NST][NB];            // FW entries: the B operands of this lane's matrix-core products
    double f[GramFin<S>::D];        // F[k][n][.] of this thread's (which, k, n)
};
template <int S, bool AGENT>
__device__ __forceinline__ void gram_prefetch(const double* bas, int tid, GramOps<S>& r) {
    using GF = GramFin<S>;
    constexpr int L = GF::L, D = GF::D, NP = GF::NP;
    const int wave = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
#pragma unroll
    for (int st = 0; st < GramOps<S>::NST; ++st) {
        const int u = 4 * st + lk, uu = u < D ? u : D - 1;           // (C is zero beyond D)
#pragma unroll
        for (int q = 0; q < GramOps<S>::NB; ++q) {
            const int p0 = 16 * ((wave + 4 * q) >> 1), pmr = p0 + li < NP ? p0 + li : NP - 1;   // (pad columns of the last block and blocks past the end: any row - their results are not stored)
            r.bop[st][q] = fin_ld<AGENT>(bas + GF::fw(pmr, uu));
        }
    }
    const int which = tid / (16 * S), k = (tid / 16) % S, n = tid & 15;      // (threads beyond 2 x 16 S read a valid row they do not use)
#pragma unroll
    for (int v = 0; v < D; ++v) r.f[v] = fin_ld<AGENT>(bas + ((which & 1) ? GF::fo(k, n, v) : GF::fg(k, n, v)));
}
// wk = the LDS work area with C in place (barrier before the call) -> the 5 S lag sums in wk + LAG (barrier after the call before they are
// read); 256 threads. `item` only serves the developer trace.
template <int S>
__device__ __forceinline__ void gram_lag_sums(const GramOps<S>& r, double* wk, int item, int tid) {
    using GF = GramFin<S>;
    constexpr int D = GF::D, NW = GF::NW, NP = GF::NP, NPB = GF::NPB;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const double* Cm = wk + GF::CM;
    double* P = wk + GF::P;
    double* lagsum = wk + GF::LAG;
    {   // P^T (32 x NP) = C (32 x 32) FW^T (32 x NP): 2 x NPB blocks of 16 x 16, (D + 3) / 4 steps of 4 each (C is zero beyond D), dealt out over the four waves
        const int wave = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
        constexpr int NB = GramOps<S>::NB;               // blocks per wave; their accumulator chains run side by side
        d4 acc[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < GramOps<S>::NST; ++st) {
            const int u = 4 * st + lk;
#pragma unroll
            for (int q = 0; q < NB; ++q)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[(16 * ((wave + 4 * q) & 1) + li) * 33 + u], r.bop[st][q], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int blk = wave + 4 * q, v0 = 16 * (blk & 1), p0 = 16 * (blk >> 1);
            if (blk < 2 * NPB && p0 + li < NP) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) P[(p0 + li) * 33 + v0 + lk + 4 * rr] = acc[q][rr];
            }
        }
    }
    __syncthreads();
    GTRACE(54);
    {
        const int which = tid / (16 * S), k = (tid / 16) % S, n = tid & 15;
        const bool on = tid < 2 * 16 * S;
        double v3[3] = {0.0, 0.0, 0.0};
        if (on) {
#pragma unroll
            for (int jl = 0; jl < 3; ++jl) {
                const double* pr = P + (k * NW + n + 2 - jl) * 33;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int v = 0; v < D; v += 2) { a0 = fma(r.f[v], pr[v], a0); a1 = fma(r.f[v + 1], pr[v + 1], a1); }
                v3[jl] = a0 + a1;
            }
        }
#pragma unroll
        for (int jl = 0; jl < 3; ++jl) {        // sum over the 16 samples n = the 16 lanes of a DPP row
            double a = v3[jl];
            a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
            v3[jl] = a;
        }
        if (on && n == 0) {
            if (!which) { lagsum[k * 5 + 0] = v3[0]; lagsum[k * 5 + 1] = v3[1]; lagsum[k * 5 + 2] = v3[2]; }
            else { lagsum[k * 5 + 3] = v3[1]; lagsum[k * 5 + 4] = v3[2]; }
        }
    }
    GTRACE(55);
}
// step 3: thread k < S turns section k's lag sums (wk + LAG, visible) into its gradients. ec: the section's design numbers (EmitCoef,
// loaded by the caller - in the segmented pass long before the hand-off that precedes this step)
template <int S>
__device__ __forceinline__ void gram_emit(const EmitCoef& ec, const double* wk, int B, int mode, float* __restrict__ gout, int item, int tid) {
    const double* lagsum = wk + GramFin<S>::LAG;
    if (tid < S) {
        const double g5[5] = {lagsum[tid * 5 + 0], lagsum[tid * 5 + 1], lagsum[tid * 5 + 2], -lagsum[tid * 5 + 3], -lagsum[tid * 5 + 4]};
        emit_section_grads(ec, g5, B, S, mode, gout, item, tid);
    }
    GTRACE(56);
}

template <int S>
__global__ void __launch_bounds__(256)
sos_gram_finalize_kernel(const double* __restrict__ dtab, int tab_bcast, const double* __restrict__ gram, int B, int C, int mode,
                         float* __restrict__ gout) {
    using GF = GramFin<S>;
    __shared__ double wk[GF::WORK];
    __shared__ double bas[GF::BASIS];
    const int tid = threadIdx.x, item = blockIdx.x;
    const double* d0 = dtab + (size_t)(tab_bcast ? 0 : item) * S * DT_STRIDE;
    GTRACE(50);
    // every global read of the kernel up front and in flight together (one trip to L2 instead of one per section / per matrix entry)
    gram_fin_coefs<S>(d0, wk + GF::CF, tid);
    gram_sum<S, false>(gram + (size_t)item * C * 1024, C, wk + GF::CM, tid);
    __syncthreads();
    GTRACE(51);
    gram_basis_responses<S, false>(wk + GF::CF, bas, tid);
    GTRACE(52);
    __syncthreads();
    GTRACE(53);
    GramOps<S> ops;
    gram_prefetch<S, false>(bas, tid, ops);
    gram_lag_sums<S>(ops, wk, item, tid);
    __syncthreads();
    gram_emit<S>(load_emit_coef(d0 + (tid < S ? tid : 0) * DT_STRIDE), wk, B, mode, gout, item, tid);
}

// The tail of the segmented Gram pass (sos_bwd_gram_kernel<SEG = 1> with fz.on). The finalize step is LINEAR in the Gram matrix, so every
// (row, segment) workgroup applies it to its own matrix - straight from its registers, the matrix never goes to memory - with the basis
// responses the design kernel left in fz.basis: 5 S lag sums per workgroup (lagbuf: 32 doubles per workgroup), all workgroups side by
// side. What is left for the workgroup that completes the item's count (common.hpp hand-off) is the sum of those few numbers over the
// item's workgroups and the design Jacobian. (First version: the last workgroup summed the 8 KiB matrices and ran the whole step alone -
// measured 13 - 21 us of tail at (8 .. 16, 2, 131072): profiles/r05/seg_tail_trace.log.) wk: GramFin<S>::WORK doubles of LDS, idle by now.
struct GramFuse {
    int on, B, mode;
    const double* dtab;
    float* gout;
    const double* basis;    // [item][GramFin<S>::BASIS], written by sos_prep_kernel
    float* cnt_tab;         // the items' tables: word LY::CNT of an item's table counts its arrivals (zeroed by the prep kernel, reset here)
    const double* segtab_adj;       // look-back launches (SEG 3): the adjoint system's segment matrix of item 0 (stride 2 (2S)^2 doubles per item)
    unsigned long long* lb_words;   // ... and their words [row][segment][2S]: the workgroup that finalizes an item invalidates the item's
    unsigned* err;                  // ... and the device error words a reader reports a word to that never arrived (common.hpp lookback_poll)
};
// red: the four waves' 1024 sums each, [wave][red_stride] doubles in LDS (visible); it may overlap wk - it is read into registers first.
template <int S>
__device__ __forceinline__ void gram_fused_tail(const GramFuse& fz, int item, int nwg, int wg_in_item, double* lagbuf_item, const double* red, int red_stride, double* wk) {
    using LY = SosLayout<S, 16>;
    using GF = GramFin<S>;
    __shared__ int s_last;
    const int tid = threadIdx.x;
#ifdef DASP_TRACE
    long long tail_t[5];
#define TAIL_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tail_t[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TAIL_STAMP(i)
#endif
    TAIL_STAMP(0);
    // everything this workgroup will want from memory, requested before anything else: the operands of the lag sums, the design numbers
    // of the last step (only the workgroup that ends up finalizing uses those)
    GramOps<S> ops;
    gram_prefetch<S, false>(fz.basis + (size_t)item * GF::BASIS, tid, ops);
    const EmitCoef ec = load_emit_coef(fz.dtab + ((size_t)item * S + (tid < S ? tid : 0)) * DT_STRIDE);
    double s4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) a += red[w * red_stride + tid + 256 * i];
        s4[i] = a;
    }
    __syncthreads();        // the sums are in registers: the work area may overwrite them
    gram_scatter<S>(s4, wk + GF::CM, tid);
    __syncthreads();
    TAIL_STAMP(1);
    gram_lag_sums<S>(ops, wk, item, tid);
    __syncthreads();
    if (tid < 5 * S) fin_st<true>(lagbuf_item + (size_t)wg_in_item * 32 + tid, wk[GF::LAG + tid]);
    TAIL_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = handoff_arrive_is_last(reinterpret_cast<int*>(fz.cnt_tab + (size_t)item * LY::TOTAL + LY::CNT), nwg);
    __syncthreads();
    if (!s_last) return;
    TAIL_STAMP(3);
    if (tid < 5 * S) {
        double acc = 0.0;
        for (int m0 = 0; m0 < nwg; m0 += 32) {         // thirty-two loads in flight (the usual count: one trip to memory); summed in index order
            double v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = fin_ld<true>(lagbuf_item + (size_t)(m0 + j < nwg ? m0 + j : m0) * 32 + tid);
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += m0 + j < nwg ? v[j] : 0.0;
        }
        wk[GF::LAG + tid] = acc;
    }
    __syncthreads();
    gram_emit<S>(ec, wk, fz.B, fz.mode, fz.gout, item, tid);
    if (fz.lb_words) {      // every workgroup of the item has taken what it needed: a second backward pass over the same tables (same tag) must not
        for (int e = tid; e < nwg * 2 * S; e += 256)      // find this one's adjoint states
            __hip_atomic_store(fz.lb_words + (size_t)item * nwg * 2 * S + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    TAIL_STAMP(4);
#ifdef DASP_TRACE
    if (item == 0 && tid == 0)
        for (int i = 0; i < 5; ++i) g_trace[57 + i] = tail_t[i];       // the stamps of ONE workgroup - the one that finalized item 0
#endif
}

