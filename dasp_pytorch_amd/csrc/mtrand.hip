// torch's CPU random stream, generated on the device: the 32-bit Mersenne Twister of torch's default CPU generator, run from the state
// the host hands over, laid out as `torch.randn` lays a float32 tensor out on the CPU. Replaces the draw at
// dasp_pytorch/functional.py:548 - `torch.randn(bs*2, 12, num_samples + num_bandpass_taps - 1)` on the global CPU generator whatever the
// device of x - which costs the reference (and cost this library's drop-in default) one host thread 0.56 s and a 0.8 GB copy per call
// at (128,2,262144), while `torch.manual_seed` parity with the reference needs exactly that stream.
//
// What is reproduced (ATen/core/MT19937RNGEngine.h, ATen/core/TransformationHelper.h uniform_real, ATen/native/cpu/
// DistributionTemplates.h normal_fill; restated in numpy and pinned to torch.randn in oracle/mt_stream.py):
//   words   x[k + 624] = x[k + 397] ^ twist(x[k], x[k + 1]), handed out tempered, 624 at a time;
//   floats  u = (y & 0xFFFFFF) * 2^-24, one word per element, element i of the tensor <- draw i;
//   normals every aligned group of 16 elements is 8 Box-Muller pairs (j, j + 8): r = sqrt(-2 log(1 - u[j])), a = 2 pi u[j + 8],
//           element j <- r cos a, element j + 8 <- r sin a; numel % 16 != 0: the LAST 16 elements are recomputed from 16 more draws.
//
// How it is made parallel. The twister is linear over GF(2): x[n + J] = XOR over the set coefficients i of g_J(t) = t^J mod p(t) of
// x[n + i] (p: the characteristic polynomial, degree 19937; host side _mt19937.py). A UNIT is 256 regenerations (159,744 words); a
// CHUNK is ceil(units / 256) units - at most 256 chunks, one workgroup of the generation kernel each, no CU a second one. The start
// state of the chunk 256 a + b units on is summed by mt_jump_kernel out of a 20,560-word window of the sequence held in LDS: from
// chunk 0's state for the 7 "giant" targets 256 a, from those by the 255 "baby" polynomials t^(b J), all workgroups of a phase side
// by side. A jump: every wave holds the whole new state (ten words per lane), reads a 26-word span per group of 16 exponents once
// and XORs registers behind scalar branches on the group's coefficient bits: ~82 us of a CU, fewer jumps than CUs split over up to
// eight workgroups. Then every chunk regenerates its blocks 224 words per step (any 227 consecutive new words are independent) on ONE
// wave whose step-to-step chain stays in registers, while seven waves turn the words into normals. HBM: 4 B written per value.
#include <atomic>
#include "common.hpp"

namespace dasp {
namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int DASP_DEVERR_MT = 5;                      // this file's word among the device error words (common.hpp: 0 .. 4 are the look-back families)
constexpr int MT_BLOCKS_PER_CHUNK = 256;               // regenerations per unit; a chunk is `stride` units (mt_plan)
constexpr int MT_N_BABY = 255, MT_N_GIANT = 7;
constexpr int MT_GROUP = 16;                           // exponents per group of a jump polynomial: one uint16 of coefficient bits
constexpr int MT_N_GROUPS = (19937 + MT_GROUP - 1) / MT_GROUP;     // 1,247
constexpr int MT_ROW = 1280;                           // uint16 per polynomial in the table (the groups, padded with zeros)
constexpr int MT_PAD_INDEX = 20560;                    // = 19937 + 623: the sequence window of one jump
constexpr int MT_SEQ_LDS = MT_PAD_INDEX + 648;         // + zeros behind it (the last lanes' spans reach 20,592)
constexpr int MT_SPAN_L = 10;                          // state words per lane of a jumping wave: 63 lanes hold the 624
constexpr int MT_JUMP_THREADS = 1024;                  // 16 waves, each a whole copy of the state for its share of the groups
#ifndef DASP_MT_GEN_WAVES
#define DASP_MT_GEN_WAVES 8
#endif
constexpr int MT_GEN_THREADS = 64 * DASP_MT_GEN_WAVES;   // one regenerating wave + seven Box-Muller waves (12 / 16 waves: no faster)
constexpr int MT_STEP = 224;                           // new words per regeneration step: a multiple of 16 not above 227
constexpr int MT_RING = 16384;                         // raw-word ring of a generating workgroup: 73 steps of 224 words; 64 KiB: byte offsets wrap as 16-bit sums
#ifndef DASP_MT_CHUNKS_TARGET
#define DASP_MT_CHUNKS_TARGET 256
#endif
constexpr int MT_CHUNKS_TARGET = DASP_MT_CHUNKS_TARGET;   // chunks of a large draw: one per CU (512, two per CU, measured slower: profiles/r06/README.md)
#ifndef DASP_MT_PROBE
#define DASP_MT_PROBE 0                              // timing probes of the generation kernel (scripts/mtprobe_build.sh): 1 no Box-Muller work, 2 no stores, 3 no priority, 4 no Box-Muller wave on the regenerating wave's SIMD
#endif
#ifndef DASP_MT_SKIP
#define DASP_MT_SKIP 0                               // timing probes of the regenerating wave: leave out 1 the write, 2 lane 0's read, 4 the x[k-624] read, 8 the twists, 16 the chain
#endif
#ifndef DASP_MT_UNIT
#define DASP_MT_UNIT 4
#endif
constexpr int MT_UNIT = DASP_MT_UNIT;                  // steps a Box-Muller wave claims at a time: 4 steps = 448 pairs = seven passes of 64 lanes

struct MtState { unsigned w[MT_N]; };                  // 2,496 bytes: travels as a kernel argument, no host-to-device copy

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7FFFFFFFu)) >> 1) ^ ((v & 1u) ? 0x9908B0DFu : 0u);
}
// a ^ (b & c) in one instruction (the compiler takes it for register masks only; the tempering masks are literals)
__device__ __forceinline__ unsigned mt_xor_and(unsigned a, unsigned b, unsigned mask) {
    unsigned r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x78" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y = mt_xor_and(y, y << 7, 0x9D2C5680u);
    y = mt_xor_and(y, y << 15, 0xEFC60000u);
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ float mt_uniform(unsigned raw) { return (float)(mt_temper(raw) & 0xFFFFFFu) * 0x1p-24f; }

// chunk 0's state from the kernel argument; the other states (chunks, then the giant jumps' targets) start as zeros (a jump shared by
// several workgroups is summed into its state with atomic XORs)
__global__ void __launch_bounds__(640) mt_seed_kernel(MtState s, unsigned* __restrict__ states, int n_states) {
    if (blockIdx.x == 0) { if (threadIdx.x < MT_N) states[threadIdx.x] = s.w[threadIdx.x]; return; }
    for (size_t k = MT_N + (size_t)(blockIdx.x - 1) * 640 + threadIdx.x; k < (size_t)n_states * MT_N; k += (size_t)(gridDim.x - 1) * 640) states[k] = 0u;
}

// ---- the regeneration chain of one wave (mt_generate_kernel, whose comments explain it, and the window of mt_jump_kernel) ----
typedef unsigned mt_u32x4 __attribute__((ext_vector_type(4)));
// LDS accesses of the regenerating wave, written out: their order on the LDS queue and the waits are the algorithm there. (The compiler
// does not know that a register is in flight between a read and its wait: every use sits behind the wait through the "+v" operands.)
__device__ __forceinline__ mt_u32x4 mt_lds_read128(unsigned addr) { mt_u32x4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory"); return r; }
__device__ __forceinline__ void mt_lds_write128(unsigned addr, mt_u32x4 w) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(w) : "memory"); }
__device__ __forceinline__ void mt_lds_write32(unsigned addr, unsigned w) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(w) : "memory"); }
template <int OUTSTANDING> __device__ __forceinline__ void mt_lds_wait(mt_u32x4& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(OUTSTANDING) : "memory");
}
// byte offsets into the 64 KiB ring wrap as 16-bit sums: one instruction (a 16-bit add clears the upper half of its result on this chip)
template <int C> __device__ __forceinline__ unsigned mt_add16(unsigned a) { return (a + (unsigned)C) & 0xFFFFu; }
// One statement for a step's LDS traffic (between separate asm statements the compiler puts a wait state): write w at a, advance a
// by ADV, read 16 bytes at the new a + C into r. r is a read-write operand: the register of a value in flight stays the variable's
// register from step to step (a copy of one before its wait would copy what was there before).
template <int ADV, int C, bool WRAP16>
__device__ __forceinline__ void mt_ring_write_advance_read(unsigned& a, mt_u32x4 w, mt_u32x4& r) {
    unsigned t;
    if (WRAP16)
        asm volatile(
#if !(DASP_MT_SKIP & 1)
                     "ds_write_b128 %2, %3\n\t"
#endif
                     "v_add_u16 %2, %4, %2\n\tv_add_u16 %1, %5, %2"
#if !(DASP_MT_SKIP & 4)
                     "\n\tds_read_b128 %0, %1"
#endif
                     : "+v"(r), "=&v"(t), "+v"(a) : "v"(w), "n"(ADV & 0xFFFF), "n"(C & 0xFFFF) : "memory");
    else
        asm volatile("ds_write_b128 %2, %3\n\tv_add_u32 %2, %4, %2\n\tv_subrev_u32 %1, %5, %2\n\tds_read_b128 %0, %1"
                     : "+v"(r), "=&v"(t), "+v"(a) : "v"(w), "n"(ADV), "n"(-C) : "memory");          // (C < 0)
}
// (lane l - 1's `from_below`) ^ b, lane 0 from lane 63 (a rotation of the whole wave)
__device__ __forceinline__ unsigned mt_xor_ror1(unsigned from_below, unsigned b) {
    unsigned r;
    asm("v_xor_b32_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(from_below), "v"(b));
    return r;
}
// lanes 60 .. 63 of `now` <- lanes 52 .. 55 of `before` (the other lanes stay)
__device__ __forceinline__ unsigned mt_keep_tail(unsigned now, unsigned before) {
    asm("v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0x8 bank_mask:0x8" : "+v"(now) : "v"(before));
    return now;
}
// twist(x[k], x[k + 1]) for the lane's four words; x[k + 4] is the next lane's first. 17 instructions: every word is shifted once
// (it is the upper word of one twist and the lower word of the next), one bit-field insert per twist, the matrix row by a sign
// extension of bit 0 and one three-operand bit operation.
__device__ __forceinline__ unsigned mt_bfi(unsigned mask, unsigned from_set, unsigned from_clear) {
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(from_set), "v"(from_clear));
    return r;
}
__device__ __forceinline__ mt_u32x4 mt_twist4(mt_u32x4 a) {
    const unsigned a4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a.x, 0x130, 0xf, 0xf, true);   // wave_shl:1: lane l reads lane l + 1
    const unsigned s0 = a.x >> 1, s1 = a.y >> 1, s2 = a.z >> 1, s3 = a.w >> 1, s4 = a4 >> 1;
    mt_u32x4 t;                                        // ((u & 0x80000000 | v & 0x7fffffff) >> 1) ^ (v & 1 ? 0x9908b0df : 0)
    t.x = mt_bfi(0x3FFFFFFFu, s1, s0) ^ ((unsigned)__builtin_amdgcn_sbfe((int)a.y, 0, 1) & 0x9908B0DFu);
    t.y = mt_bfi(0x3FFFFFFFu, s2, s1) ^ ((unsigned)__builtin_amdgcn_sbfe((int)a.z, 0, 1) & 0x9908B0DFu);
    t.z = mt_bfi(0x3FFFFFFFu, s3, s2) ^ ((unsigned)__builtin_amdgcn_sbfe((int)a.w, 0, 1) & 0x9908B0DFu);
    t.w = mt_bfi(0x3FFFFFFFu, s4, s3) ^ ((unsigned)__builtin_amdgcn_sbfe((int)a4, 0, 1) & 0x9908B0DFu);
    return t;
}

// One regeneration step of the wave that owns the chain (mt_generate_kernel's comments): w(v) = tw(v) ^ rot(z(v - 1)) written at q_b,
// step v + 2's x[k - 624] side asked for into a_refill, tw(v + 1) from a_ready (asked for a step ago). WRAP16: q_b wraps at 64 KiB.
template <bool WRAP16>
__device__ __forceinline__ void mt_regen_step(unsigned& q_b, mt_u32x4& z, mt_u32x4& tw, mt_u32x4& a_refill, mt_u32x4& a_ready) {
    mt_u32x4 w;
#if DASP_MT_SKIP & 16
    w = tw;
#else
    w.x = mt_xor_ror1(z.y, tw.x);
    w.y = mt_keep_tail(mt_xor_ror1(z.z, tw.y), z.y);
    w.z = mt_keep_tail(mt_xor_ror1(z.w, tw.z), z.z);
    w.w = mt_keep_tail(tw.w ^ z.x, z.w);
#endif
    // step v + 1's x[k - 624] side: asked for a step ago. (The wait in FRONT of this step's write: the counter is in order, behind the
    // write it would wait for the write as well - 64 cycles per step, found with probe builds.)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_ready) :: "memory");
    // the write (lanes 56 .. 63 write into the next step's slots, which that step overwrites); behind it step v + 2's x[k - 624] side
    mt_ring_write_advance_read<4 * MT_STEP, 4 * MT_STEP - 4 * MT_N, WRAP16>(q_b, w, a_refill);
#if DASP_MT_SKIP & 8
    tw = a_ready;
#else
    tw = mt_twist4(a_ready);
#endif
    z = w;
}
// The start of the chain: q_b = byte address of the lane's first word of step 1 (the 624 words in front of it are the start state)
template <bool WRAP16>
__device__ __forceinline__ void mt_regen_start(int lane, unsigned q_b, mt_u32x4& z, mt_u32x4& tw, mt_u32x4& a0, mt_u32x4& a1) {
    const unsigned m = WRAP16 ? 0xFFFFu : 0xFFFFFFFFu;
    mt_u32x4 first = mt_lds_read128((q_b - 4u * MT_N) & m);                   // step 1: x[4 l .. 4 l + 3]
    // "step 0": the last 224 words of the start state; lane 63: the four words below them (lane 0's, as lane 55 of "step -1" would hold them)
    z = mt_lds_read128(lane == 63 ? (q_b - 4u * 4 * 63 - 4u * MT_STEP - 16u) & m : (q_b - 4u * MT_STEP) & m);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(first), "+v"(z) :: "memory");
    tw = mt_twist4(first);
    a0 = mt_lds_read128((q_b + 4u * MT_STEP - 4u * MT_N) & m);                // step 2's (inside the start state)
    a1 = a0;                                                                   // (defined; refilled in step 1)
}

// One jump: dst = g(T) src. Chunk ch starts stride * ch = 256 a + b units (of 256 regenerations, J words) behind chunk 0.
// giant: src = chunk 0, polynomial N_BABY + blockIdx (t^(256 (blockIdx + 1) J)), dst = giants[blockIdx] (behind the chunks' states);
// baby: chunk blockIdx + 1: src = giants[a - 1] (a = 0: chunk 0), polynomial b - 1 (t^(b J)); b = 0: a copy. Word 0 of a jumped state is
// right in its top bit only - the one bit of it the recurrence reads. `parts` workgroups share a jump when there are fewer jumps than
// CUs (each takes a share of the groups of exponents; blockIdx = jump * parts + part).
__global__ void __launch_bounds__(MT_JUMP_THREADS)
mt_jump_kernel(unsigned* __restrict__ states, const unsigned short* __restrict__ table, int giant, int n_chunks, int parts, int stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned seq[];
    const int tid = threadIdx.x, job = blockIdx.x / parts, part = blockIdx.x % parts;
    unsigned* giants = states + (size_t)n_chunks * MT_N;
    const unsigned* src;
    unsigned* dst;
    int poly;
    if (giant) { src = states; poly = MT_N_BABY + job; dst = giants + (size_t)job * MT_N; }
    else {
        const int m = stride * (job + 1), a = m >> 8, b = m & 255;
        src = a ? giants + (size_t)(a - 1) * MT_N : states;
        dst = states + (size_t)(job + 1) * MT_N;
        poly = b - 1;
        if (b == 0) {                                                          // the chunk starts where a giant jump landed
            if (part == 0)
                for (int k = tid; k < MT_N; k += MT_JUMP_THREADS) { if (parts == 1) dst[k] = src[k]; else atomicXor(dst + k, src[k]); }
            return;
        }
    }

    for (int k = tid; k < MT_N; k += MT_JUMP_THREADS) seq[k] = src[k];
    for (int k = MT_PAD_INDEX + tid; k < MT_SEQ_LDS; k += MT_JUMP_THREADS) seq[k] = 0u;
    __syncthreads();
    // the window: 19,936 more words = 89 steps of 224 by one wave, its chain in registers (the generation kernel's; 227 lanes with a
    // barrier per step took 14.5 us); 90 steps, the overhang and the last lanes' garbage zeroed again
    if (tid < 64) {
        if ((unsigned)(unsigned long long)seq != 0u) __builtin_trap();         // (the window is the kernel's only LDS object: address 0)
        unsigned q_b = 4u * (MT_N + 4 * tid);
        mt_u32x4 A[2], z, tw;
        mt_regen_start<false>(tid, q_b, z, tw, A[0], A[1]);
        for (int v = 0; v < 45; ++v) {
            mt_regen_step<false>(q_b, z, tw, A[1], A[0]);
            mt_regen_step<false>(q_b, z, tw, A[0], A[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]) :: "memory");
        for (int k = MT_PAD_INDEX + tid; k < MT_SEQ_LDS; k += 64) seq[k] = 0u;
    }
    __syncthreads();

    // new[k] = XOR over the set coefficients i of x[i + k], k = 0 .. 623. Every wave holds the whole sum for its share of the groups of
    // 16 exponents: lane l the ten words k = 10 l .. 10 l + 9. For a group 16 g .. 16 g + 15 the lane reads the 26 words from 16 g + 10 l
    // on ONCE (13 aligned 8-byte reads; lane stride 40 bytes: no bank conflict inside a half wave) and every set exponent 16 g + s is
    // ten XORs of registers s .. s + 9 - behind a scalar branch on the group's coefficient bits (the polynomial is a bit mask in the
    // table: 2.5 KB). Per exponent that is ~6.5 bytes of LDS per lane where a read per exponent and two-word lane was 8 for two words,
    // and no address arithmetic: the jump was 127 us of LDS reads and address instructions, see profiles/r06/README.md.
    // The groups' bits come by one vector load per lane (lane j: the wave's j-th and (64 + j)-th group) and go round by v_readlane.
    const int lane = tid & 63, n_waves = (MT_JUMP_THREADS / 64) * parts, wave = part * (MT_JUMP_THREADS / 64) + (tid >> 6);
    const unsigned short* row = table + (size_t)poly * MT_ROW;
    const int g_lo = wave + lane * n_waves, g_hi = wave + (64 + lane) * n_waves;
    const unsigned bits_lo = g_lo < MT_N_GROUPS ? row[g_lo] : 0u, bits_hi = g_hi < MT_N_GROUPS ? row[g_hi] : 0u;
    unsigned acc[MT_SPAN_L];
#pragma unroll
    for (int j = 0; j < MT_SPAN_L; ++j) acc[j] = 0u;
    const unsigned long long* lane_base = reinterpret_cast<const unsigned long long*>(seq) + (MT_SPAN_L / 2) * lane;
    for (int k = 0, g = wave; g < MT_N_GROUPS; ++k, g += n_waves) {
        const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)(k < 64 ? bits_lo : bits_hi), k & 63);
        if (bits == 0u) continue;
        const unsigned long long* p = lane_base + (MT_GROUP / 2) * g;
        unsigned span[MT_GROUP + MT_SPAN_L];
#pragma unroll
        for (int r = 0; r < (MT_GROUP + MT_SPAN_L) / 2; ++r) { const unsigned long long v = p[r]; span[2 * r] = (unsigned)v; span[2 * r + 1] = (unsigned)(v >> 32); }
#pragma unroll
        for (int e = 0; e < MT_GROUP; ++e)
            if (bits & (1u << e)) {
#pragma unroll
                for (int j = 0; j < MT_SPAN_L; ++j) acc[j] ^= span[e + j];
            }
    }
    // fold the waves' sums through the LDS (the window is done with)
    __syncthreads();
    if (lane < 63) {
#pragma unroll
        for (int j = 0; j < MT_SPAN_L; ++j) seq[(tid >> 6) * 640 + MT_SPAN_L * lane + j] = acc[j];
    }
    __syncthreads();
    for (int k = tid; k < MT_N; k += MT_JUMP_THREADS) {
        unsigned v = 0u;
#pragma unroll
        for (int w = 0; w < MT_JUMP_THREADS / 64; ++w) v ^= seq[w * 640 + k];
        if (parts == 1) dst[k] = v; else atomicXor(dst + k, v);
    }
}

// sqrt(-2 ln(1 - u)) for u = k 2^-24, k < 2^24: the argument of the logarithm lies in [2^-24, 1] - no denormals, infinities or NaNs - so
// the library routines' range handling (scaling by 2^32, class tests, the square root's last-bit correction) is left out: v_log_f32
// (log2, 1 ulp) times ln 2 in two pieces as the library does it, v_sqrt_f32 (1 ulp). 13 vector instructions less per pair of values.
__device__ __forceinline__ float mt_radius(float u) {
    const float l2 = __builtin_amdgcn_logf(1.f - u);
    const float hi = l2 * 0x1.62e42ep-1f;
    const float ln = fmaf(l2, 0x1.62e42ep-1f, -hi) + fmaf(l2, 0x1.efa39ep-25f, hi);        // ln 2 = 0x1.62e42e p-1 + 0x1.efa39e p-25
    return __builtin_amdgcn_sqrtf(-2.f * ln);
}

// cosine and sine of a in [0, 2 pi): quadrant by Cody-Waite, Cephes' single-precision kernels on [-pi/4, pi/4] (the arithmetic torch's
// vectorised normal_fill runs; the scalar one calls libm - both within an ulp or two of this)
__device__ __forceinline__ void mt_sincos(float a, float& s, float& c) {
    const float k = rintf(a * 0.636619772367581343f);
    float r = fmaf(-k, 1.5703125f, a);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.549789948768648e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.f));
    const unsigned q = (unsigned)(int)k;
    const float ss = (q & 1u) ? cp : sp, cc = (q & 1u) ? sp : cp;
    s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ss) ^ ((q << 30) & 0x80000000u));          // quadrants 2, 3: -sin
    c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cc) ^ (((q + 1u) << 30) & 0x80000000u));   // quadrants 1, 2: -cos
}

// One chunk: blocks 256 c + 1 .. of the sequence from states[c] (block 256 c), every aligned group of 16 draws whose last word lies
// in those blocks (chunk 0: also the groups inside the state it starts from), the 16 tail draws it owns, and - the last chunk - the
// generator state afterwards. word q of the chunk (q = 0 .. 623: the start state) is draw 624 (256 c) + q - (624 - rem).
//
// A pipeline inside the workgroup, no barrier in the loop. Wave 0 regenerates: steps of 224 words (any 227 consecutive new words are
// independent), four consecutive words per lane, into a ring of 16,384 words in the LDS. The serial chain of the chunk never touches
// the LDS: a step's x[k - 227] words are the previous step's output one lane over (a rotation of the wave), its x[k - 624] side was
// read from the ring a step ahead. What is left is ONE wave's instruction issue: 30 instructions per step at ~6.5 cycles each (a lone
// wave issues an independent vector instruction every 5.5 cycles, a dependent one every 8.6, a 16-byte LDS write costs it 26:
// tools/ubench6.hip, ubench7.hip). After every unit of MT_UNIT steps the wave publishes how far the sequence reaches (`avail`). Waves
// 1-7 turn units into normals: each claims the next unit from a counter (the wave that shares its SIMD with the regenerating one
// simply claims fewer), waits for `avail`, reads the unit's words (two per pair: 448 pairs = seven full passes of the wave),
// Box-Muller, stores, and publishes the unit it is at (`cur`); the regenerating wave stays at most MT_LEAD steps ahead of the slowest
// of them. It also runs up to MT_UNIT - 1 steps past the chunk's end: valid words nobody reads.
// History (profiles/r06/README.md), per chunk of 256 regenerations: every wave doing everything between barriers 235 us; four
// regenerating and four Box-Muller waves with two barriers per 448 words 165 us (the regeneration with its barriers alone: 125);
// this pipeline with the chain through the LDS 136 us dealt out in turn, 88 us with claimed units (the regenerating wave alone: 82, an
// LDS round trip per step); the chain in registers 85; x[k - 624] side asked for a step ahead, 16-bit address sums 78; lane 0's words
// by the rotation instead of an LDS read and - the larger part - the wait moved in front of the step's write (the LDS counter is in
// order: behind the write it waited for the write, 64 cycles per step) 71 (the regenerating wave alone 63, without the stores 65).
// Every wait is bounded: a wave that polls 2^22 times gives up, the output starts with a NaN and the device error word of this family is
// set - the next call returns DASP_ERR_DEVICE (it cannot happen: the eight waves of a workgroup are resident together).
constexpr int MT_CONSUMERS = MT_GEN_THREADS / 64 - 1;
constexpr int MT_GEN_LDS = 4 * (MT_RING + 4 + MT_CONSUMERS + 1);          // bytes of LDS of a generating workgroup
constexpr int MT_LEAD = (MT_RING - 47) / MT_STEP;                          // steps the regenerating wave may be ahead of the slowest reader
// step v writes words 624 + 224 (v - 1) .. + 255 (all 64 lanes store): slots of words MT_RING below those; the readers of steps > v - MT_LEAD
// read from 624 + 224 (v - MT_LEAD) - 15 on
static_assert(MT_RING >= MT_LEAD * MT_STEP + 32 + 15 && MT_RING >= MT_N + MT_STEP + 32 && (MT_RING & (MT_RING - 1)) == 0, "ring size");
static_assert(MT_UNIT * MT_CONSUMERS + MT_UNIT <= MT_LEAD, "every Box-Muller wave at a unit of its own, the regenerating one a unit ahead");

__device__ __forceinline__ int mt_step_end(int v, int q_end) { const int e = MT_N + MT_STEP * v; return e < q_end ? e : q_end; }   // words [0, e) of the chunk exist after step v

__global__ void __launch_bounds__(MT_GEN_THREADS)
mt_generate_kernel(const unsigned* __restrict__ states, float* __restrict__ out, long long n, int rem, long long beta_max,
                   unsigned* __restrict__ final_state, float* __restrict__ tail_u, int bpc, unsigned* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned ring[];       // MT_RING words (64 KiB: asked for at the launch) and the words below
    int& avail = *reinterpret_cast<int*>(ring + MT_RING);                  // words [0, avail) of the chunk exist
    int& next_unit = *reinterpret_cast<int*>(ring + MT_RING + 1);          // the next unit to claim
    int& gave_up = *reinterpret_cast<int*>(ring + MT_RING + 2);
    int* cur = reinterpret_cast<int*>(ring + MT_RING + 4);                 // cur[j]: the unit wave 1 + j is at (everything below it of that wave's is read)
    const int tid = threadIdx.x, c = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long beta0 = (long long)c * bpc;
    const long long left_blocks = beta_max - beta0;
    const int nblk = left_blocks < bpc ? (int)left_blocks : bpc;
    const long long draw0 = beta0 * MT_N - (MT_N - rem);                       // draw index of the chunk's word 0
    const long long n_groups = n / 16;
    const int phi = (16 - rem % 16) % 16;                                      // group starts: q = phi (mod 16)
    const int q_end = MT_N * (nblk + 1);
    const int n_steps = (q_end - MT_N + MT_STEP - 1) / MT_STEP;                // regeneration steps 1 .. n_steps
    const int n_regen = (n_steps + MT_UNIT - 1) / MT_UNIT;                     // units the regenerating wave runs
    const int n_units = n_regen > 0 ? n_regen : 1;                             // unit u: steps MT_UNIT u + 1 .. MT_UNIT (u + 1); unit 0 also the start state's groups

    for (int k = tid; k < MT_N; k += MT_GEN_THREADS) ring[k] = states[(size_t)c * MT_N + k];
    if (tid < MT_CONSUMERS) cur[tid] = 0;
    if (tid == 0) { avail = MT_N; next_unit = 0; gave_up = 0; }
    __syncthreads();

    if (wave == 0) {
        // ---- regeneration: lane l owns words q0 + 4 l .. q0 + 4 l + 3 of every step (lanes 56-63: the first 32 words of the next step's
        // slots, overwritten by that step); x[k] = x[k - 227] ^ twist(x[k - 624], x[k - 623]) ----
        // Step v = tw(v) ^ rot(z(v - 1)): the x[k - 227] words of lane l are words 1, 2, 3 of lane l - 1's previous output and word 0 of
        // its own - registers and a rotation of the wave by one lane, no LDS. Lane 0 needs the last three words of step v - 2 (lane 55's):
        // z(v) is w(v) with lanes 60 .. 63 replaced by lanes 52 .. 55 of z(v - 1), so that lane 63, where the rotation fetches lane 0's
        // words, holds lane 55's of a step earlier. tw(v + 1), the twists of the x[k - 624] side, needs steps <= v - 1 only: its words are
        // read behind step v - 1's write and twisted at the end of step v, a whole step after they were asked for. Nothing in the loop
        // waits for the LDS; per step one 16-byte write and one 16-byte read.
#if DASP_MT_PROBE != 3
        __builtin_amdgcn_s_setprio(3);                                         // the chain goes first on the SIMD it shares with a Box-Muller wave
#endif
        // (the ring is the kernel's only LDS object: its LDS address is 0, and byte offsets into its 64 KiB wrap as 16-bit sums)
        if ((unsigned)(unsigned long long)ring != 0u) __builtin_trap();
        static_assert(4 * MT_RING == 65536, "the regenerating wave's addresses are 16-bit sums");
        unsigned q_b = 4u * (MT_N + 4 * lane);                                 // byte offset (wrapped) of the lane's first word of the step
        mt_u32x4 A[2], z, tw;                                                  // A: the x[k - 624] side of the steps of either parity
        mt_regen_start<true>(lane, q_b, z, tw, A[0], A[1]);
        int cleared = 0;                                                       // every step <= cleared has been read
        static_assert(MT_UNIT % 2 == 0, "the registers of steps of one parity alternate inside a unit");
        for (int u = 0; u < n_regen; ++u) {
            for (int spin = 0; MT_UNIT * (u + 1) - MT_LEAD > cleared; ++spin) {                      // (asked for once per few units)
                int lowest = 1 << 28;
#pragma unroll
                for (int j = 0; j < MT_CONSUMERS; ++j) {
                    const int r = __hip_atomic_load(&cur[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    lowest = r < lowest ? r : lowest;
                }
                cleared = lowest >= (1 << 20) ? (1 << 28) : MT_UNIT * lowest;  // units are claimed in order: every unit below the lowest is finished
                if (MT_UNIT * (u + 1) - MT_LEAD > cleared) __builtin_amdgcn_s_sleep(2);
                if (spin > (1 << 22)) { gave_up = 1; break; }
            }
#pragma unroll
            for (int sub = 0; sub < MT_UNIT; ++sub)                            // step v = MT_UNIT u + sub + 1: its parity's register is refilled
                mt_regen_step<true>(q_b, z, tw, A[(sub + 1) & 1], A[sub & 1]);
            if (lane == 0) mt_lds_write32(4u * MT_RING, (unsigned)(MT_N + MT_STEP * MT_UNIT * (u + 1)));   // behind the data in this wave's LDS order
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]) :: "memory");
    } else {
        // ---- Box-Muller ----
        const int cj = wave - 1;
#if DASP_MT_PROBE == 4
        if (wave == 4) { if (lane == 0) cur[cj] = 1 << 28; } else
#endif
        for (;;) {
            int u = 0;
            if (lane == 0) u = __hip_atomic_fetch_add(&next_unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            u = __builtin_amdgcn_readfirstlane(u);
            if (lane == 0) __hip_atomic_store(&cur[cj], u < n_units ? u : (1 << 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (u >= n_units) break;
            const int s_hi = MT_UNIT * (u + 1) < n_steps ? MT_UNIT * (u + 1) : n_steps;
            const int hi = mt_step_end(s_hi, q_end);
            const int lo = u > 0 ? mt_step_end(MT_UNIT * u, q_end) : c == 0 ? MT_N - rem : MT_N;    // words [lo, hi) are this unit's
            int spin = 0;
            while (__hip_atomic_load(&avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < hi) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1 << 22)) { gave_up = 1; break; }
            }
            asm volatile("" ::: "memory");
            // the groups whose last word lies in this unit (their first may lie up to 15 words back)
            const int qs0 = u == 0 && c == 0 ? MT_N - rem : lo + (phi ? phi - 16 : 0);
            const int pairs = hi >= qs0 + 16 ? ((hi - qs0) >> 4) * 8 : 0;
#if DASP_MT_PROBE != 1
            // (every unit but the tensor's last lies wholly inside the tensor: no test per pair, and 32-bit offsets from the unit's first draw)
            float* __restrict__ o = out + (draw0 + qs0);
            const int pairs_in = (int)((n_groups - ((draw0 + qs0) >> 4) < (long long)(pairs >> 3) ? n_groups - ((draw0 + qs0) >> 4) : (long long)(pairs >> 3)) * 8);
            for (int p = lane; p < pairs_in; p += 64) {
                const int rel = 16 * (p >> 3) + (p & 7), qa = qs0 + rel;
                const float ua = mt_uniform(ring[qa & (MT_RING - 1)]), ub = mt_uniform(ring[(qa + 8) & (MT_RING - 1)]);
                const float rad = mt_radius(ua);
                float sn, co;
                mt_sincos(6.283185307179586f * ub, sn, co);
#if DASP_MT_PROBE == 2
                if (rad * co + rad * sn == 123.456f)
#endif
                { o[rel] = rad * co; o[rel + 8] = rad * sn; }
            }
#endif
            if ((n & 15) && lane < 16) {                                       // the 16 draws behind the tensor: kept as uniforms for mt_tail_kernel
                const long long q = n + lane - draw0;
                if (q >= lo && q < hi) tail_u[lane] = mt_uniform(ring[(int)q & (MT_RING - 1)]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's ring reads have returned before it says so (the next claim)
        }
    }
    __syncthreads();
    if (gave_up && tid == 0) {                                                 // (a sticky device error as for the look-back words: common.hpp)
        out[0] = __builtin_nanf("");
        if (err) __hip_atomic_store(err + DASP_DEVERR_MT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (c == gridDim.x - 1 && nblk > 0)
        for (int k = tid; k < MT_N; k += MT_GEN_THREADS) final_state[k] = ring[(MT_N * nblk + k) & (MT_RING - 1)];
}

__global__ void __launch_bounds__(64) mt_tail_kernel(const float* __restrict__ tail_u, float* __restrict__ out, long long n) {
    const int j = threadIdx.x;
    if (j < 8) {
        const float rad = mt_radius(tail_u[j]);
        float s, co;
        mt_sincos(6.283185307179586f * tail_u[j + 8], s, co);
        out[n - 16 + j] = rad * co;
        out[n - 8 + j] = rad * s;
    }
}

struct MtPlan { long long total, beta_max; int n_chunks, left_after, stride; };
MtPlan mt_plan(int left, long long n) {
    MtPlan p;
    const int rem = left - 1;
    p.total = n + ((n & 15) ? 16 : 0);
    const long long last_word = MT_N - rem + p.total - 1;
    p.beta_max = last_word / MT_N;
    p.left_after = (int)(MT_N * (p.beta_max + 1) - last_word);
    // One chunk per CU at most: a jump costs ~145 us of a CU whatever the chunk (256 of them, one round of the device, as much as one),
    // a chunk's generation ~85 us per unit of 256 regenerations when its workgroup has the CU to itself (the regenerating wave's
    // instruction issue) and no less per unit when several share it. (128,2,262144) (1,280 units): 640 chunks of 2 units were 31 + 436
    // us of jumps and 322 us of generation, 256 chunks of 5 units: see profiles/r06/README.md.
    const long long units = (p.beta_max + MT_BLOCKS_PER_CHUNK - 1) / MT_BLOCKS_PER_CHUNK;
    p.stride = units > MT_CHUNKS_TARGET ? (int)((units + MT_CHUNKS_TARGET - 1) / MT_CHUNKS_TARGET) : 1;
    const long long bpc = (long long)MT_BLOCKS_PER_CHUNK * p.stride;
    p.n_chunks = p.beta_max == 0 ? 1 : (int)((p.beta_max + bpc - 1) / bpc);
    return p;
}

}  // namespace
}  // namespace dasp

using namespace dasp;

extern "C" {

// {blocks per chunk, baby polynomials, giant polynomials, list slot, row stride (uint16), pad exponent, max chunks per call, 0}
int dasp_mt_layout(int* out8) {
    if (!out8) return DASP_ERR_ARG;
    const int v[8] = {MT_BLOCKS_PER_CHUNK, MT_N_BABY, MT_N_GIANT, MT_GROUP, MT_ROW, MT_PAD_INDEX, (MT_N_GIANT + 1) * (MT_N_BABY + 1), 0};
    for (int i = 0; i < 8; ++i) out8[i] = v[i];
    return DASP_OK;
}

// Largest n one call takes from any generator position (the loop over pieces is the caller's: pieces are multiples of 16).
long long dasp_mt_max_values(void) { return (long long)((MT_N_GIANT + 1) * (MT_N_BABY + 1) / 2 - 1) * 2 * MT_BLOCKS_PER_CHUNK * MT_N; }

// 32-bit words of device scratch for n values from a generator with `left`: chunk start states | the giant jumps' seven | state afterwards (624) | tail draws (16)
long dasp_mt_scratch_words(int left, long long n) {
    if (left < 1 || left > MT_N || n < 16) return -1;
    const MtPlan p = mt_plan(left, n);
    if ((long long)p.stride * (p.n_chunks - 1) >= (MT_N_GIANT + 1) * (MT_N_BABY + 1)) return -1;
    return (long)(p.n_chunks + MT_N_GIANT) * MT_N + MT_N + 16;
}

// out[0 .. n) <- what `torch.randn(n)` (float32, CPU, n >= 16) returns from the at::mt19937 state (state_host[624], left);
// *left_after <- the generator's `left` afterwards; the state words afterwards are scratch[n_chunks * 624 .. + 624) once the stream
// has run (unchanged - and not written - when *regenerated == 0: the draws fitted the current block). table: _mt19937.build_table()
// on the device. Asynchronous on `stream`; nothing is read back here.
int dasp_mt_randn(const unsigned* state_host, int left, float* out, long long n, const unsigned short* table, unsigned* scratch,
                  int* left_after, int* regenerated, long* final_state_offset_words, void* stream) {
    if (!state_host || !out || !table || !scratch || left < 1 || left > MT_N || n < 16) return DASP_ERR_ARG;
    const MtPlan p = mt_plan(left, n);
    if ((long long)p.stride * (p.n_chunks - 1) >= (MT_N_GIANT + 1) * (MT_N_BABY + 1)) return DASP_ERR_UNSUPPORTED;
    if (error_pending()) return DASP_ERR_DEVICE;                               // a wait that gave up in an earlier launch (sticky: dasp_device_error_clear)
    unsigned* err = error_words_device();
    hipStream_t st = (hipStream_t)stream;
    unsigned* states = scratch;
    unsigned* final_state = scratch + (size_t)(p.n_chunks + MT_N_GIANT) * MT_N;
    float* tail_u = reinterpret_cast<float*>(final_state + MT_N);
    MtState s;
    for (int k = 0; k < MT_N; ++k) s.w[k] = state_host[k];
    hipLaunchKernelGGL(mt_seed_kernel, dim3(p.n_chunks > 1 ? 1 + (p.n_chunks + MT_N_GIANT + 63) / 64 : 1), dim3(640), 0, st, s, states, p.n_chunks + MT_N_GIANT);
    {   // 83 KiB of LDS per workgroup: above the 64 KiB a kernel gets unasked - asked for once per device (the call costs the host ~10 us)
        static std::atomic<bool> asked[64];
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !asked[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MT_SEQ_LDS * 4);
            if (e != hipSuccess) return (int)e;
            if (MT_GEN_LDS > 65536) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_generate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MT_GEN_LDS);
                if (e != hipSuccess) return (int)e;
            }
            if (dev >= 0 && dev < 64) asked[dev].store(true, std::memory_order_release);
        }
    }
    // Several workgroups per jump where that is fewer rounds x time: a workgroup with 1/k of a jump's groups takes ~9.5 + 52.5 / k us (the
    // window it regenerates first is the 9.5; measured 62 / 27 / 16 us at k = 1 / 3 / 8), one workgroup per CU at a time (83 KiB of LDS).
    auto parts_for = [](int jobs) {
        int best = 1;
        float best_t = 1e30f;
        for (int k = 1; k <= 8; ++k) {
            const float t = (float)((jobs * k + 255) / 256) * (9.5f + 52.5f / (float)k);
            if (t < best_t - 0.5f) { best_t = t; best = k; }
        }
        return best;
    };
    const int giants = (p.stride * (p.n_chunks - 1)) >> 8;                     // the last chunk starts 256 giants + b units on
    if (giants > 0) {
        const int k = parts_for(giants);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(giants * k), dim3(MT_JUMP_THREADS), MT_SEQ_LDS * 4, st, states, table, 1, p.n_chunks, k, p.stride);
    }
    if (p.n_chunks > 1) {
        const int jobs = p.n_chunks - 1, k = parts_for(jobs);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(jobs * k), dim3(MT_JUMP_THREADS), MT_SEQ_LDS * 4, st, states, table, 0, p.n_chunks, k, p.stride);
    }
    hipLaunchKernelGGL(mt_generate_kernel, dim3(p.n_chunks), dim3(MT_GEN_THREADS), MT_GEN_LDS, st, states, out, n, left - 1, p.beta_max, final_state, tail_u,
                       MT_BLOCKS_PER_CHUNK * p.stride, err);
    if (n & 15) hipLaunchKernelGGL(mt_tail_kernel, dim3(1), dim3(64), 0, st, tail_u, out, n);
    if (left_after) *left_after = p.left_after;
    if (regenerated) *regenerated = p.beta_max > 0;
    if (final_state_offset_words) *final_state_offset_words = (long)(p.n_chunks + MT_N_GIANT) * MT_N;
    return (int)hipGetLastError();
}

}  // extern "C"
