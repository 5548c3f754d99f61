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
// x[n + i] (p: the characteristic polynomial, degree 19937; host side _mt19937.py). A CHUNK is 256 regenerations (159,744 words) and
// one workgroup of the generation kernel; the start state of chunk c is the state 256 c regenerations on, which mt_jump_kernel sums
// out of a 20,560-word window of the sequence held in LDS - from chunk 0's state for the chunks 256 a (7 "giant" polynomials, only
// above 256 chunks) and from chunk 256 a for the 255 behind it ("baby" polynomials), all workgroups of a phase side by side. A jump is
// ~10 k LDS reads of 8 bytes per lane (ds_read_b64: 256 B/clk/CU; lanes own two state words, coefficients split by parity so that
// every read is 8-byte aligned): LDS-bound, ~50 us on one CU. Then every chunk regenerates its blocks 224 words per step (any 227
// consecutive new words are independent), and Box-Muller runs on the same wave 448 words at a time. HBM: 4 B written per value.
#include "common.hpp"

namespace dasp {
namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_BLOCKS_PER_CHUNK = 256;               // ... or twice that for large draws (mt_plan: `stride` = 2), from the same table
constexpr int MT_N_BABY = 255, MT_N_GIANT = 7;
constexpr int MT_SLOT = 10112, MT_STRIDE = 8 + 2 * MT_SLOT;     // list slot per parity class: 79 batches of 128 exponents
constexpr int MT_PAD_INDEX = 20560;                    // = 19937 + 623: the sequence window of one jump
constexpr int MT_SEQ_LDS = MT_PAD_INDEX + 648;         // + zeros behind it: list padding reads them (base PAD_INDEX, lanes to 2*319+1)
constexpr int MT_JUMP_LANES = 320;                     // 313 lanes own two state words each (one more for the odd class's neighbour word)
#ifndef DASP_MT_SLICES
#define DASP_MT_SLICES 3
#endif
constexpr int MT_JUMP_SLICES = DASP_MT_SLICES;         // slices of 320 lanes share the exponent lists (batch b goes to slice b % SLICES)
constexpr int MT_JUMP_THREADS = MT_JUMP_SLICES * MT_JUMP_LANES;
constexpr int MT_GEN_THREADS = 512;                    // four regenerating waves + four Box-Muller waves
constexpr int MT_STEP = 224;                           // new words per regeneration step: a multiple of 16 not above 227
constexpr int MT_RING = 1024;                          // raw-word ring of a generating workgroup (a step looks 624 words back)

struct MtState { unsigned w[MT_N]; };                  // 2,496 bytes: travels as a kernel argument, no host-to-device copy

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7FFFFFFFu)) >> 1) ^ ((v & 1u) ? 0x9908B0DFu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ float mt_uniform(unsigned raw) { return (float)(mt_temper(raw) & 0xFFFFFFu) * 0x1p-24f; }

// chunk 0's state from the kernel argument; the other chunks' states start as zeros (a jump shared by several workgroups is summed
// into its state with atomic XORs)
__global__ void __launch_bounds__(640) mt_seed_kernel(MtState s, unsigned* __restrict__ states, int n_chunks) {
    if (blockIdx.x == 0) { if (threadIdx.x < MT_N) states[threadIdx.x] = s.w[threadIdx.x]; return; }
    for (size_t k = MT_N + (size_t)(blockIdx.x - 1) * 640 + threadIdx.x; k < (size_t)n_chunks * MT_N; k += (size_t)(gridDim.x - 1) * 640) states[k] = 0u;
}

// One jump: states[dst] = g(T) states[src]. giant: src = chunk 0, polynomial N_BABY + blockIdx, dst = 256 (blockIdx + 1);
// baby: a = blockIdx / 255, b = blockIdx % 255 + 1, src = 256 a, polynomial b - 1, dst = src + b. Word 0 of a jumped state is right in
// its top bit only - the one bit of it the recurrence reads. `parts` workgroups share a jump when there are fewer jumps than CUs (each
// takes every parts-th batch of the exponent lists; blockIdx = jump * parts + part).
__global__ void __launch_bounds__(MT_JUMP_THREADS)
mt_jump_kernel(unsigned* __restrict__ states, const unsigned short* __restrict__ table, int giant, int n_chunks, int parts, int stride) {
    extern __shared__ unsigned seq[];
    const int tid = threadIdx.x, job = blockIdx.x / parts, part = blockIdx.x % parts;
    int src, dst, poly;
    // (stride 2: chunks of 512 regenerations - a group is 128 chunks, chunk a * 128 + b lies a * 256 J + 2 b J words on: the same giant
    // polynomials, every other baby polynomial)
    const int gs = 256 / stride;
    if (giant) { src = 0; poly = MT_N_BABY + job; dst = gs * (job + 1); }
    else { const int a = job / (gs - 1), b = job % (gs - 1) + 1; src = gs * a; poly = stride * b - 1; dst = src + b; }
    if (dst >= n_chunks) return;

    for (int k = tid; k < MT_N; k += MT_JUMP_THREADS) seq[k] = states[(size_t)src * MT_N + k];
    for (int k = MT_PAD_INDEX + tid; k < MT_SEQ_LDS; k += MT_JUMP_THREADS) seq[k] = 0u;
    __syncthreads();
    // the window: 19,936 more words, 227 at a time
    for (int q = MT_N; q < MT_PAD_INDEX; q += 227) {
        const int k = q + tid;
        if (tid < 227 && k < MT_PAD_INDEX) seq[k] = seq[k - (MT_N - MT_M)] ^ mt_twist(seq[k - MT_N], seq[k - MT_N + 1]);
        __syncthreads();
    }

    // The exponent lists: 128 exponents (64 dwords) per wave and batch, one dword per lane (vector memory: its counter is not the
    // LDS's - as scalar loads the list put one L2 round trip into every group of eight reads: 187 us per jump), handed to the whole
    // wave lane by lane (v_readlane). Two halves of 320 lanes take alternate batches: ten waves hide the LDS latency better than five,
    // the LDS bandwidth is the same. Lanes 313 .. 319 of a half run along on zeros and garbage inside the padded window.
    const unsigned* row = reinterpret_cast<const unsigned*>(table + (size_t)poly * MT_STRIDE);
    const int half = __builtin_amdgcn_readfirstlane(tid / MT_JUMP_LANES), t = tid % MT_JUMP_LANES, lane = tid & 63;      // 320 lanes = five whole waves
    unsigned e0 = 0u, e1 = 0u, o0 = 0u, o1 = 0u;
    const uint2* win = reinterpret_cast<const uint2*>(seq) + t;                  // words 2 t, 2 t + 1 of the window at exponent 0
#define MT_ACC(a0, a1, word, sub)                                                              \
    { const uint2 v0 = win[(((word) & 0xFFFFu) - (sub)) >> 1], v1 = win[(((word) >> 16) - (sub)) >> 1]; \
      a0 ^= v0.x ^ v1.x; a1 ^= v0.y ^ v1.y; }
#define MT_CLASS(a0, a1, list, count, sub)                                                     \
    { const unsigned* lp = (list);                                                             \
      const int nb = (int)(count) / 128;                                                       \
      const int first = MT_JUMP_SLICES * part + half, step = MT_JUMP_SLICES * parts;           \
      if (first < nb) {                                                                        \
          unsigned nxt = lp[64 * first + lane];                                                \
          for (int bt = first; bt < nb; bt += step) {                                          \
              const unsigned cur = nxt;                                                        \
              nxt = lp[64 * (bt + step < nb ? bt + step : bt) + lane];    /* the next batch, asked for before this batch's reads */ \
              _Pragma("unroll 8")                                                              \
              for (int k = 0; k < 64; ++k) {                                                   \
                  const unsigned w = __builtin_amdgcn_readlane(cur, k);                        \
                  MT_ACC(a0, a1, w, sub)                                                       \
              }                                                                                \
          }                                                                                    \
      } }
    MT_CLASS(e0, e1, row + 4, row[0], 0u)
    MT_CLASS(o0, o1, row + 4 + MT_SLOT / 2, row[1], 1u)                          // odd exponent i: the aligned pair one word below
#undef MT_CLASS
#undef MT_ACC
    // fold the halves, then: word 2 t = even sum + the odd class's UPPER word of this lane; word 2 t + 1 = even sum + the odd class's
    // LOWER word of lane t + 1
    __syncthreads();
    if (half > 0) { unsigned* d = seq + 4 * ((half - 1) * MT_JUMP_LANES + t); d[0] = e0; d[1] = e1; d[2] = o0; d[3] = o1; }
    __syncthreads();
    if (half == 0)
        for (int h = 1; h < MT_JUMP_SLICES; ++h) {
            const unsigned* d = seq + 4 * ((h - 1) * MT_JUMP_LANES + t);
            e0 ^= d[0]; e1 ^= d[1]; o0 ^= d[2]; o1 ^= d[3];
        }
    __syncthreads();
    if (half == 0) seq[t] = o0;
    __syncthreads();
    if (half == 0 && t < MT_N / 2) {
        unsigned* out = states + (size_t)dst * MT_N + 2 * t;
        const unsigned w0 = e0 ^ o1, w1 = e1 ^ seq[t + 1];
        if (parts == 1) { out[0] = w0; out[1] = w1; }
        else { atomicXor(out, w0); atomicXor(out + 1, w1); }
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
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// One chunk: blocks 256 c + 1 .. of the sequence from states[c] (block 256 c), every aligned group of 16 draws whose last word lies
// in those blocks (chunk 0: also the groups inside the state it starts from), the 16 tail draws it owns, and - the last chunk - the
// generator state afterwards. word q of the chunk (q = 0 .. 623: the start state) is draw 624 (256 c) + q - (624 - rem).
// Eight waves in two roles: waves 0-3 regenerate (two steps of 224 words per round, a barrier behind each - the serial chain of the
// chunk), waves 4-7 turn the 448 words of the PREVIOUS round into normals meanwhile (radius before the middle barrier, angle and stores
// behind it). One role for everything was 1,380 cycles per round, the regeneration's LDS round trips and the Box-Muller arithmetic one
// after the other on every wave: 235 us per chunk.
struct MtPair { float rad, ang; long long i; bool on; };
__device__ __forceinline__ MtPair mt_pair_radius(const unsigned* ring, int qa, long long draw0, long long n_groups, bool on) {
    MtPair r;
    r.i = draw0 + qa;
    r.on = on && (r.i >> 4) < n_groups;
    const float ua = mt_uniform(ring[qa & (MT_RING - 1)]), ub = mt_uniform(ring[(qa + 8) & (MT_RING - 1)]);
    r.rad = mt_radius(ua);
    r.ang = 6.283185307179586f * ub;
    return r;
}
__device__ __forceinline__ void mt_pair_store(const MtPair& r, float* __restrict__ out) {
    float s, co;
    mt_sincos(r.ang, s, co);
    if (r.on) { out[r.i] = r.rad * co; out[r.i + 8] = r.rad * s; }
}

__global__ void __launch_bounds__(MT_GEN_THREADS)
mt_generate_kernel(const unsigned* __restrict__ states, float* __restrict__ out, long long n, int rem, long long beta_max,
                   unsigned* __restrict__ final_state, float* __restrict__ tail_u, int bpc) {
    __shared__ unsigned ring[MT_RING];
    const int tid = threadIdx.x, c = blockIdx.x, lt = tid & 255;
    const bool producer = tid < 256;                                           // waves 0-3
    const long long beta0 = (long long)c * bpc;
    const long long left_blocks = beta_max - beta0;
    const int nblk = left_blocks < bpc ? (int)left_blocks : bpc;
    const long long draw0 = beta0 * MT_N - (MT_N - rem);                       // draw index of the chunk's word 0
    const long long n_groups = n / 16;
    const int phi = (16 - rem % 16) % 16;                                      // group starts: q = phi (mod 16)

    for (int k = tid; k < MT_N; k += MT_GEN_THREADS) ring[k] = states[(size_t)c * MT_N + k];
    __syncthreads();

    const int q_end = MT_N * (nblk + 1);
    int q_gen = MT_N;                                                          // words [0, q_gen) exist
    int q_done = c == 0 ? MT_N - rem : 608 + phi + (phi == 0 ? 16 : 0);        // next group start (first group ending inside block 1)
    int q_tail = c == 0 ? MT_N - rem : MT_N;                                   // tail draws are looked for in [q_tail, q_gen)
    for (;;) {
        const int pairs = ((q_gen - q_done) >> 4) * 8;                         // complete groups in [q_done, q_gen): 224 per round (chunk 0's first: up to 304)
        const bool more = q_gen < q_end;
        const int cnt_a = !more ? 0 : q_end - q_gen < MT_STEP ? q_end - q_gen : MT_STEP;
        const int cnt_b = q_end - q_gen - cnt_a < MT_STEP ? q_end - q_gen - cnt_a : MT_STEP;
        MtPair pr;
        if (producer) {
            const int q = q_gen + lt;
            if (lt < cnt_a)
                ring[q & (MT_RING - 1)] = ring[(q - (MT_N - MT_M)) & (MT_RING - 1)]
                                          ^ mt_twist(ring[(q - MT_N) & (MT_RING - 1)], ring[(q - MT_N + 1) & (MT_RING - 1)]);
        } else {
            for (int p = lt + 256; p < pairs; p += 256)                        // (only chunk 0's first round has more than 256 pairs)
                mt_pair_store(mt_pair_radius(ring, q_done + 16 * (p >> 3) + (p & 7), draw0, n_groups, true), out);
            pr = mt_pair_radius(ring, q_done + 16 * (lt >> 3) + (lt & 7), draw0, n_groups, lt < pairs);
            if ((n & 15) && lt < 16) {                                         // the 16 draws behind the tensor: kept as uniforms for mt_tail_kernel
                const long long q = n + lt - draw0;
                if (q >= q_tail && q < q_gen) tail_u[lt] = mt_uniform(ring[(int)q & (MT_RING - 1)]);
            }
        }
        __syncthreads();
        if (producer) {
            const int q = q_gen + cnt_a + lt;
            if (lt < cnt_b)
                ring[q & (MT_RING - 1)] = ring[(q - (MT_N - MT_M)) & (MT_RING - 1)]
                                          ^ mt_twist(ring[(q - MT_N) & (MT_RING - 1)], ring[(q - MT_N + 1) & (MT_RING - 1)]);
        } else {
            mt_pair_store(pr, out);
        }
        __syncthreads();
        q_done += 2 * pairs;
        q_tail = q_gen;
        q_gen += cnt_a + cnt_b;
        if (!more) break;
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
    // chunks of 256 regenerations; twice that from 512 chunks on: a jump costs ~170 us of a CU whatever the chunk, the generation of a
    // chunk ~190 us per 256 regenerations - at (128,2,262144) (327,670 regenerations) 1,279 jumps are five rounds of the device, 639 are three
    // (measured: jumps 0.89 -> 0.47 ms at an unchanged 0.47 ms of generation, profiles/r06/mtrand_kernel_stats_b128.csv)
    const long long w256 = (p.beta_max + MT_BLOCKS_PER_CHUNK - 1) / MT_BLOCKS_PER_CHUNK;
    p.stride = w256 > 512 ? 2 : 1;
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
    const int v[8] = {MT_BLOCKS_PER_CHUNK, MT_N_BABY, MT_N_GIANT, MT_SLOT, MT_STRIDE, MT_PAD_INDEX, (MT_N_GIANT + 1) * (MT_N_BABY + 1), 0};
    for (int i = 0; i < 8; ++i) out8[i] = v[i];
    return DASP_OK;
}

// Largest n one call takes from any generator position (the loop over pieces is the caller's: pieces are multiples of 16).
long long dasp_mt_max_values(void) { return (long long)((MT_N_GIANT + 1) * (MT_N_BABY + 1) / 2 - 1) * 2 * MT_BLOCKS_PER_CHUNK * MT_N; }

// 32-bit words of device scratch for n values from a generator with `left`: chunk start states | state afterwards (624) | tail draws (16)
long dasp_mt_scratch_words(int left, long long n) {
    if (left < 1 || left > MT_N || n < 16) return -1;
    const MtPlan p = mt_plan(left, n);
    if (p.n_chunks > (MT_N_GIANT + 1) * (MT_N_BABY + 1) / p.stride) return -1;
    return (long)p.n_chunks * MT_N + MT_N + 16;
}

// out[0 .. n) <- what `torch.randn(n)` (float32, CPU, n >= 16) returns from the at::mt19937 state (state_host[624], left);
// *left_after <- the generator's `left` afterwards; the state words afterwards are scratch[n_chunks * 624 .. + 624) once the stream
// has run (unchanged - and not written - when *regenerated == 0: the draws fitted the current block). table: _mt19937.build_table()
// on the device. Asynchronous on `stream`; nothing is read back here.
int dasp_mt_randn(const unsigned* state_host, int left, float* out, long long n, const unsigned short* table, unsigned* scratch,
                  int* left_after, int* regenerated, long* final_state_offset_words, void* stream) {
    if (!state_host || !out || !table || !scratch || left < 1 || left > MT_N || n < 16) return DASP_ERR_ARG;
    const MtPlan p = mt_plan(left, n);
    if (p.n_chunks > (MT_N_GIANT + 1) * (MT_N_BABY + 1) / p.stride) return DASP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned* states = scratch;
    unsigned* final_state = scratch + (size_t)p.n_chunks * MT_N;
    float* tail_u = reinterpret_cast<float*>(final_state + MT_N);
    MtState s;
    for (int k = 0; k < MT_N; ++k) s.w[k] = state_host[k];
    hipLaunchKernelGGL(mt_seed_kernel, dim3(p.n_chunks > 1 ? 1 + (p.n_chunks + 63) / 64 : 1), dim3(640), 0, st, s, states, p.n_chunks);
    {   // 83 KiB of LDS per workgroup: above the 64 KiB a kernel gets unasked (per device, so every call)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MT_SEQ_LDS * 4);
        if (e != hipSuccess) return (int)e;
    }
    auto parts_for = [](int jobs) { const int k = 256 / jobs; return k < 1 ? 1 : k > 8 ? 8 : k; };      // fewer jumps than CUs: several workgroups per jump
    const int gs = 256 / p.stride;                  // chunks per group (one giant jump each)
    if (p.n_chunks > gs) {
        const int jobs = (p.n_chunks - 1) / gs, k = parts_for(jobs);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(jobs * k), dim3(MT_JUMP_THREADS), MT_SEQ_LDS * 4, st, states, table, 1, p.n_chunks, k, p.stride);
    }
    if (p.n_chunks > 1) {
        const int jobs = p.n_chunks > gs ? ((p.n_chunks + gs - 1) / gs) * (gs - 1) : p.n_chunks - 1, k = parts_for(jobs);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(jobs * k), dim3(MT_JUMP_THREADS), MT_SEQ_LDS * 4, st, states, table, 0, p.n_chunks, k, p.stride);
    }
    hipLaunchKernelGGL(mt_generate_kernel, dim3(p.n_chunks), dim3(MT_GEN_THREADS), 0, st, states, out, n, left - 1, p.beta_max, final_state, tail_u,
                       MT_BLOCKS_PER_CHUNK * p.stride);
    if (n & 15) hipLaunchKernelGGL(mt_tail_kernel, dim3(1), dim3(64), 0, st, tail_u, out, n);
    if (left_after) *left_after = p.left_after;
    if (regenerated) *regenerated = p.beta_max > 0;
    if (final_state_offset_words) *final_state_offset_words = (long)p.n_chunks * MT_N;
    return (int)hipGetLastError();
}

}  // extern "C"
