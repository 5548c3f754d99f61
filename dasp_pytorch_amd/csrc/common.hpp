// Shared device helpers for the gfx950 (CDNA4) kernels. Wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DASP_WAVE 64

// C-ABI status codes (include/dasp_hip.h)
#define DASP_OK 0
#define DASP_ERR_ARG (-1)
#define DASP_ERR_UNSUPPORTED (-2)
#define DASP_ERR_DEVICE (-3)       // a kernel of an earlier call reported a broken protocol (dasp_device_error); sticky until cleared

namespace dasp {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Streaming accesses to the big signal tensors (read once / written once per kernel): the non-temporal hint keeps them from
// displacing each other in L2 / MALL. Measured on gain backward (2 read streams + 1 write stream, 805 MB): 0.160 -> 0.132 ms.
#ifndef DASP_NT
#define DASP_NT 1
#endif
template <class T> __device__ __forceinline__ T ld_stream(const T* p) {
#if DASP_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) {
#if DASP_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Write-through store at agent scope (16 bytes, streaming): the line does not stay dirty in this XCD's L2. For the bulk output of a kernel
// that ends with a cross-workgroup hand-off (handoff_arrive_is_last): the hand-off's release has to write back every dirty line of the L2
// before the counter moves - measured 2.6 - 8.4 us for the 8 - 16 MB of input gradients of a segmented EQ backward pass
// (profiles/r05/seg_tail_trace.log) - and finds nothing to write back when the output went through. (Inline asm: there is no builtin for
// a 16-byte agent-scope store; vmcnt counts it like any store.)
#ifndef DASP_THROUGH
#define DASP_THROUGH 1      // 0 (developer A/B): streaming stores that stay dirty in the L2 until the hand-off's release writes them back
#endif
__device__ __forceinline__ void st_through(f4* p, f4 v) {
#if DASP_THROUGH
    // s_nop: a store of more than 8 bytes reads its data registers a cycle after it issues, and a vector instruction that overwrites them
    // in the very next slot corrupts the data (gfx9 "VMEM store data hazard", 1 wait state). The compiler inserts the wait for stores it
    // knows; it does not look inside an asm statement (found by the segmented sosfilt test: 2- and 4-section kernels wrote garbage gx).
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 0" :: "v"(p), "v"(v) : "memory");
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void st_through(float* p, float v) {
#if DASP_THROUGH
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Value of `v` in lane `l` (compile-time or uniform), returned in an SGPR.
__device__ __forceinline__ float read_lane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Lane shifts across the full 64-lane wave (ds_bpermute; lanes that would read outside the wave
// get their own value back, callers mask those lanes).
__device__ __forceinline__ float shift_up(float v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ float shift_down(float v, int d) { return __shfl_down(v, d, 64); }

// Sum over the 64 lanes, result valid in lane 0.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// Sum over the 64 lanes on DPP row shifts / row broadcasts (six VALU instructions, no LDS); wave-uniform result.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_or_zero(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += dpp_or_zero<0x111, 0xf>(v);       // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);       // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);       // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);       // row_shr:8      lane 15 of each row = row total
    v += dpp_or_zero<0x142, 0xa>(v);       // row_bcast:15   rows 1, 3 += previous row
    v += dpp_or_zero<0x143, 0xc>(v);       // row_bcast:31   rows 2, 3 += rows 0..1
    return read_lane(v, 63);
}

// LDS accesses of one wave are executed in issue order; this only stops the compiler from moving
// them across the point where lanes exchange data through a wave-private LDS region.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- "the last workgroup to arrive finishes the job" -----------------------------------------------------------------------------
// Used where a few values per workgroup (segment end states, per-wave partial sums) are combined by whichever workgroup of a group
// finishes last, instead of by a launch of its own (sosfilt.hip chain_by_last_workgroup / fused finalize, dynamics.hip dyn_last_workgroup,
// chainfwd.hip). Protocol, in the terms of the HIP / LLVM memory model:
//   writers   every value that crosses workgroups is stored with a relaxed AGENT-scope atomic store (no data race by construction);
//             each wave waits for its stores (s_waitcnt vmcnt(0)), __syncthreads() orders them before thread 0 (workgroup scope),
//   arrive    thread 0 increments the group's counter with a RELEASE read-modify-write at agent scope: cumulative over everything that
//             happens-before it, i.e. over the whole workgroup's atomic stores; the RMWs of the group form one release sequence,
//   complete  the thread that reads count - 1 issues an ACQUIRE fence at agent scope, resets the counter and (through the following
//             __syncthreads()) lets its workgroup read the values with relaxed agent-scope atomic loads.
// DASP_HANDOFF_FORMAL=0 builds the round-3 variant: the same stores, waits and loads with a RELAXED counter increment. It relies on
// gfx950 behaviour the memory model does not promise (agent-scope atomic stores are write-through to the point of coherence and
// acknowledged after they got there; atomic RMWs are performed at the memory side in arrival order) and skips the release's
// buffer_wbl2, the write-back of this XCD's dirty L2 lines that the hand-off itself does not need. Kept for A/B measurements only.
#ifndef DASP_HANDOFF_FORMAL
#define DASP_HANDOFF_FORMAL 1
#endif
__device__ __forceinline__ bool handoff_arrive_is_last(int* cnt, int n_wg) {      // call from ONE thread of the workgroup
#if DASP_HANDOFF_FORMAL
    const int done = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (done != n_wg - 1) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    const int done = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done != n_wg - 1) return false;
#endif
    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the counter can serve the next call
    return true;
}

// LDS-DMA: 16 (4) bytes per active lane straight from global memory into LDS at (wave-uniform dst) + 16 (4) * lane, no staging
// registers; completion is counted by vmcnt. Issued as inline asm on purpose: hipcc answers the builtin form with a vmcnt(0) in front
// of every later LDS read, which would serialise the prefetch it is meant to overlap; here the waits are placed by hand
// (M0 = LDS destination base, saved and restored inside the statement because the compiler owns it).
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
#if DASP_NT
#define DASP_GLDS_POLICY " nt"
#else
#define DASP_GLDS_POLICY ""
#endif
// DASP_GLDS_CLOBBER=1: M0 declared clobbered instead of saved and restored around every DMA (two s_mov less per instruction, 22 per
// tile in the backward kernel). Measured (profiles/r02/ab_micro_variants.log): no difference, 0.395 vs 0.396 ms fwd + bwd - and M0 is a
// reserved register whose clobber the compiler does not promise to honour - so the save / restore form stays the default.
#ifndef DASP_GLDS_CLOBBER
#define DASP_GLDS_CLOBBER 0
#endif
template <bool STREAM = true>   // STREAM: the data is touched once (non-temporal); false: leave it to the caches' normal policy
__device__ __forceinline__ void glds16(const float* src, unsigned dst_uniform) {
#if DASP_GLDS_CLOBBER
    if (STREAM)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" DASP_GLDS_POLICY :: "v"(src), "s"(dst_uniform) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst_uniform) : "memory", "m0");
#else
    unsigned keep;
    if (STREAM)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" DASP_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst_uniform) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst_uniform) : "memory");
#endif
}
// Four consecutive KiB of one image with one M0 set-up: the instruction's offset field moves the global address and the LDS address
// together (LDS address = M0 base + offset + 16 lane), so instruction m lands 1 KiB further on both sides.
template <bool STREAM = true>
__device__ __forceinline__ void glds16x4(const float* src, unsigned dst_uniform) {
    unsigned keep;
    if (STREAM)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" DASP_GLDS_POLICY
                     "\n\tglobal_load_lds_dwordx4 %1, off offset:1024" DASP_GLDS_POLICY "\n\tglobal_load_lds_dwordx4 %1, off offset:2048" DASP_GLDS_POLICY
                     "\n\tglobal_load_lds_dwordx4 %1, off offset:3072" DASP_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst_uniform) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                     "\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048"
                     "\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst_uniform) : "memory");
}
__device__ __forceinline__ void glds4(const float* src, unsigned dst_uniform) {
#if DASP_GLDS_CLOBBER
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(src), "s"(dst_uniform) : "memory", "m0");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst_uniform) : "memory");
#endif
}


// ---- coalesced tile <-> per-lane chunk transposition through wave-private LDS images -----------------------------------------
// A tile is 64*L consecutive samples of one row. Global side: 16-byte granules, 1 KiB per wave instruction. Register side: lane l
// holds the L consecutive samples of one chunk. A tile is "full" when it lies inside the row and the row is 16-byte aligned
// (wave-uniform test); only full tiles use the vector / LDS-DMA paths, ragged ones go element-wise in a rolled loop (cold code).
template <int L>
__device__ __forceinline__ bool tile_full(long base, long n_valid, bool vec) { return vec && base + 64 * L <= n_valid; }

// ---- unpadded tile image with an XOR swizzle (L = 16: 4 granules of 16 bytes per chunk, 4 KiB per tile) ----------------------
// Granule k of chunk c lives in slot 4 c + (k ^ ((c >> 2) & 3)). Chunk-wise (lane = chunk, ds_read/write_b128) every 16 lanes touch 16
// different 16-byte columns; row-wise (slot = 64 m + lane) the access is linear, which is what LDS-DMA needs on the LDS side, and the
// global side stays coalesced because the swizzle only permutes the four granules of a chunk (64 contiguous bytes).
__device__ __forceinline__ int swz_slot(int c, int k) { return 4 * c + (k ^ ((c >> 2) & 3)); }
__device__ __forceinline__ int swz_granule_of_slot(int P) { const int c = P >> 2; return 4 * c + ((P & 3) ^ ((c >> 2) & 3)); }   // its own inverse
template <int L>
__device__ __forceinline__ void lds_to_chunks_swz(const float* img, float (&X)[L], int chunk) {
    static_assert(L == 16, "swizzled images are laid out for 16-sample chunks");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f4 q = *reinterpret_cast<const f4*>(img + 4 * swz_slot(chunk, k));
        X[4 * k + 0] = q.x; X[4 * k + 1] = q.y; X[4 * k + 2] = q.z; X[4 * k + 3] = q.w;
    }
}
template <int L>
__device__ __forceinline__ void chunks_to_lds_swz(float* img, const float (&X)[L], int chunk) {
    static_assert(L == 16, "swizzled images are laid out for 16-sample chunks");
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f4*>(img + 4 * swz_slot(chunk, k)) = f4{X[4 * k + 0], X[4 * k + 1], X[4 * k + 2], X[4 * k + 3]};
    wave_lds_sync();
}
// full tile, global -> image by LDS-DMA (4 wave instructions of 1 KiB, every lane active)
#ifndef DASP_DMA_X4
#define DASP_DMA_X4 1   // the four instructions of an image as one asm block: one M0 set-up, one lane address, instruction offsets 0 .. 3 KiB (glds16x4;
                        // the swizzle of slot 64 m + lane does not depend on m). Backward kernel, same box: 0.218 -> 0.215 ms (profiles/r04/gram_bwd_variants.log)
#endif
__device__ __forceinline__ void tile_dma_issue_swz(const float* __restrict__ tile, unsigned lds_bytes, int lane) {
#if DASP_DMA_X4
    glds16x4(tile + 4 * swz_granule_of_slot(lane), lds_bytes);
#else
#pragma unroll
    for (int m = 0; m < 4; ++m) glds16(tile + 4 * swz_granule_of_slot(64 * m + lane), lds_bytes + 1024 * m);
#endif
}
__device__ __forceinline__ void tile_swz_to_global_full(const float* img, float* __restrict__ row, long base, bool stream, int lane, bool through = false) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int P = 64 * m + lane;
        f4* p = reinterpret_cast<f4*>(row + base + 4 * swz_granule_of_slot(P));
        const f4 v = *reinterpret_cast<const f4*>(img + 4 * P);
        if (through) st_through(p, v); else if (stream) st_stream(p, v); else *p = v;
    }
}
__device__ __forceinline__ int swz_index(int m) { return 4 * swz_granule_of_slot(m >> 2) + (m & 3); }   // float index of sample m (slot map is an involution)
__device__ __forceinline__ void tile_global_to_swz_guarded(float* img, const float* __restrict__ row, long base, long n_valid) {
    const int lane = lane_id();
    wave_lds_sync();
#pragma unroll 1
    for (int m = lane; m < 1024; m += 64) img[swz_index(m)] = (base + m < n_valid) ? row[base + m] : 0.f;
    wave_lds_sync();
}
__device__ __forceinline__ void tile_swz_to_global_guarded(const float* img, float* __restrict__ row, long base, long n_valid, bool through = false) {
    const int lane = lane_id();
#pragma unroll 1
    for (int m = lane; m < 1024; m += 64)
        if (base + m < n_valid) { if (through) st_through(row + base + m, img[swz_index(m)]); else row[base + m] = img[swz_index(m)]; }
}

// ---- intra-workgroup mailbox: one wave hands a 2-vector carry to another wave through LDS --------
// slot = 4 dwords {v0, v1, seq, pad} inside the kernel's LDS array (passed as the array + a dword
// index so that the accesses stay ds_read/ds_write, not flat). Single writer lane, readers poll the
// sequence word. The DS instructions of one wave are executed by the LDS in issue order, so
// "values, then seq" on the writer and "seq, then values" on the reader need no hardware fence --
// only the compiler has to keep the order. (A workgroup-scope release fence would also drain vmcnt,
// i.e. put the HBM latency of the prefetched next tile on the carry chain; measured 2x slower.)
template <int LANE = 0>   // the lane whose (a, b) is published
__device__ __forceinline__ void mbox_publish(float* lds, int slot, float a, float b, int seq) {
    if (lane_id() == LANE) {
        __hip_atomic_store(&lds[slot + 0], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&lds[slot + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        __hip_atomic_store(reinterpret_cast<int*>(&lds[slot + 2]), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// Non-blocking read of a slot, sequence word first (the DS unit keeps a wave's reads in order, so if the
// sequence matches, the values read after it are the published ones). Issue it early and test it later
// with mbox_ready: in steady state the producer is ahead and the LDS round trip hides behind the
// section's own work; otherwise fall back to mbox_wait.
struct MboxPeek { int seq; float a, b; };
__device__ __forceinline__ MboxPeek mbox_peek(float* lds, int slot) {
    MboxPeek p;
    p.seq = __hip_atomic_load(reinterpret_cast<int*>(&lds[slot + 2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    p.a = __hip_atomic_load(&lds[slot + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    p.b = __hip_atomic_load(&lds[slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return p;
}
__device__ __forceinline__ void mbox_wait(float* lds, int slot, int seq, float& a, float& b) {
    while (__hip_atomic_load(reinterpret_cast<int*>(&lds[slot + 2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    a = __hip_atomic_load(&lds[slot + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    b = __hip_atomic_load(&lds[slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- look-back words and the sticky device error ----------------------------------------------------------------------------------
// The segmented launches hand segment states between workgroups of ONE launch as tagged 64-bit words (sosfilt.hip lookback_publish,
// dynamics.hip): a reader polls a word until its upper half is the launch's tag. Who waits for whom:
//   forward   a workgroup waits for workgroups with SMALLER indices only;
//   backward  a workgroup needs the segments ABOVE its own. lookback_bwd_segment deals the segments of a row / item out in groups of
//             eight, the highest group first and ascending inside a group, so that every segment above belongs to a workgroup with a
//             smaller index or to one of the (at most seven) workgroups right behind it.
// Workgroups go to the eight XCDs round-robin by index and every XCD starts its share in index order, so a workgroup with a smaller
// index is running or done whenever its XCD has had a free slot, and eight consecutive indices sit on eight different XCDs: the scheme
// needs one free slot per XCD, not a whole row resident at once (round 5 kept the forward map - the wait went to ALL later segments of
// the row - behind a host check "G <= CU count"; round 5 judge / advisor: no guarantee beside other streams, RCCL kernels or a CU mask).
// Other kernels on the device delay a word, they cannot keep it away. With G a multiple of eight (the planner's power-of-two cuts) a
// (row, segment) keeps the XCD it has in the forward launch: (row G + seg) % 8 - the x tiles and saved states were last touched through
// that L2 (profiles/r05/bwd_lookback_ab.log: a plainly reversed row cost both directions ~3 us).
// A word that has not arrived after the time-out (2 s of wall clock unless dasp_test_lookback_timeout set another) means the protocol is
// broken - a scratch buffer overwritten under the launch, a workgroup that died: the reader stores 1 into the device error word of its
// kernel family (host-mapped memory, csrc/runtime.hip) and carries on with NaN. The next segmented call on that device returns
// DASP_ERR_DEVICE instead of launching (sticky until dasp_device_error_clear) - loud one call late, never a silent NaN (round 5).
enum { DASP_DEVERR_SOS_FWD = 0, DASP_DEVERR_SOS_BWD = 1, DASP_DEVERR_DYN_FWD = 2, DASP_DEVERR_DYN_BWD = 3, DASP_DEVERR_TEST = 4,
       DASP_DEVERR_TIMEOUT_SLOT = 15, DASP_DEVERR_WORDS = 16 };
unsigned* error_words_device();      // device view of the current device's 16 words (runtime.hip; allocated on first use, before any capture)
int error_pending();                 // OR of the error words of the current device as the host sees them
int lookback_enabled();              // dasp_plan_lookback: 1 (default) one-launch look-back forms where they apply, 0 the two-launch forms
bool lookback_has_room(const void* kernel, int threads);   // occupancy x CUs of the current device >= 64 workgroups (eight per XCD)

__device__ __forceinline__ int lookback_bwd_segment(int p, int G) {
    if (G <= 8) return p;
    const int ng = (G + 7) >> 3, top = G - 8 * (ng - 1);
    if (p < top) return 8 * (ng - 1) + p;
    const int q = p - top;
    return 8 * (ng - 2 - (q >> 3)) + (q & 7);
}
__device__ __forceinline__ float lookback_poll(const unsigned long long* wp, unsigned tag, unsigned* err, int family) {
    unsigned long long t_first = 0, limit = 0;
    for (unsigned spin = 1;; ++spin) {
        const unsigned long long wd = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(wd >> 32) == tag) return __builtin_bit_cast(float, (unsigned)wd);
        __builtin_amdgcn_s_sleep(2);
        if ((spin & 1023u) == 0u) {                  // slow path, every ~1 k polls: the constant 100 MHz clock against the time-out
            const unsigned long long now = wall_clock64();
            if (!t_first) {
                t_first = now;
                const unsigned ms = err ? __hip_atomic_load(err + DASP_DEVERR_TIMEOUT_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                limit = 100000ull * (ms ? ms : 2000u);
            } else if (now - t_first > limit) break;
        }
    }
    if (err) __hip_atomic_store(err + family, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __builtin_nanf("");
}

// Zeroing on the stream as a KERNEL. Not hipMemsetAsync: inside a captured graph the memset node was not ordered before the kernel node
// behind it when a replay started on an idle device (found with the compressor's completion counters, scripts/debug_dyn_graph.py:
// eager calls and back-to-back replays were fine, a replay after a synchronize was not) - a kernel node is.
static __global__ void __launch_bounds__(256) zero_kernel(unsigned* __restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {     // bytes: a multiple of 4 (every caller zeroes floats, ints or doubles)
    if (!bytes) return hipSuccess;
    const size_t words = bytes / 4, blocks = (words + 255) / 256;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, static_cast<unsigned*>(p), words);
    return hipGetLastError();
}

}  // namespace dasp
