// Shared device helpers for the gfx950 (CDNA4) kernels. Wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DASP_WAVE 64

// C-ABI status codes (include/dasp_hip.h)
#define DASP_OK 0
#define DASP_ERR_ARG (-1)
#define DASP_ERR_UNSUPPORTED (-2)

namespace dasp {

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

// LDS accesses of one wave are executed in issue order; this only stops the compiler from moving
// them across the point where lanes exchange data through a wave-private LDS region.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- coalesced tile <-> per-lane chunk transposition through a wave-private LDS buffer ---------
// A tile is 64*L consecutive samples of one row. Global side: lane l, vector j holds samples
// [j*256 + 4l, +4) (1 KiB per wave instruction). Register side: lane l holds the L consecutive
// samples [l*L, (l+1)*L). LDS image: chunk-major with a 4-float pad per chunk (stride L+4) which
// makes both the ds_write_b128 (8-lane groups) and the ds_read_b128 (16-lane groups) conflict free
// for L = 8, 16, 32.

template <int L>
__device__ __forceinline__ void tile_load_global(const float* __restrict__ row, long base, long n_valid,
                                                 bool vec, float4 (&v)[L / 4]) {
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < L / 4; ++j) {
        const long idx = base + (long)(j * 64 + lane) * 4;
        if (vec && idx + 3 < n_valid) {
            v[j] = *reinterpret_cast<const float4*>(row + idx);
        } else {
            v[j].x = (idx + 0 < n_valid) ? row[idx + 0] : 0.f;
            v[j].y = (idx + 1 < n_valid) ? row[idx + 1] : 0.f;
            v[j].z = (idx + 2 < n_valid) ? row[idx + 2] : 0.f;
            v[j].w = (idx + 3 < n_valid) ? row[idx + 3] : 0.f;
        }
    }
}

template <int L>
__device__ __forceinline__ void tile_store_global(float* __restrict__ row, long base, long n_valid, bool vec,
                                                  const float4 (&v)[L / 4]) {
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < L / 4; ++j) {
        const long idx = base + (long)(j * 64 + lane) * 4;
        if (vec && idx + 3 < n_valid) {
            *reinterpret_cast<float4*>(row + idx) = v[j];
        } else {
            if (idx + 0 < n_valid) row[idx + 0] = v[j].x;
            if (idx + 1 < n_valid) row[idx + 1] = v[j].y;
            if (idx + 2 < n_valid) row[idx + 2] = v[j].z;
            if (idx + 3 < n_valid) row[idx + 3] = v[j].w;
        }
    }
}

// coalesced float4s -> lane chunks
template <int L>
__device__ __forceinline__ void tile_to_chunks(float* tbuf, const float4 (&v)[L / 4], float (&X)[L]) {
    constexpr int LP = L + 4;
    const int lane = lane_id();
    wave_lds_sync();  // previous readers of tbuf are done
#pragma unroll
    for (int j = 0; j < L / 4; ++j) {
        const int m = j * 256 + 4 * lane;
        *reinterpret_cast<float4*>(&tbuf[(m / L) * LP + (m % L)]) = v[j];
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < L / 4; ++i) {
        const float4 q = *reinterpret_cast<const float4*>(&tbuf[lane * LP + 4 * i]);
        X[4 * i + 0] = q.x; X[4 * i + 1] = q.y; X[4 * i + 2] = q.z; X[4 * i + 3] = q.w;
    }
}

// lane chunks -> coalesced float4s
template <int L>
__device__ __forceinline__ void chunks_to_tile(float* tbuf, const float (&X)[L], float4 (&v)[L / 4]) {
    constexpr int LP = L + 4;
    const int lane = lane_id();
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < L / 4; ++i) {
        float4 q; q.x = X[4 * i + 0]; q.y = X[4 * i + 1]; q.z = X[4 * i + 2]; q.w = X[4 * i + 3];
        *reinterpret_cast<float4*>(&tbuf[lane * LP + 4 * i]) = q;
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < L / 4; ++j) {
        const int m = j * 256 + 4 * lane;
        v[j] = *reinterpret_cast<const float4*>(&tbuf[(m / L) * LP + (m % L)]);
    }
}

// ---- intra-workgroup mailbox: one wave hands a 2-vector carry to another wave through LDS --------
// slot = {v0, v1, seq, pad}. Single writer lane, readers poll the sequence word. LDS operations of
// a wave complete in order, so value-then-seq / seq-then-value is sufficient within a workgroup.
__device__ __forceinline__ void mbox_publish(volatile float* slot, float a, float b, int seq) {
    if (lane_id() == 0) {
        slot[0] = a;
        slot[1] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        reinterpret_cast<volatile int*>(slot)[2] = seq;
    }
}
__device__ __forceinline__ void mbox_wait(volatile float* slot, int seq, float& a, float& b) {
    while (reinterpret_cast<volatile int*>(slot)[2] != seq) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    a = slot[0];
    b = slot[1];
}

}  // namespace dasp
