// Shared host/device helpers for the gfx950 kernels (wave = 64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MS3D_WAVE 64
#define MS3D_PL_ROWS 64  // output rows per tile of a pair list (ms3d_kmap_pairlist_build)
#define MS3D_PL_PARTS 256  // a pair list is cut into this many parts of near-equal batch count (one per block of a launch)
// int offset of the pick list (int4 per tile, 16-byte aligned) inside the tile_start array of a pair list
#define MS3D_PL_SCHED_OFFSET(tiles) (((tiles) + 1 + MS3D_PL_PARTS + 1 + 3) & ~3)

// Every launcher returns 0 on success or a non-zero hipError_t; nothing ever calls exit().
#define MS3D_CHECK(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return (int)_e;              \
    } while (0)
#define MS3D_LAUNCH_CHECK() MS3D_CHECK(hipGetLastError())

static inline int ms3d_divup(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t ms3d_align(size_t x) { return (x + 255) & ~(size_t)255; }

#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
// number of set bits of a wave ballot below this lane
__device__ __forceinline__ int ballot_rank(unsigned long long m)
{
    return __popcll(m & ((1ull << lane_id()) - 1ull));
}
#endif
