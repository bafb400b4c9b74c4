// Batched radius search for gfx950.  Replaces ballquery_batch_p_cuda_ (reference
// bfs_cluster/bfs_cluster.cu:15-60), which tests every point of the scene against every other
// (O(n * n_scene)), keeps a 1000-int stack array per thread and hands out output offsets with a
// global atomicAdd (=> non-deterministic starts, host retry loop).
//
// Pipeline (all on `stream`, one 8-byte D2H at the end for the count the wrapper needs):
//   1. insert   : cell key (scene, floor(x/cell)) of every point into an open-addressing hash
//                 table in HBM; count points per cell                       [n threads]
//   2. scan     : exclusive scan of cell counts -> cell starts              [scan.hip]
//   3. scatter  : points copied as float4 (x,y,z,idx) into cell order        [n threads]
//   4. merge    : per non-empty cell, the points of its 27-cell neighbourhood sorted by point index into one
//                 contiguous "merged candidate list" (float4 x,y,z,idx): T = neighbourhood sizes [thread per slot],
//                 scan -> list starts, then one wave per cell (<= 512 candidates, bitonic network in registers) or one
//                 workgroup per cell (bitonic network in LDS, in global scratch beyond 16384 keys)
//   5. count    : one wave per query point (in cell order) sweeps the merged list of ITS cell 64 candidates at a time
//                 and ballots the hits; len = min(hits, 1000)                [n waves]
//   6. scan     : exclusive scan of len -> canonical start
//   7. fill     : the same sweep; the list is in ascending index, so the ballot-compacted hits ARE the output in the
//                 reference's order and the first 1000 of them are the 1000 lowest indices the reference keeps
//                 (bfs_cluster.cu:38-43).  No per-query sort: the points of a cell share one sorted list (round 1
//                 sorted every hit list on its own: 232 k x 512 keys per shifted-coordinate query against 5.9 M here).
// The cell edge is 1.01*radius, so the 27-cell neighbourhood is a strict superset of the ball
// even under f32 rounding of the cell coordinate; membership itself is decided by the pinned
// expression d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) < r*r, bit-identical to the oracle.
#include "common.h"
#include "scan.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr int BQ_CAP = 1000;
constexpr unsigned long long EMPTY_KEY = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ unsigned long long cell_key(int b, int cx, int cy, int cz)
{
    return ((unsigned long long)(unsigned)(b & 0xFF) << 54) | ((unsigned long long)(unsigned)((cx + 131072) & 0x3FFFF) << 36) |
           ((unsigned long long)(unsigned)((cy + 131072) & 0x3FFFF) << 18) | (unsigned long long)(unsigned)((cz + 131072) & 0x3FFFF);
}
__device__ __forceinline__ int cell_coord(float x, float inv_cell) { return (int)floorf(x * inv_cell); }

__global__ void bq_init_kernel(unsigned long long *keys, int *cell_count, int *cell_fill, int H, int *totals)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < H) {
        keys[t] = EMPTY_KEY;
        cell_count[t] = 0;
        cell_fill[t] = 0;
    }
    if (t < 8) totals[t] = 0;
}

__global__ void bq_insert_kernel(int n, float inv_cell, const float *__restrict__ xyz,
                                 const uint8_t *__restrict__ batch_idxs, unsigned long long *keys,
                                 int *cell_count, int *slot_of_point, unsigned mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = cell_key(batch_idxs[i], cell_coord(xyz[i * 3 + 0], inv_cell),
                                            cell_coord(xyz[i * 3 + 1], inv_cell), cell_coord(xyz[i * 3 + 2], inv_cell));
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) break;
        slot = (slot + 1) & mask;
    }
    slot_of_point[i] = (int)slot;
    atomicAdd(&cell_count[slot], 1);
}

__global__ void bq_scatter_kernel(int n, const float *__restrict__ xyz, const int *__restrict__ slot_of_point,
                                  const int *__restrict__ cell_start, int *cell_fill, float4 *__restrict__ cell_pts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int slot = slot_of_point[i];
    const int pos = cell_start[slot] + atomicAdd(&cell_fill[slot], 1);
    cell_pts[pos] = make_float4(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], __int_as_float(i));
}

// lanes 0..26 look up one neighbour cell each; per-wave LDS gets the exclusive prefix [28] and starts [27]
__device__ __forceinline__ int gather_cells(int b, int cx, int cy, int cz, const unsigned long long *__restrict__ keys,
                                            const int *__restrict__ cell_start, const int *__restrict__ cell_count,
                                            unsigned mask, int *s_prefix, int *s_start)
{
    const int l = lane_id();
    int cnt = 0, st = 0;
    if (l < 27) {
        const unsigned long long key = cell_key(b, cx + (l % 3) - 1, cy + ((l / 3) % 3) - 1, cz + (l / 9) - 1);
        unsigned slot = (unsigned)mix64(key) & mask;
        for (;;) {
            const unsigned long long k = keys[slot];
            if (k == key) {
                cnt = cell_count[slot];
                st = cell_start[slot];
                break;
            }
            if (k == EMPTY_KEY) break;
            slot = (slot + 1) & mask;
        }
    }
    const int incl = wave_incl_scan(cnt);
    if (l < 27) {
        s_prefix[l] = incl - cnt;
        s_start[l] = st;
    }
    const int total = __shfl(incl, 26, 64);
    if (l == 0) s_prefix[27] = total;
    __builtin_amdgcn_wave_barrier();
    return total;
}

__device__ __forceinline__ int candidate_pos(int t, const int *s_prefix, const int *s_start)
{
    // largest c in [0,27) with prefix[c] <= t   (prefix has 28 entries, prefix[27] = total)
    int lo = 0, hi = 27;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    return s_start[lo] + (t - s_prefix[lo]);
}

__device__ __forceinline__ void key_cell(unsigned long long key, int &b, int &cx, int &cy, int &cz)
{
    b = (int)(key >> 54) & 0xFF;
    cx = (int)((key >> 36) & 0x3FFFF) - 131072;
    cy = (int)((key >> 18) & 0x3FFFF) - 131072;
    cz = (int)(key & 0x3FFFF) - 131072;
}

constexpr int BQ_WAVE_SORT = 512;    // neighbourhoods up to this size are sorted by one wave in registers
constexpr int BQ_MID_SORT = 4096;    // ... up to this size by one workgroup in 16 KB of LDS,
constexpr int BQ_LDS_SORT = 16384;   // ... up to this size in 64 KB of LDS, beyond it in global memory

// T[slot] = number of points in the 27 cells around a non-empty cell (its "neighbourhood"), 0 for empty slots.
// Neighbourhoods too large for the one-wave sort are queued for the workgroup sort.
__global__ void bq_neighbourhood_kernel(int H, const unsigned long long *__restrict__ keys,
                                        const int *__restrict__ cell_count, unsigned mask, int *__restrict__ T,
                                        int *__restrict__ big_list, int n_cap, int *counters)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= H) return;
    const unsigned long long key = keys[slot];
    int total = 0;
    if (key != EMPTY_KEY) {
        int b, cx, cy, cz;
        key_cell(key, b, cx, cy, cz);
        for (int c = 0; c < 27; c++) {
            if (c == 13) {
                total += cell_count[slot];
                continue;
            }
            const unsigned long long nk = cell_key(b, cx + (c % 3) - 1, cy + ((c / 3) % 3) - 1, cz + (c / 9) - 1);
            unsigned s = (unsigned)mix64(nk) & mask;
            for (;;) {
                const unsigned long long k = keys[s];
                if (k == nk) {
                    total += cell_count[s];
                    break;
                }
                if (k == EMPTY_KEY) break;
                s = (s + 1) & mask;
            }
        }
        // three queues by size class (the LDS a workgroup needs follows the class, and with it the occupancy)
        if (total > BQ_LDS_SORT) big_list[2 * n_cap + atomicAdd(&counters[4], 1)] = slot;
        else if (total > BQ_MID_SORT) big_list[n_cap + atomicAdd(&counters[3], 1)] = slot;
        else if (total > BQ_WAVE_SORT) big_list[atomicAdd(&counters[2], 1)] = slot;
    }
    T[slot] = total;
}

// ascending bitonic sort of 512 keys, key e = 8 * lane + r
__device__ __forceinline__ void bitonic_sort_512(int (&v)[8])
{
    const int l = lane_id();
#pragma unroll
    for (int k = 2; k <= 512; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 8) {
                const int lj = j >> 3;
                const bool lower = (l & lj) == 0;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const bool asc = ((l * 8 + r) & k) == 0;
                    const int o = __shfl_xor(v[r], lj, 64);
                    v[r] = (lower == asc) ? min(v[r], o) : max(v[r], o);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    if ((r & j) != 0) continue;
                    const bool asc = ((l * 8 + r) & k) == 0;
                    const int a = v[r], b = v[r | j];
                    const int lo = min(a, b), hi = max(a, b);
                    v[r] = asc ? lo : hi;
                    v[r | j] = asc ? hi : lo;
                }
            }
        }
    }
}

__device__ __forceinline__ float4 point_of(const float *__restrict__ xyz, int i)
{
    return make_float4(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], __int_as_float(i));
}

// The merged candidate list of a cell: the points of its 27-cell neighbourhood in ASCENDING point index, as
// float4 (x, y, z, index).  Every query point of the cell sweeps this one list, and because it is already in
// index order the hits come out in the order the reference emits them (bfs_cluster.cu:33-46 walks the scene in
// index order) -- no per-query sort.  The sort is paid once per cell instead of once per point: 5.9 M keys
// instead of 232 k x 512 on the shifted-coordinate query of the benchmark.
// One wave per cell: <= 64 candidates in one register per lane, <= 512 in eight.
__global__ __launch_bounds__(256) void bq_merge_wave_kernel(int H, const float *__restrict__ xyz,
                                                            const unsigned long long *__restrict__ keys,
                                                            const int *__restrict__ cell_start,
                                                            const int *__restrict__ cell_count,
                                                            const float4 *__restrict__ cell_pts, unsigned mask,
                                                            const int *__restrict__ T, const int *__restrict__ mstart,
                                                            float4 *__restrict__ merged)
{
    __shared__ int lds[4 * (28 + 27)];
    int *s_prefix = lds + wave_id() * (28 + 27);
    int *s_start = s_prefix + 28;
    const int l = lane_id();
    const int waves = blockDim.x >> 6;
    for (int slot = blockIdx.x * waves + wave_id(); slot < H; slot += gridDim.x * waves) {
        const int total = T[slot];
        if (total == 0 || total > BQ_WAVE_SORT) continue;
        int b, cx, cy, cz;
        key_cell(keys[slot], b, cx, cy, cz);
        gather_cells(b, cx, cy, cz, keys, cell_start, cell_count, mask, s_prefix, s_start);
        float4 *out = merged + mstart[slot];
        if (total <= 64) {
            int v = 0x7fffffff;
            if (l < total) v = __float_as_int(cell_pts[candidate_pos(l, s_prefix, s_start)].w);
#pragma unroll
            for (int kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const int o = __shfl_xor(v, j, 64);
                    const bool asc = (l & kk) == 0, lower = (l & j) == 0;
                    v = (lower == asc) ? min(v, o) : max(v, o);
                }
            if (l < total) out[l] = point_of(xyz, v);
        } else {
            int v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int e = l * 8 + r;
                v[r] = e < total ? __float_as_int(cell_pts[candidate_pos(e, s_prefix, s_start)].w) : 0x7fffffff;
            }
            bitonic_sort_512(v);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int e = l * 8 + r;
                if (e < total) out[e] = point_of(xyz, v[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ascending merge network of 512 keys held 8 per lane (key e = 8 * lane + r): the half-cleaners j = 256 .. 1 of the
// all-ascending bitonic network, without leaving the registers
__device__ __forceinline__ void halfclean_512(int (&v)[8])
{
    const int l = lane_id();
#pragma unroll
    for (int j = 256; j >= 8; j >>= 1) {
        const int lj = j >> 3;
        const bool lower = (l & lj) == 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int o = __shfl_xor(v[r], lj, 64);
            v[r] = lower ? min(v[r], o) : max(v[r], o);
        }
    }
#pragma unroll
    for (int j = 4; j >= 1; j >>= 1)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if ((r & j) != 0) continue;
            const int a = v[r], b = v[r | j];
            v[r] = min(a, b);
            v[r | j] = max(a, b);
        }
}

// One workgroup per large neighbourhood (> 512 candidates).  LDS variant (MAXK keys): every wave sorts 512-key chunks
// in registers, the chunks are then merged with the all-ascending bitonic network -- only the steps that cross a
// 512-key chunk (the mirrored compare and the half-cleaners with stride >= 512) go through LDS with a barrier, the
// strides 256 .. 1 of every merge run in registers again: 6 barriers instead of 78 for 4096 keys.
// MAXK == 0: degenerate inputs (a whole scene collapsed into a few cells, > 16384 candidates): the plain network on a
// global scratch array; keys beyond `total` are +inf by convention and a comparator that reaches one is a no-op, so
// any length sorts in place without padding.
template <int MAXK>
__global__ __launch_bounds__(256) void bq_merge_block_kernel(const float *__restrict__ xyz,
                                                             const unsigned long long *__restrict__ keys,
                                                             const int *__restrict__ cell_start,
                                                             const int *__restrict__ cell_count,
                                                             const float4 *__restrict__ cell_pts, unsigned mask,
                                                             const int *__restrict__ T, const int *__restrict__ mstart,
                                                             const int *__restrict__ big_list,
                                                             const int *__restrict__ n_big, int *__restrict__ sort_scratch,
                                                             float4 *__restrict__ merged)
{
    __shared__ int s_prefix[28];
    __shared__ int s_start[27];
    __shared__ int s_keys[MAXK > 0 ? MAXK : 1];
    const int nbig = *n_big;
    const int l = lane_id(), wv = wave_id();
    for (int w = blockIdx.x; w < nbig; w += gridDim.x) {
        const int slot = big_list[w];
        const int total = T[slot];
        int b, cx, cy, cz;
        key_cell(keys[slot], b, cx, cy, cz);
        __syncthreads();  // the previous cell's keys are no longer read
        if (threadIdx.x < 64) gather_cells(b, cx, cy, cz, keys, cell_start, cell_count, mask, s_prefix, s_start);
        __syncthreads();
        int P = 1024;
        while (P < total) P <<= 1;
        float4 *out = merged + mstart[slot];
        if (MAXK > 0) {
            // candidate indices cell by cell (coalesced), then +inf up to the next multiple of 512
            for (int c = 0; c < 27; c++) {
                const int n_c = s_prefix[c + 1] - s_prefix[c], st = s_start[c], at = s_prefix[c];
                for (int e = threadIdx.x; e < n_c; e += 256) s_keys[at + e] = __float_as_int(cell_pts[st + e].w);
            }
            for (int e = total + threadIdx.x; e < P; e += 256) s_keys[e] = 0x7fffffff;
            __syncthreads();
            const int nchunk = P >> 9;
            for (int c = wv; c < nchunk; c += 4) {
                int v[8];
                const int4 lo4 = *(const int4 *)&s_keys[c * 512 + l * 8], hi4 = *(const int4 *)&s_keys[c * 512 + l * 8 + 4];
                v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
                bitonic_sort_512(v);
                *(int4 *)&s_keys[c * 512 + l * 8] = make_int4(v[0], v[1], v[2], v[3]);
                *(int4 *)&s_keys[c * 512 + l * 8 + 4] = make_int4(v[4], v[5], v[6], v[7]);
            }
            __syncthreads();
            for (int k = 1024; k <= P; k <<= 1) {
                const int hk = k >> 1, sh = __ffs(hk) - 1;
                for (int t = threadIdx.x; t < (P >> 1); t += 256) {  // mirrored compare
                    const int blk = t >> sh, in = t & (hk - 1);
                    const int lo = blk * k + in, hi = blk * k + k - 1 - in;
                    const int a = s_keys[lo], c2 = s_keys[hi];
                    if (a > c2) {
                        s_keys[lo] = c2;
                        s_keys[hi] = a;
                    }
                }
                __syncthreads();
                for (int j = k >> 2; j >= 512; j >>= 1) {
                    for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const int a = s_keys[lo], c2 = s_keys[lo | j];
                        if (a > c2) {
                            s_keys[lo] = c2;
                            s_keys[lo | j] = a;
                        }
                    }
                    __syncthreads();
                }
                for (int c = wv; c < nchunk; c += 4) {
                    int v[8];
                    const int4 lo4 = *(const int4 *)&s_keys[c * 512 + l * 8], hi4 = *(const int4 *)&s_keys[c * 512 + l * 8 + 4];
                    v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
                    halfclean_512(v);
                    if (k < P) {
                        *(int4 *)&s_keys[c * 512 + l * 8] = make_int4(v[0], v[1], v[2], v[3]);
                        *(int4 *)&s_keys[c * 512 + l * 8 + 4] = make_int4(v[4], v[5], v[6], v[7]);
                    } else {
                        // last merge: the chunk is final, write the candidates straight from the registers
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                            const int e = c * 512 + l * 8 + r;
                            if (e < total) out[e] = point_of(xyz, v[r]);
                        }
                    }
                }
                __syncthreads();
            }
        } else {
            int *kbuf = sort_scratch + mstart[slot];
            for (int c = 0; c < 27; c++) {
                const int n_c = s_prefix[c + 1] - s_prefix[c], st = s_start[c], at = s_prefix[c];
                for (int e = threadIdx.x; e < n_c; e += 256) kbuf[at + e] = __float_as_int(cell_pts[st + e].w);
            }
            __syncthreads();
            auto cmpx = [&](int lo, int hi) {
                if (hi < total) {
                    const int a = kbuf[lo], c2 = kbuf[hi];
                    if (a > c2) {
                        kbuf[lo] = c2;
                        kbuf[hi] = a;
                    }
                }
            };
            for (int k = 2; k <= P; k <<= 1) {
                const int hk = k >> 1, sh = __ffs(hk) - 1;
                for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                    const int blk = t >> sh, in = t & (hk - 1);
                    cmpx(blk * k + in, blk * k + k - 1 - in);
                }
                __syncthreads();
                for (int j = k >> 2; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // element with bit j clear
                        cmpx(lo, lo | j);
                    }
                    __syncthreads();
                }
            }
            for (int e = threadIdx.x; e < total; e += 256) out[e] = point_of(xyz, kbuf[e]);
        }
    }
}

// One wave per query point, taken in CELL order (the waves of a workgroup then sweep the same merged list back to
// back and find it in L1/L2).  The list is in ascending point index, so the ballot-compacted hits are the output.
// COUNT pass: len = min(hits, 1000); FILL pass: the first `len` hits go to idx[start ..] (positions >= thre are not
// written, bfs_cluster.cu:51-58).
template <bool FILL>
__global__ __launch_bounds__(256) void bq_sweep_kernel(int n, float radius, long thre, const float4 *__restrict__ cell_pts,
                                                       const int *__restrict__ slot_of_point, const int *__restrict__ T,
                                                       const int *__restrict__ mstart,
                                                       const float4 *__restrict__ merged, int *__restrict__ len,
                                                       const int *__restrict__ start, int *__restrict__ idx,
                                                       int *__restrict__ start_len, int *flags)
{
    const int l = lane_id();
    const int waves = blockDim.x >> 6;
    const float r2 = radius * radius;  // bfs_cluster.cu:23
    for (int sp = blockIdx.x * waves + wave_id(); sp < n; sp += gridDim.x * waves) {
        const float4 me = cell_pts[sp];
        const int i = __float_as_int(me.w);
        const float ox = me.x, oy = me.y, oz = me.z;
        const int slot = slot_of_point[i];
        const int total = T[slot];
        const float4 *__restrict__ cand = merged + mstart[slot];
        int my_len = BQ_CAP, my_start = 0;
        if (FILL) {
            my_len = len[i];
            my_start = start[i];
            if (l == 0) {
                start_len[i * 2 + 0] = my_start;
                start_len[i * 2 + 1] = my_len;
                if (my_len >= BQ_CAP && (flags[0] & 2) == 0) atomicOr(flags, 2);  // a list reached the cap: graph may be directed
            }
        }
        int nhits = 0;
        // four 64-candidate slices per trip: the four loads of a lane are issued together, then consumed in order
        for (int t0 = 0; t0 < total && nhits < my_len; t0 += 256) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = cand[min(t0 + 64 * u + l, total - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u + l;
                const float dx = ox - p[u].x, dy = oy - p[u].y, dz = oz - p[u].z;
                const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
                const bool hit = (t < total) && (d2 < r2);
                const unsigned long long m = __ballot(hit);
                if (FILL && hit) {
                    const int rank = nhits + ballot_rank(m);
                    if (rank < my_len && (long)my_start + rank < thre) idx[my_start + rank] = __float_as_int(p[u].w);
                }
                nhits += __popcll(m);
            }
        }
        if (!FILL && l == 0) len[i] = min(nhits, BQ_CAP);
    }
}

struct BqWorkspace {
    unsigned long long *keys;
    int *cell_count, *cell_start, *cell_fill, *slot_of_point, *len, *start, *total, *flags, *T, *mstart, *big_list,
        *sort_scratch;
    float4 *cell_pts, *merged;
    void *scan_ws;
    int H;
};

int table_size(int n)
{
    int H = 1024;
    while (H < 2 * n) H <<= 1;
    return H;
}

size_t carve(BqWorkspace &w, int n, void *base)
{
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *r = base ? (void *)(p + off) : nullptr;
        off += ms3d_align(bytes);
        return r;
    };
    w.H = table_size(n);
    w.keys = (unsigned long long *)take(sizeof(unsigned long long) * w.H);
    w.cell_count = (int *)take(sizeof(int) * w.H);
    w.cell_start = (int *)take(sizeof(int) * w.H);
    w.cell_fill = (int *)take(sizeof(int) * w.H);
    w.T = (int *)take(sizeof(int) * w.H);
    w.mstart = (int *)take(sizeof(int) * w.H);
    w.big_list = (int *)take(sizeof(int) * 3 * (size_t)n);
    w.slot_of_point = (int *)take(sizeof(int) * n);
    w.len = (int *)take(sizeof(int) * n);
    w.start = (int *)take(sizeof(int) * n);
    w.cell_pts = (float4 *)take(sizeof(float4) * n);
    // a point is a candidate of every non-empty cell among the 27 around its own: at most 27 n merged entries
    w.merged = (float4 *)take(sizeof(float4) * 27 * (size_t)n);
    w.sort_scratch = (int *)take(sizeof(int) * 27 * (size_t)n);
    w.total = (int *)take(sizeof(int) * 8);   // [0] nActive  [1] flags  [2..4] queued neighbourhoods per size class  [5] merged entries
    w.flags = base ? w.total + 1 : nullptr;
    w.scan_ws = take(ms3d_scan_workspace_bytes());
    return off;
}

}  // namespace

extern "C" {

size_t ms3d_ballquery_workspace_bytes(int n)
{
    BqWorkspace w;
    return carve(w, n > 0 ? n : 1, nullptr);
}

int ms3d_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz, const uint8_t *batch_idxs,
                           const int *batch_offsets, int n_scenes, int max_scene_points, int *idx, int *start_len,
                           int *n_active, int *capped_out, void *workspace, size_t workspace_bytes,
                           ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)n_scenes; (void)max_scene_points; (void)batch_offsets;
    *n_active = 0;
    if (capped_out) *capped_out = 0;
    if (n <= 0) return 0;
    if (n > (1 << 26)) return MS3D_E_UNSUPPORTED;  // 27 n merged entries are addressed with 32-bit offsets
    BqWorkspace w;
    if (carve(w, n, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    const float inv_cell = 1.0f / (radius * 1.01f);
    const unsigned mask = (unsigned)w.H - 1u;
    const long thre = (long)n * (long)meanActive;

    bq_init_kernel<<<ms3d_divup(w.H, 256), 256, 0, stream>>>(w.keys, w.cell_count, w.cell_fill, w.H, w.total);
    MS3D_LAUNCH_CHECK();
    bq_insert_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, inv_cell, xyz, batch_idxs, w.keys, w.cell_count,
                                                           w.slot_of_point, mask);
    MS3D_LAUNCH_CHECK();
    int rc = ms3d_exclusive_scan_i32(w.cell_count, w.cell_start, w.H, nullptr, w.scan_ws, stream);
    if (rc) return rc;
    bq_scatter_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, xyz, w.slot_of_point, w.cell_start, w.cell_fill,
                                                            w.cell_pts);
    MS3D_LAUNCH_CHECK();
    // merged, index-sorted candidate list per non-empty cell
    bq_neighbourhood_kernel<<<ms3d_divup(w.H, 256), 256, 0, stream>>>(w.H, w.keys, w.cell_count, mask, w.T, w.big_list, n,
                                                                    w.total);
    MS3D_LAUNCH_CHECK();
    rc = ms3d_exclusive_scan_i32(w.T, w.mstart, w.H, w.total + 5, w.scan_ws, stream);
    if (rc) return rc;
    bq_merge_wave_kernel<<<min(ms3d_divup(w.H, 4), 256 * 32), 256, 0, stream>>>(w.H, xyz, w.keys, w.cell_start,
                                                                               w.cell_count, w.cell_pts, mask, w.T,
                                                                               w.mstart, w.merged);
    MS3D_LAUNCH_CHECK();
    bq_merge_block_kernel<BQ_MID_SORT><<<256 * 8, 256, 0, stream>>>(xyz, w.keys, w.cell_start, w.cell_count, w.cell_pts, mask,
                                                                  w.T, w.mstart, w.big_list, w.total + 2, nullptr, w.merged);
    MS3D_LAUNCH_CHECK();
    bq_merge_block_kernel<BQ_LDS_SORT><<<256 * 2, 256, 0, stream>>>(xyz, w.keys, w.cell_start, w.cell_count, w.cell_pts, mask,
                                                                  w.T, w.mstart, w.big_list + n, w.total + 3, nullptr,
                                                                  w.merged);
    MS3D_LAUNCH_CHECK();
    bq_merge_block_kernel<0><<<256 * 2, 256, 0, stream>>>(xyz, w.keys, w.cell_start, w.cell_count, w.cell_pts, mask, w.T,
                                                        w.mstart, w.big_list + 2 * (size_t)n, w.total + 4, w.sort_scratch,
                                                        w.merged);
    MS3D_LAUNCH_CHECK();
    const int grid = min(ms3d_divup(n, 4), 256 * 32);
    bq_sweep_kernel<false><<<grid, 256, 0, stream>>>(n, radius, thre, w.cell_pts, w.slot_of_point, w.T, w.mstart, w.merged,
                                                    w.len, nullptr, nullptr, nullptr, w.flags);
    MS3D_LAUNCH_CHECK();
    rc = ms3d_exclusive_scan_i32(w.len, w.start, n, w.total, w.scan_ws, stream);
    if (rc) return rc;
    bq_sweep_kernel<true><<<grid, 256, 0, stream>>>(n, radius, thre, w.cell_pts, w.slot_of_point, w.T, w.mstart, w.merged,
                                                   w.len, w.start, idx, start_len, w.flags);
    MS3D_LAUNCH_CHECK();
    int host[2] = {0, 0};
    MS3D_CHECK(hipMemcpyAsync(host, w.total, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
    MS3D_CHECK(hipStreamSynchronize(stream));
    *n_active = host[0];
    if (capped_out) *capped_out = (host[1] & 2) ? 1 : 0;
    return 0;
}

}  // extern "C"
