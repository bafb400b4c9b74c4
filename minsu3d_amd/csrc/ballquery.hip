// Batched radius search for gfx950.  Replaces ballquery_batch_p_cuda_ (reference
// bfs_cluster/bfs_cluster.cu:15-60), which tests every point of the scene against every other
// (O(n * n_scene)), keeps a 1000-int stack array per thread and hands out output offsets with a
// global atomicAdd (=> non-deterministic starts, host retry loop).
//
// Pipeline (all on `stream`, one 4-byte D2H at the end for the count the wrapper needs):
//   1. insert   : cell key (scene, floor(x/cell)) of every point into an open-addressing hash
//                 table in HBM; count points per cell                       [n threads]
//   2. scan     : exclusive scan of cell counts -> cell starts              [scan.hip]
//   3. scatter  : points copied as float4 (x,y,z,idx) into cell order so a cell's candidates are
//                 one contiguous, coalesced 16 B/lane read                   [n threads]
//   4. count    : one wave per query point: lanes 0..26 probe the 27 neighbour cells, the wave
//                 then sweeps the concatenated candidate ranges 64 at a time and ballots the
//                 hits; len = min(hits, 1000)                                [n waves]
//   5. scan     : exclusive scan of len -> canonical start
//   6. fill     : same sweep; hits are emitted in ASCENDING point index:
//                   len <= 64 : ballot-compacted into registers, 64-lane bitonic sort
//                   len  > 64 : hits set bits in a per-wave LDS bitmap over the scene's index
//                               range; set bits are enumerated in order with popcount prefix
//                               sums and cut at the 1000th (the reference keeps the 1000
//                               lowest indices and breaks, bfs_cluster.cu:38-43)
// The cell edge is 1.01*radius, so the 27-cell neighbourhood is a strict superset of the ball
// even under f32 rounding of the cell coordinate; membership itself is decided by the pinned
// expression d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) < r*r, bit-identical to the oracle.
#include "common.h"
#include "scan.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr int BQ_CAP = 1000;
constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int DEFAULT_BITMAP_BITS = 262144;  // covers the reference's max_num_point = 250000 per scene

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

__global__ void bq_init_kernel(unsigned long long *keys, int *cell_count, int *cell_fill, int H, int *flags)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < H) {
        keys[t] = EMPTY_KEY;
        cell_count[t] = 0;
        cell_fill[t] = 0;
    }
    if (t == 0) flags[0] = 0;
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

struct WaveCells {
    int total;  // candidates in the 27 cells
};

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

constexpr int BQ_MED = 512;  // hit lists up to this length are sorted in registers

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

template <bool FILL>
__global__ __launch_bounds__(256) void bq_query_kernel(int n, float radius, float inv_cell, long thre,
                                                       const float *__restrict__ xyz,
                                                       const uint8_t *__restrict__ batch_idxs,
                                                       const int *__restrict__ batch_offsets,
                                                       const unsigned long long *__restrict__ keys,
                                                       const int *__restrict__ cell_start,
                                                       const int *__restrict__ cell_count,
                                                       const float4 *__restrict__ cell_pts, unsigned mask,
                                                       int *__restrict__ len, const int *__restrict__ start,
                                                       int *__restrict__ idx, int *__restrict__ start_len,
                                                       int bitmap_words, int *flags)
{
    extern __shared__ int lds[];
    const int per_wave = 28 + 27 + (FILL ? BQ_MED + bitmap_words : 64);
    int *s_prefix = lds + wave_id() * per_wave;
    int *s_start = s_prefix + 28;
    int *s_hits = s_start + 27;
    unsigned *s_bits = (unsigned *)(s_hits + (FILL ? BQ_MED : 64));
    const int l = lane_id();
    const int waves = blockDim.x >> 6;
    const float r2 = radius * radius;  // bfs_cluster.cu:23

    if (FILL) {
        for (int w = l; w < bitmap_words; w += 64) s_bits[w] = 0u;
        __builtin_amdgcn_wave_barrier();
    }
    // queries are taken in CELL order (the cell-sorted copy carries the original index): the waves of a workgroup and
    // of neighbouring workgroups then sweep the same 27 cells back to back and find them in L1/L2, while in input
    // order (a scan permutes its points) every query streamed its ~2000 candidates from L2
    for (int sp = blockIdx.x * waves + wave_id(); sp < n; sp += gridDim.x * waves) {
        const float4 me = cell_pts[sp];
        const int i = __float_as_int(me.w);
        const float ox = me.x, oy = me.y, oz = me.z;
        const int b = batch_idxs[i];
        const int total = gather_cells(b, cell_coord(ox, inv_cell), cell_coord(oy, inv_cell), cell_coord(oz, inv_cell),
                                       keys, cell_start, cell_count, mask, s_prefix, s_start);
        int my_len = 0, my_start = 0, sbeg = 0;
        bool small = true, medium = false;  // <= 64 hits: one bitonic pass in registers; <= 512: LDS list + 512-key sort
        if (FILL) {
            my_len = len[i];
            my_start = start[i];
            small = my_len <= 64;
            medium = !small && my_len <= BQ_MED;
            sbeg = batch_offsets[b];
            if (!small && !medium && (batch_offsets[b + 1] - sbeg) > bitmap_words * 32) {
                if (l == 0) atomicOr(flags, 1);  // scene larger than the LDS bitmap
                continue;
            }
            if (l == 0) {
                start_len[i * 2 + 0] = my_start;
                start_len[i * 2 + 1] = my_len;
                if (my_len >= BQ_CAP && (flags[0] & 2) == 0) atomicOr(flags, 2);  // a list reached the cap: graph may be directed
            }
        }
        int nhits = 0;
        int wlo = 0x7fffffff, whi = -1;  // bitmap words touched by this query (bitmap path only)
        // four 64-candidate slices per trip: the four L2 loads of a lane are issued together (one dependent load per
        // trip made this loop latency-bound: ~1000 cycles per 64 candidates), then consumed in candidate order
        // candidate t lives in the cell `cur` with prefix[cur] <= t < prefix[cur + 1]; a lane's t only grows, so it keeps
        // (cell end, position offset) in registers and walks forward instead of bisecting the prefix table in LDS for
        // every candidate (5 dependent LDS reads each -- the sweep was bound by them, not by the candidate loads)
        int cur = 0, cur_end = s_prefix[1], cur_off = s_start[0] - s_prefix[0];
        for (int t0 = 0; t0 < total; t0 += 256) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = min(t0 + 64 * u + l, total - 1);
                while (t >= cur_end) {  // empty cells are skipped here as well; at most 26 steps per query
                    cur++;
                    cur_end = s_prefix[cur + 1];
                    cur_off = s_start[cur] - s_prefix[cur];
                }
                p[u] = cell_pts[cur_off + t];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u + l;
                const float dx = ox - p[u].x, dy = oy - p[u].y, dz = oz - p[u].z;
                const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
                const bool hit = (t < total) && (d2 < r2);
                const int k = __float_as_int(p[u].w);
                const unsigned long long m = __ballot(hit);
                if (FILL && hit) {
                    if (small || medium)
                        s_hits[nhits + ballot_rank(m)] = k;
                    else {
                        const int w = (k - sbeg) >> 5;
                        atomicOr(&s_bits[w], 1u << ((k - sbeg) & 31));
                        wlo = min(wlo, w);
                        whi = max(whi, w);
                    }
                }
                nhits += __popcll(m);
            }
        }
        if (!FILL) {
            if (l == 0) len[i] = min(nhits, BQ_CAP);
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        if (small) {
            int v = (l < my_len) ? s_hits[l] : 0x7fffffff;
#pragma unroll
            for (int kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const int o = __shfl_xor(v, j, 64);
                    const bool asc = (l & kk) == 0, lower = (l & j) == 0;
                    v = (lower == asc) ? min(v, o) : max(v, o);
                }
            if (l < my_len && (long)my_start + l < thre) idx[my_start + l] = v;
        } else if (medium) {
            // 65..512 hits (the bulk of a shifted-coordinate query): 8 keys per lane, bitonic network over 512 keys.
            // The scene-wide bitmap below costs a scan of the whole scene's index range per query when the hits are
            // scattered over it (a scan permutes its points): ~2500 instructions against ~1000 here.
            int v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = (l * 8 + r < my_len) ? s_hits[l * 8 + r] : 0x7fffffff;
            bitonic_sort_512(v);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int e = l * 8 + r;
                if (e < my_len && (long)my_start + e < thre) idx[my_start + e] = v[r];
            }
        } else {
            // ordered emission of the first `my_len` set bits; lanes take interleaved words so runs of
            // consecutive indices spread over the wave
            int emitted = 0;
            wlo = wave_min(wlo);
            whi = -wave_min(-whi);
            const int w_begin = wlo & ~63, w_end = whi + 1;   // only the touched word range is scanned and cleared
            for (int w0 = w_begin; w0 < w_end && emitted < my_len; w0 += 64) {
                const int w = w0 + l;
                unsigned bits = (w < w_end) ? s_bits[w] : 0u;
                if (__ballot(bits != 0u) == 0ull) continue;
                // exclusive prefix of the per-lane popcounts by bit planes: a word rarely holds more than 1-3 hits, so two
                // ballots replace the 6-step shuffle scan + 6-step shuffle sum (the scan of the scene-wide bitmap is what
                // the fill pass of a dense query spends its time on)
                const int pc = __popc(bits);
                int rank = emitted;
                for (int plane = 0;; plane++) {
                    const unsigned long long rest = __ballot((pc >> plane) != 0);
                    if (rest == 0ull) break;
                    const unsigned long long m1 = __ballot((pc >> plane) & 1);
                    rank += ballot_rank(m1) << plane;
                    emitted += __popcll(m1) << plane;
                }
                while (bits) {
                    const int bit = __ffs(bits) - 1;
                    bits &= bits - 1;
                    if (rank < my_len && (long)my_start + rank < thre) idx[my_start + rank] = sbeg + (w << 5) + bit;
                    rank++;
                }
            }
            for (int w = w_begin + l; w < w_end; w += 64) s_bits[w] = 0u;  // leave the bitmap clean for the next query
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct BqWorkspace {
    unsigned long long *keys;
    int *cell_count, *cell_start, *cell_fill, *slot_of_point, *len, *start, *total, *flags;
    float4 *cell_pts;
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
    w.slot_of_point = (int *)take(sizeof(int) * n);
    w.len = (int *)take(sizeof(int) * n);
    w.start = (int *)take(sizeof(int) * n);
    w.cell_pts = (float4 *)take(sizeof(float4) * n);
    w.total = (int *)take(sizeof(int) * 2);
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
    (void)n_scenes;
    *n_active = 0;
    if (capped_out) *capped_out = 0;
    if (n <= 0) return 0;
    BqWorkspace w;
    if (carve(w, n, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    const float inv_cell = 1.0f / (radius * 1.01f);
    const unsigned mask = (unsigned)w.H - 1u;
    const long thre = (long)n * (long)meanActive;

    bq_init_kernel<<<ms3d_divup(w.H, 256), 256, 0, stream>>>(w.keys, w.cell_count, w.cell_fill, w.H, w.flags);
    MS3D_LAUNCH_CHECK();
    bq_insert_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, inv_cell, xyz, batch_idxs, w.keys, w.cell_count,
                                                           w.slot_of_point, mask);
    MS3D_LAUNCH_CHECK();
    int rc = ms3d_exclusive_scan_i32(w.cell_count, w.cell_start, w.H, nullptr, w.scan_ws, stream);
    if (rc) return rc;
    bq_scatter_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, xyz, w.slot_of_point, w.cell_start, w.cell_fill,
                                                            w.cell_pts);
    MS3D_LAUNCH_CHECK();

    // count pass: 4 waves per workgroup, small LDS
    {
        const int waves = 4;
        const size_t lds = sizeof(int) * (size_t)waves * (28 + 27 + 64);
        const int grid = min(ms3d_divup(n, waves), 256 * 32);
        bq_query_kernel<false><<<grid, waves * 64, lds, stream>>>(n, radius, inv_cell, thre, xyz, batch_idxs,
                                                                 batch_offsets, w.keys, w.cell_start, w.cell_count,
                                                                 w.cell_pts, mask, w.len, nullptr, nullptr, nullptr, 0,
                                                                 w.flags);
        MS3D_LAUNCH_CHECK();
    }
    rc = ms3d_exclusive_scan_i32(w.len, w.start, n, w.total, w.scan_ws, stream);
    if (rc) return rc;
    // fill pass: per-wave bitmap over the largest scene
    {
        int bits = max_scene_points > 0 ? max_scene_points : DEFAULT_BITMAP_BITS;
        if (bits > n) bits = n;
        const int words = (ms3d_divup(bits, 32) + 63) / 64 * 64;
        const size_t per_wave = sizeof(int) * (size_t)(28 + 27 + BQ_MED + words);
        if (per_wave > 150 * 1024) return MS3D_E_UNSUPPORTED;
        int waves = (int)((64 * 1024) / per_wave);
        waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
        const size_t lds = per_wave * waves;
        if (lds > 64 * 1024)
            MS3D_CHECK(hipFuncSetAttribute((const void *)bq_query_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = min(ms3d_divup(n, waves), 256 * 16);
        bq_query_kernel<true><<<grid, waves * 64, lds, stream>>>(n, radius, inv_cell, thre, xyz, batch_idxs,
                                                                batch_offsets, w.keys, w.cell_start, w.cell_count,
                                                                w.cell_pts, mask, w.len, w.start, idx, start_len, words,
                                                                w.flags);
        MS3D_LAUNCH_CHECK();
    }
    int host[2] = {0, 0};
    MS3D_CHECK(hipMemcpyAsync(host, w.total, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
    MS3D_CHECK(hipStreamSynchronize(stream));
    if (host[1] & 1) return MS3D_E_UNSUPPORTED;
    *n_active = host[0];
    if (capped_out) *capped_out = (host[1] & 2) ? 1 : 0;
    return 0;
}

}  // extern "C"
