// Coordinate engine for the sparse-voxel backbone (the part of MinkowskiEngine's coordinate manager the
// reference relies on: call sites model/module/backbone.py:38, common.py:69,77, general_model.py:187-191,
// data/dataset/general_dataset.py:159-163).
//
//   * 64-bit key (batch:19 | x:15 | y:15 | z:15, biased) in an open-addressing table in HBM
//   * sparse_quantize / stride-2 downsample: value = atomicMin(row) so the FIRST row of every
//     coordinate is its representative and unique rows come out in first-occurrence order
//     (flag -> exclusive scan -> rank); deterministic, no sort
//   * kernel maps as OUTPUT-STATIONARY neighbour tables, stored offset-major  nbr[k][V_out]  so that the
//     16 output rows a wave owns read 64 contiguous bytes per offset:
//       k3 s1 : nbr[k][i] = row at c_i + o_k*ts            (k = ix + 3*iy + 9*iz, x fastest)
//       k2 s2 : down[k][p] = child of coarse row p at offset k;  up[k][f] = (k == koff[f]) ? parent[f] : -1
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "scan.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ bool in_range(int v) { return v >= -16384 && v < 16384; }
__device__ __forceinline__ unsigned long long pack_key(int b, int x, int y, int z)
{
    return ((unsigned long long)(unsigned)(b & 0x7FFFF) << 45) | ((unsigned long long)(unsigned)((x + 16384) & 0x7FFF) << 30) |
           ((unsigned long long)(unsigned)((y + 16384) & 0x7FFF) << 15) | (unsigned long long)(unsigned)((z + 16384) & 0x7FFF);
}
__device__ __forceinline__ int floor_div(int v, int d) { return (v >= 0) ? (v / d) : -((-v + d - 1) / d); }

__global__ void table_clear_kernel(unsigned long long *keys, int *vals, int H, int *range_flag)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < H) {
        keys[t] = EMPTY_KEY;
        vals[t] = 0x7fffffff;
    }
    if (t == 0) *range_flag = 0;
}

// insert (optionally the stride-`q` floored coordinate) with value = min row; slot_of_row remembers the slot
// A coordinate outside [-16384, 16384) or a batch / cluster id outside [0, 524288) does not fit the key and would alias
// another voxel: range_flag is raised and the entry points that read a count back return MS3D_E_UNSUPPORTED.
__global__ void table_insert_kernel(const int *__restrict__ coords, int n, int q, unsigned long long *keys, int *vals,
                                    unsigned mask, int *__restrict__ slot_of_row, int *range_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    int x = c.y, y = c.z, z = c.w;
    if (!(in_range(x) && in_range(y) && in_range(z) && c.x >= 0 && c.x <= 0x7FFFF) && *range_flag == 0) atomicOr(range_flag, 1);
    if (q > 1) {
        x = floor_div(x, q) * q;
        y = floor_div(y, q) * q;
        z = floor_div(z, q) * q;
    }
    const unsigned long long key = pack_key(c.x, x, y, z);
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&vals[slot], i);
    if (slot_of_row) slot_of_row[i] = (int)slot;
}

__global__ void first_flag_kernel(int n, const int *__restrict__ slot_of_row, const int *__restrict__ vals,
                                  int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (vals[slot_of_row[i]] == i) ? 1 : 0;
}

// unique_idx[rank[i]] = i for representatives; inverse[i] = rank[representative(i)]
__global__ void unique_emit_kernel(int n, const int *__restrict__ slot_of_row, const int *__restrict__ vals,
                                   const int *__restrict__ rank, int *__restrict__ unique_idx,
                                   int *__restrict__ inverse)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rep = vals[slot_of_row[i]];
    if (rep == i && unique_idx) unique_idx[rank[i]] = i;
    inverse[i] = rank[rep];
}

// coarse coords + in-cell offset index for the stride-2 map
__global__ void downsample_emit_kernel(int n, int ts, const int *__restrict__ coords, const int *__restrict__ slot_of_row,
                                       const int *__restrict__ vals, const int *__restrict__ rank,
                                       int *__restrict__ out_coords, int *__restrict__ parent, int *__restrict__ koff)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const int t2 = ts * 2;
    const int qx = floor_div(c.y, t2) * t2, qy = floor_div(c.z, t2) * t2, qz = floor_div(c.w, t2) * t2;
    const int rep = vals[slot_of_row[i]];
    const int p = rank[rep];
    parent[i] = p;
    koff[i] = (c.y - qx) / ts + 2 * ((c.z - qy) / ts) + 4 * ((c.w - qz) / ts);
    if (rep == i) reinterpret_cast<int4 *>(out_coords)[p] = make_int4(c.x, qx, qy, qz);
}

__device__ __forceinline__ int table_lookup(const unsigned long long *__restrict__ keys, const int *__restrict__ vals,
                                            unsigned mask, unsigned long long key)
{
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const unsigned long long k = keys[slot];
        if (k == key) return vals[slot];
        if (k == EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
}

// one thread per (offset, row): writes nbr[k][i]; consecutive threads -> consecutive rows (coalesced stores)
__global__ void kmap_k3_kernel(const int *__restrict__ coords, int V, int ts, const unsigned long long *__restrict__ keys,
                               const int *__restrict__ vals, unsigned mask, int *__restrict__ nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= V) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const int x = c.y + (k % 3 - 1) * ts, y = c.z + ((k / 3) % 3 - 1) * ts, z = c.w + (k / 9 - 1) * ts;
    int r = -1;
    if (k == 13)
        r = i;
    else if (in_range(x) && in_range(y) && in_range(z))
        r = table_lookup(keys, vals, mask, pack_key(c.x, x, y, z));
    nbr[(size_t)k * V + i] = r;
}

// The same table from HALF the lookups (round 6): j = nbr[k][i] <=> i = nbr[26 - k][j], so offsets 0..12 are looked up and
// every hit also writes its mirror entry (each (26 - k, j) has at most one such writer; the rest of offsets 14..26 was
// pre-filled with -1).  A full-resolution level is 27 x 427k probes of which 80 % miss and walk to an empty slot: the
// lookups, not the stores, are what the kernel costs (it runs on the input pipeline's stream inside the grouping window).
__global__ void kmap_k3_sym_kernel(const int *__restrict__ coords, int V, int ts, const unsigned long long *__restrict__ keys,
                                   const int *__restrict__ vals, unsigned mask, int *__restrict__ nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;          // 0..13
    if (i >= V) return;
    if (k == 13) { nbr[(size_t)13 * V + i] = i; return; }
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const int x = c.y + (k % 3 - 1) * ts, y = c.z + ((k / 3) % 3 - 1) * ts, z = c.w + (k / 9 - 1) * ts;
    int r = -1;
    if (in_range(x) && in_range(y) && in_range(z)) r = table_lookup(keys, vals, mask, pack_key(c.x, x, y, z));
    nbr[(size_t)k * V + i] = r;
    if (r >= 0) nbr[(size_t)(26 - k) * V + r] = i;
}

__global__ void fill_minus1_kernel(int *p, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = -1;
}
__global__ void kmap_k2_kernel(const int *__restrict__ parent, const int *__restrict__ koff, int Vf, int Vc,
                               int *__restrict__ nbr_down, int *__restrict__ nbr_up)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Vf) return;
    const int p = parent[f], k = koff[f];
    nbr_down[(size_t)k * Vc + p] = f;
    nbr_up[(size_t)k * Vf + f] = p;
}

// 64-bit spatial sort key: batch index, then the 45-bit Morton code of the biased (x,y,z)
__device__ __forceinline__ unsigned long long spread3(unsigned v)
{
    unsigned long long x = v & 0x7FFFull;  // 15 bits -> every third bit
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ void morton_keys_kernel(const int *__restrict__ coords, int V, long long *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const unsigned long long m = spread3((unsigned)(c.y + 16384)) | (spread3((unsigned)(c.z + 16384)) << 1) |
                                 (spread3((unsigned)(c.w + 16384)) << 2);
    keys[i] = (long long)(((unsigned long long)(unsigned)(c.x & 0x3FFFF) << 45) | m);
}

struct CoordWs {
    unsigned long long *keys;
    int *vals, *slot_of_row, *flag, *rank, *total;
    void *scan_ws;
    int H;
};
int table_size(int n)
{
    int H = 1024;
    while (H < 2 * n) H <<= 1;
    return H;
}
size_t carve(CoordWs &w, int n, void *base)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *r = base ? (void *)((char *)base + off) : nullptr;
        off += ms3d_align(bytes);
        return r;
    };
    w.H = table_size(n);
    w.keys = (unsigned long long *)take(sizeof(unsigned long long) * w.H);
    w.vals = (int *)take(sizeof(int) * w.H);
    w.slot_of_row = (int *)take(sizeof(int) * n);
    w.flag = (int *)take(sizeof(int) * n);
    w.rank = (int *)take(sizeof(int) * n);
    w.total = (int *)take(sizeof(int) * 2);
    w.scan_ws = take(ms3d_scan_workspace_bytes());
    return off;
}

int build_table(CoordWs &w, const int *coords, int n, int q, hipStream_t stream)
{
    table_clear_kernel<<<ms3d_divup(w.H, 256), 256, 0, stream>>>(w.keys, w.vals, w.H, w.total + 1);
    MS3D_LAUNCH_CHECK();
    table_insert_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(coords, n, q, w.keys, w.vals, (unsigned)w.H - 1u,
                                                              w.slot_of_row, w.total + 1);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int rank_first(CoordWs &w, int n, int *count_host, hipStream_t stream)
{
    first_flag_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, w.slot_of_row, w.vals, w.flag);
    MS3D_LAUNCH_CHECK();
    int rc = ms3d_exclusive_scan_i32(w.flag, w.rank, n, w.total, w.scan_ws, stream);
    if (rc) return rc;
    if (count_host) {
        int h[2] = {0, 0};   // [0] number of distinct coordinates  [1] a coordinate did not fit the key (table_insert_kernel)
        MS3D_CHECK(hipMemcpyAsync(h, w.total, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        MS3D_CHECK(hipStreamSynchronize(stream));
        *count_host = h[0];
        if (h[1]) return MS3D_E_UNSUPPORTED;
    }
    return 0;
}

// ------------------------------------------------------------------ pair lists (tile-compacted kernel maps)
// One wave per tile of MS3D_PL_ROWS output rows.  Per offset the valid (input row, output row) pairs of the tile are
// compacted with a ballot and padded to a multiple of 16 (a "batch": one MFMA group in the convolution kernels).
// entry = (input row, (k << 8) | local output row); pad entries read input row 0 and carry local row MS3D_PL_ROWS.
// RPT = rows per tile: 64 (one row per lane) or 128 (lane l owns rows l and 64 + l; the pairs of the lower half come
// first inside an (offset) group, both halves in ascending row)
template <int KT, int RPT>
__global__ __launch_bounds__(256) void pairlist_count_kernel(const int *__restrict__ nbr, int K, int Vout, int tiles,
                                                             int *__restrict__ tile_nb)
{
    constexpr int H = RPT >= 64 ? RPT / 64 : 1, LIVE = RPT >= 64 ? 64 : RPT;   // RPT = 32: the upper half of the wave idles
    const int tile = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (tile > tiles) return;
    if (tile == tiles) {  // slot for the grand total of the exclusive scan
        if (lane_id() == 0) tile_nb[tiles] = 0;
        return;
    }
    int nb = 0;
    int cnt[KT];
#pragma unroll
    for (int k = 0; k < KT; k++) cnt[k] = 0;
#pragma unroll
    for (int h = 0; h < H; h++) {
        const int row = tile * RPT + 64 * h + lane_id();
        const bool ok = row < Vout && lane_id() < LIVE;
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const int v = (ok && k < K) ? nbr[(size_t)min(k, K - 1) * Vout + (ok ? row : 0)] : -1;
            cnt[k] += __popcll(__ballot(v >= 0));
        }
    }
#pragma unroll
    for (int k = 0; k < KT; k++) nb += (cnt[k] + 15) >> 4;
    if (lane_id() == 0) tile_nb[tile] = nb;
}

template <int KT, int RPT>
__global__ __launch_bounds__(256) void pairlist_fill_kernel(const int *__restrict__ nbr, int K, int Vout, int tiles,
                                                            const int *__restrict__ tile_start, int2 *__restrict__ entries)
{
    constexpr int H = RPT >= 64 ? RPT / 64 : 1, LIVE = RPT >= 64 ? 64 : RPT;
    const int tile = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (tile >= tiles) return;
    const int l = lane_id();
    int v[H][KT];   // all table entries of the tile first (one round trip), then the compaction
#pragma unroll
    for (int h = 0; h < H; h++) {
        const int row = tile * RPT + 64 * h + l;
        const bool ok = row < Vout && l < LIVE;
#pragma unroll
        for (int k = 0; k < KT; k++) v[h][k] = (ok && k < K) ? nbr[(size_t)min(k, K - 1) * Vout + (ok ? row : 0)] : -1;
    }
    size_t base = (size_t)tile_start[tile] * 16;
#pragma unroll
    for (int k = 0; k < KT; k++) {
        int n = 0;
#pragma unroll
        for (int h = 0; h < H; h++) {
            const unsigned long long m = __ballot(v[h][k] >= 0);
            if (v[h][k] >= 0) entries[base + n + ballot_rank(m)] = make_int2(v[h][k], (k << 8) | (64 * h + l));
            n += __popcll(m);
        }
        const int n16 = (n + 15) & ~15;
        if (l < n16 - n) entries[base + n + l] = make_int2(0, (k << 8) | RPT);
        base += n16;
    }
}

// Schedule of a pair list for the kernels that walk it.  A wave works through a tile as a chain of dependent round
// trips per batch group, tiles differ ~4x in batches (14..63 on 2 cm scans) and neighbouring tiles are correlated, so
// a launch lasts as long as its most loaded wave.  The tiles are therefore cut into MS3D_PL_PARTS parts of near-equal
// batch count (a tile belongs to the part its first batch falls into; one part per block), and inside a part the
// tiles are listed longest first (per run of 64 tiles), the order in which the block's waves pick them up.
__global__ __launch_bounds__(256) void pairlist_parts_kernel(const int *__restrict__ tile_start, int tiles,
                                                             int *__restrict__ part_start)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const long total = tile_start[tiles];
    auto part_of = [&](int tile) {
        const long p = total > 0 ? (long)tile_start[tile] * MS3D_PL_PARTS / total : 0;
        return (int)(p < MS3D_PL_PARTS - 1 ? p : MS3D_PL_PARTS - 1);
    };
    const int mine = part_of(t), prev = t > 0 ? part_of(t - 1) : -1;
    for (int q = prev + 1; q <= mine; q++) part_start[q] = t;
    if (t == tiles - 1)
        for (int q = mine + 1; q <= MS3D_PL_PARTS; q++) part_start[q] = tiles;
}

__global__ __launch_bounds__(64) void pairlist_order_kernel(const int *__restrict__ tile_start,
                                                            const int *__restrict__ part_start, int4 *__restrict__ order)
{
    const int l = lane_id();
    const int t0 = part_start[blockIdx.x], t1 = part_start[blockIdx.x + 1];
    for (int c = t0; c < t1; c += 64) {
        const int tile = c + l;
        const bool ok = tile < t1;
        const int cnt = ok ? tile_start[tile + 1] - tile_start[tile] : -1;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const int cj = __builtin_amdgcn_readlane(cnt, j);
            rank += (cj > cnt || (cj == cnt && j < l)) ? 1 : 0;
        }
        if (ok) order[c + rank] = make_int4(tile, tile_start[tile], tile_start[tile + 1], 0);  // all a wave needs to start the tile
    }
}

// Part cut of an offset-major list.  kt_start is an exclusive scan in (offset, tile) order, so the number of pairs in
// tiles [0, t) over all offsets is sum_k (kt_start[k * tiles + t] - kt_start[k * tiles]): no second scan.  Writes that
// per-tile prefix ([tiles + 1]) and part_start (same rule as pairlist_parts_kernel: a tile belongs to the part its
// first pair falls into).
__global__ __launch_bounds__(256) void offsetlist_parts_kernel(const int *__restrict__ kt_start, int K, int tiles,
                                                               int *__restrict__ part_start, int *__restrict__ tile_prefix)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles) return;
    const long total = kt_start[(size_t)K * tiles];
    long mine = 0, prev = 0;
    for (int k = 0; k < K; k++) {
        const int base = kt_start[(size_t)k * tiles];
        mine += kt_start[(size_t)k * tiles + t] - base;
        if (t > 0) prev += kt_start[(size_t)k * tiles + t - 1] - base;
    }
    tile_prefix[t] = (int)mine;
    if (t == tiles) return;
    auto part_of = [&](long first) {
        const long p = total > 0 ? first * MS3D_PL_PARTS / total : 0;
        return (int)(p < MS3D_PL_PARTS - 1 ? p : MS3D_PL_PARTS - 1);
    };
    const int q1 = part_of(mine), q0 = t > 0 ? part_of(prev) : -1;
    for (int q = q0 + 1; q <= q1; q++) part_start[q] = t;
    if (t == tiles - 1)
        for (int q = q1 + 1; q <= MS3D_PL_PARTS; q++) part_start[q] = tiles;
}

// Offset-major pair lists (the classic in/out index pairs per kernel offset) for the backward-weight kernel:
// kt_start[k * tiles + tile] = first pair of (offset k, 64-row tile), pairs in ascending output row inside it.
template <int KT>
__global__ __launch_bounds__(256) void offsetlist_count_kernel(const int *__restrict__ nbr, int K, int Vout, int tiles,
                                                               int *__restrict__ kt_count)
{
    const int tile = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (tile >= tiles) return;
    const int l = lane_id();
    const int row = tile * MS3D_PL_ROWS + l;
    const bool ok = row < Vout;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < KT; k++) {
        const int v = (ok && k < K) ? nbr[(size_t)min(k, K - 1) * Vout + (ok ? row : 0)] : -1;
        const int n = __popcll(__ballot(v >= 0));
        if (l == k) mine = n;
    }
    if (l < K) kt_count[(size_t)l * tiles + tile] = mine;
    if (tile == 0 && l == 0) kt_count[(size_t)K * tiles] = 0;  // slot for the grand total
}

template <int KT>
__global__ __launch_bounds__(256) void offsetlist_fill_kernel(const int *__restrict__ nbr, int K, int Vout, int tiles,
                                                              const int *__restrict__ kt_start, int2 *__restrict__ entries)
{
    const int tile = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (tile >= tiles) return;
    const int l = lane_id();
    const int row = tile * MS3D_PL_ROWS + l;
    const bool ok = row < Vout;
    int v[KT], st[KT];
#pragma unroll
    for (int k = 0; k < KT; k++) {
        v[k] = (ok && k < K) ? nbr[(size_t)min(k, K - 1) * Vout + (ok ? row : 0)] : -1;
        st[k] = kt_start[(size_t)min(k, K - 1) * tiles + tile];
    }
#pragma unroll
    for (int k = 0; k < KT; k++) {
        const unsigned long long m = __ballot(v[k] >= 0);
        if (v[k] >= 0) entries[(size_t)st[k] + ballot_rank(m)] = make_int2(v[k], row);
    }
}

}  // namespace

extern "C" {

size_t ms3d_coord_workspace_bytes(int n)
{
    CoordWs w;
    return carve(w, n > 0 ? n : 1, nullptr);
}

int ms3d_sparse_quantize(const int *coords, int n, int *unique_idx, int *inverse, int *n_unique, void *workspace,
                         size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    *n_unique = 0;
    if (n <= 0) return 0;
    CoordWs w;
    if (carve(w, n, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    int rc = build_table(w, coords, n, 1, stream);
    if (rc) return rc;
    rc = rank_first(w, n, n_unique, stream);
    if (rc) return rc;
    unique_emit_kernel<<<ms3d_divup(n, 256), 256, 0, stream>>>(n, w.slot_of_row, w.vals, w.rank, unique_idx, inverse);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_kmap_k3(const int *coords, int V, int tensor_stride, int *nbr, void *workspace, size_t workspace_bytes,
                 ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (V <= 0) return 0;
    CoordWs w;
    if (carve(w, V, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    int rc = build_table(w, coords, V, 1, stream);
    if (rc) return rc;
    static const bool sym = [] { const char *e = getenv("MS3D_KMAP_SYM"); return !e || atoi(e) != 0; }();
    if (sym) {
        MS3D_CHECK(hipMemsetAsync(nbr + (size_t)14 * V, 0xFF, sizeof(int) * (size_t)13 * V, stream));
        dim3 grid(ms3d_divup(V, 256), 14);
        kmap_k3_sym_kernel<<<grid, 256, 0, stream>>>(coords, V, tensor_stride, w.keys, w.vals, (unsigned)w.H - 1u, nbr);
    } else {
        dim3 grid(ms3d_divup(V, 256), 27);
        kmap_k3_kernel<<<grid, 256, 0, stream>>>(coords, V, tensor_stride, w.keys, w.vals, (unsigned)w.H - 1u, nbr);
    }
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_downsample(const int *coords, int V, int tensor_stride, int *out_coords, int *parent, int *koff, int *n_coarse,
                    void *workspace, size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    *n_coarse = 0;
    if (V <= 0) return 0;
    CoordWs w;
    if (carve(w, V, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    int rc = build_table(w, coords, V, tensor_stride * 2, stream);
    if (rc) return rc;
    rc = rank_first(w, V, n_coarse, stream);
    if (rc) return rc;
    downsample_emit_kernel<<<ms3d_divup(V, 256), 256, 0, stream>>>(V, tensor_stride, coords, w.slot_of_row, w.vals, w.rank,
                                                                 out_coords, parent, koff);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_morton_keys(const int *coords, int V, long long *keys, ms3d_stream_t stream)
{
    if (V <= 0) return 0;
    morton_keys_kernel<<<ms3d_divup(V, 256), 256, 0, (hipStream_t)stream>>>(coords, V, keys);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_kmap_k2(const int *parent, const int *koff, int Vf, int Vc, int *nbr_down, int *nbr_up, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (Vf <= 0) return 0;
    fill_minus1_kernel<<<ms3d_divup((long)Vc * 8, 256), 256, 0, stream>>>(nbr_down, (long)Vc * 8);
    MS3D_LAUNCH_CHECK();
    fill_minus1_kernel<<<ms3d_divup((long)Vf * 8, 256), 256, 0, stream>>>(nbr_up, (long)Vf * 8);
    MS3D_LAUNCH_CHECK();
    kmap_k2_kernel<<<ms3d_divup(Vf, 256), 256, 0, stream>>>(parent, koff, Vf, Vc, nbr_down, nbr_up);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_kmap_pairlist_tiles(int Vout) { return ms3d_divup(Vout, MS3D_PL_ROWS); }

// ints of the tile_start array: tiles + 1 batch offsets, part_start[MS3D_PL_PARTS + 1], (pad to 16 bytes,) the pick
// list int4[tiles] = (tile, first batch, end batch, 0)
int ms3d_kmap_pairlist_header_ints_rows(int Vout, int rows_per_tile)
{
    const int tiles = ms3d_divup(Vout, rows_per_tile);
    return MS3D_PL_SCHED_OFFSET(tiles) + 4 * tiles;
}
int ms3d_kmap_pairlist_header_ints(int Vout) { return ms3d_kmap_pairlist_header_ints_rows(Vout, MS3D_PL_ROWS); }

size_t ms3d_kmap_pairlist_capacity_rows(int K, int Vout, int rows_per_tile)
{
    // every (tile, offset) group pads by < 16 entries; + one group of slack (an empty last tile still reads its first slots)
    return (size_t)K * ((size_t)Vout + 15 * (size_t)ms3d_divup(Vout, rows_per_tile)) + 128;
}
size_t ms3d_kmap_pairlist_capacity(int K, int Vout) { return ms3d_kmap_pairlist_capacity_rows(K, Vout, MS3D_PL_ROWS); }

// host-side note of every list's tile size, keyed by the address of its header (see the header file)
static std::mutex g_pl_rows_lock;
static std::unordered_map<const int *, int> g_pl_rows;
int ms3d_kmap_pairlist_rows_of(const int *tile_start)
{
    if (!tile_start) return 0;
    std::lock_guard<std::mutex> guard(g_pl_rows_lock);
    auto it = g_pl_rows.find(tile_start);
    return it == g_pl_rows.end() ? MS3D_PL_ROWS : it->second;
}

int ms3d_kmap_pairlist_build_rows(const int *nbr, int K, int Vout, int rows_per_tile, int *tile_start, int *entries,
                                  void *workspace, size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (Vout <= 0) return 0;
    if (K > 27 || (rows_per_tile != 32 && rows_per_tile != 64 && rows_per_tile != 128)) return MS3D_E_UNSUPPORTED;
    {
        std::lock_guard<std::mutex> guard(g_pl_rows_lock);
        if (g_pl_rows.size() > 65536)                        // addresses of long-gone lists: entries that say the default go
            for (auto it = g_pl_rows.begin(); it != g_pl_rows.end();) it = it->second == MS3D_PL_ROWS ? g_pl_rows.erase(it) : std::next(it);
        g_pl_rows[tile_start] = rows_per_tile;
    }
    if (workspace_bytes < ms3d_scan_workspace_bytes()) return MS3D_E_WORKSPACE;
    const int tiles = ms3d_divup(Vout, rows_per_tile);
    const int grid = ms3d_divup((long)(tiles + 1) * 64, 256);
    const bool wide = rows_per_tile == 128, narrow = rows_per_tile == 32;
    if (K <= 8) {
        if (wide) pairlist_count_kernel<8, 128><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
        else if (narrow) pairlist_count_kernel<8, 32><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
        else pairlist_count_kernel<8, 64><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
    } else {
        if (wide) pairlist_count_kernel<27, 128><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
        else if (narrow) pairlist_count_kernel<27, 32><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
        else pairlist_count_kernel<27, 64><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start);
    }
    MS3D_LAUNCH_CHECK();
    int rc = ms3d_exclusive_scan_i32(tile_start, tile_start, tiles + 1, nullptr, workspace, stream);
    if (rc) return rc;
    int *part_start = tile_start + tiles + 1;
    int4 *order = reinterpret_cast<int4 *>(tile_start + MS3D_PL_SCHED_OFFSET(tiles));
    pairlist_parts_kernel<<<ms3d_divup(tiles, 256), 256, 0, stream>>>(tile_start, tiles, part_start);
    MS3D_LAUNCH_CHECK();
    pairlist_order_kernel<<<MS3D_PL_PARTS, 64, 0, stream>>>(tile_start, part_start, order);
    MS3D_LAUNCH_CHECK();
    int2 *ent = reinterpret_cast<int2 *>(entries);
    if (K <= 8) {
        if (wide) pairlist_fill_kernel<8, 128><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
        else if (narrow) pairlist_fill_kernel<8, 32><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
        else pairlist_fill_kernel<8, 64><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
    } else {
        if (wide) pairlist_fill_kernel<27, 128><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
        else if (narrow) pairlist_fill_kernel<27, 32><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
        else pairlist_fill_kernel<27, 64><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, tile_start, ent);
    }
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_kmap_pairlist_build(const int *nbr, int K, int Vout, int *tile_start, int *entries, void *workspace,
                             size_t workspace_bytes, ms3d_stream_t stream)
{
    return ms3d_kmap_pairlist_build_rows(nbr, K, Vout, MS3D_PL_ROWS, tile_start, entries, workspace, workspace_bytes, stream);
}

// ints of the kt_start array: K * tiles + 1 pair offsets, part_start[MS3D_PL_PARTS + 1], pair prefix per tile [tiles + 1]
size_t ms3d_kmap_offsetlist_header_ints(int K, int Vout)
{
    const size_t tiles = ms3d_divup(Vout > 0 ? Vout : 0, MS3D_PL_ROWS);
    return (size_t)K * tiles + 1 + MS3D_PL_PARTS + 1 + tiles + 1;
}

size_t ms3d_kmap_offsetlist_capacity(int K, int Vout) { return (size_t)K * (size_t)(Vout > 0 ? Vout : 0); }

int ms3d_kmap_offsetlist_build(const int *nbr, int K, int Vout, int *kt_start, int *entries, void *workspace,
                               size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (Vout <= 0) return 0;
    if (K > 27) return MS3D_E_UNSUPPORTED;
    if (workspace_bytes < ms3d_scan_workspace_bytes()) return MS3D_E_WORKSPACE;
    const int tiles = ms3d_divup(Vout, MS3D_PL_ROWS);
    const int grid = ms3d_divup((long)tiles * 64, 256);
    if (K <= 8)
        offsetlist_count_kernel<8><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, kt_start);
    else
        offsetlist_count_kernel<27><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, kt_start);
    MS3D_LAUNCH_CHECK();
    int rc = ms3d_exclusive_scan_i32(kt_start, kt_start, K * tiles + 1, nullptr, workspace, stream);
    if (rc) return rc;
    if (K <= 8)
        offsetlist_fill_kernel<8><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, kt_start, reinterpret_cast<int2 *>(entries));
    else
        offsetlist_fill_kernel<27><<<grid, 256, 0, stream>>>(nbr, K, Vout, tiles, kt_start, reinterpret_cast<int2 *>(entries));
    MS3D_LAUNCH_CHECK();
    // tile ranges of near-equal pair count for the workgroups of the backward-weight kernel (equal ROW ranges differ
    // ~1.4x in pairs on a scan, and the launch lasts as long as its fullest workgroup)
    int *part_start = kt_start + (size_t)K * tiles + 1, *tile_prefix = part_start + MS3D_PL_PARTS + 1;
    offsetlist_parts_kernel<<<ms3d_divup(tiles + 1, 256), 256, 0, stream>>>(kt_start, K, tiles, part_start, tile_prefix);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
