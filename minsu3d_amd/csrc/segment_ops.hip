// Segment (per-proposal) reductions for gfx950: sec_mean / sec_min / sec_max, roipool fp/bp,
// global_avg_pool fp/bp.  Replaces the reference kernels in sec_mean/sec_mean.cu:12-86 and
// roipool/roipool.cu:12-119, which launch min(C,32) threads (3 threads for the C=3 coordinate
// case) and walk each segment serially from global memory.
//
// Design: one 64-lane wave per proposal.  Rows are pulled in 64-row tiles with one coalesced
// wave load per channel group into LDS, so HBM/L2 traffic is S*C*4 bytes read once.
//   * order-independent reductions (min / max / argmax) are done per lane over the tile and
//     combined with a (value, row) tie-break that reproduces the serial "first extremum wins";
//   * order-DEPENDENT float sums (sec_mean's divide-then-add chain, avg-pool's sum) keep the
//     reference's exact serial order: lane c walks the staged tile row by row for channel c,
//     so the result is bit-identical while the loads stay coalesced.
#include <stdint.h>
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAX_C_STAGE = 64;  // channels staged per pass

enum SegOp { OP_MEAN = 0, OP_SUM_DIV = 1 };

// sequential-order float accumulation (bit-exact with the reference's serial loops).  Lane c owns channel c's chain of
// dependent adds; the rows travel through LDS in tiles of 2048 floats, the NEXT tile's global loads are in flight while
// the current tile is added (two LDS buffers), 8 LDS reads are issued per 8 adds.
template <int OP>
__global__ __launch_bounds__(256) void seg_serial_sum_kernel(int P, int C, const float *__restrict__ inp,
                                                             const int *__restrict__ offsets,
                                                             float *__restrict__ out)
{
    constexpr int TF = 2048, PER = TF / 64;   // floats per tile, per lane
    __shared__ float tile[WAVES_PER_BLOCK][2][TF];
    const int w = wave_id(), l = lane_id();
    for (int p = blockIdx.x * WAVES_PER_BLOCK + w; p < P; p += gridDim.x * WAVES_PER_BLOCK) {
        const int s = offsets[p], e = offsets[p + 1];
        const float count = (float)(e - s);
        for (int c0 = 0; c0 < C; c0 += MAX_C_STAGE) {
            const int cw = min(MAX_C_STAGE, C - c0);
            const int rt = TF / cw;                       // rows per tile
            const int ntiles = (e - s + rt - 1) / rt;
            float reg[PER];
            auto load = [&](int ti) {
                const int r0 = s + ti * rt;
                const int nfl = min(rt, e - r0) * cw;
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    const int q = l + 64 * k;
                    const int r = q / cw, c = q - r * cw;
                    // consecutive lanes read consecutive addresses when cw == C
                    float v = q < nfl ? inp[(size_t)(r0 + r) * C + c0 + c] : 0.f;
                    if (OP == OP_MEAN) v = v / count;  // sec_mean.cu:22 divides before adding
                    reg[k] = v;
                }
            };
            auto store = [&](int ti) {
                float *dst = tile[w][ti & 1];
#pragma unroll
                for (int k = 0; k < PER; k++) dst[l + 64 * k] = reg[k];
            };
            float acc = 0.f;
            if (ntiles > 0) {
                load(0);
                store(0);
            }
            for (int ti = 0; ti < ntiles; ti++) {
                if (ti + 1 < ntiles) load(ti + 1);          // in flight during the adds below
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): this wave's LDS writes of tile ti landed
                const int rows = min(rt, e - (s + ti * rt));
                if (l < cw) {
                    const float *t = tile[w][ti & 1] + l;
                    int r = 0;
                    for (; r + 8 <= rows; r += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = t[(r + u) * cw];
#pragma unroll
                        for (int u = 0; u < 8; u++) acc += v[u];
                    }
                    for (; r < rows; r++) acc += t[r * cw];
                }
                __builtin_amdgcn_wave_barrier();
                if (ti + 1 < ntiles) store(ti + 1);
            }
            if (l < cw) {
                if (OP == OP_SUM_DIV) acc = acc / count;  // roipool.cu:78 sums then divides
                out[(size_t)p * C + c0 + l] = acc;
            }
        }
    }
}

// The same for 3-channel rows (coordinates: sec_mean in the proposal voxelisation, on the critical path between the
// grouping and the ScoreNet).  The sum is a chain of dependent adds per channel -- three lanes work, 4-5 cycles per row
// at best -- so everything else has to stay off that chain: tiles of 256 rows (contiguous memory, 12 floats per lane),
// the NEXT tile's global loads in flight while the current tile is added, double-buffered in LDS, 32 LDS reads issued
// per 32 adds.  248 -> ~120 us for the benchmark's proposals (largest ~40k points).
template <int OP>
__global__ __launch_bounds__(256) void seg_serial_sum3_kernel(int P, const float *__restrict__ inp,
                                                              const int *__restrict__ offsets, float *__restrict__ out)
{
    constexpr int RT = 256, PER = RT * 3 / 64;   // rows per tile, floats per lane and tile
    __shared__ float tile[WAVES_PER_BLOCK][2][RT * 3];
    const int w = wave_id(), l = lane_id();
    for (int p = blockIdx.x * WAVES_PER_BLOCK + w; p < P; p += gridDim.x * WAVES_PER_BLOCK) {
        const int s = offsets[p], e = offsets[p + 1];
        const float count = (float)(e - s);
        const int ntiles = (e - s + RT - 1) / RT;
        float reg[PER];
        auto load = [&](int ti) {
            const int r0 = s + ti * RT;
            const int nfl = min(RT, e - r0) * 3;
            const float *src = inp + (size_t)r0 * 3;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int q = l + 64 * k;
                float v = q < nfl ? src[q] : 0.f;
                if (OP == OP_MEAN) v = v / count;   // sec_mean.cu:22 divides before adding
                reg[k] = v;
            }
        };
        auto store = [&](int ti) {
            float *dst = tile[w][ti & 1];
#pragma unroll
            for (int k = 0; k < PER; k++) dst[l + 64 * k] = reg[k];
        };
        float acc = 0.f;
        if (ntiles > 0) {
            load(0);
            store(0);
        }
        for (int ti = 0; ti < ntiles; ti++) {
            if (ti + 1 < ntiles) load(ti + 1);          // in flight during the adds below
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): this wave's LDS writes of tile ti landed
            const int rows = min(RT, e - (s + ti * RT));
            if (l < 3) {
                const float *t = tile[w][ti & 1] + l;
                int r = 0;
                for (; r + 32 <= rows; r += 32) {
                    float v[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) v[u] = t[(r + u) * 3];
#pragma unroll
                    for (int u = 0; u < 32; u++) acc += v[u];
                }
                for (; r < rows; r++) acc += t[r * 3];
            }
            __builtin_amdgcn_wave_barrier();
            if (ti + 1 < ntiles) store(ti + 1);
        }
        if (l < 3) {
            if (OP == OP_SUM_DIV) acc = acc / count;
            out[(size_t)p * 3 + l] = acc;
        }
    }
}

// min / max (+argmax): lane-parallel over rows, then wave reduction with first-index tie-break
template <bool IS_MAX, bool WITH_ARG>
__global__ __launch_bounds__(256) void seg_extreme_kernel(int P, int C, const float *__restrict__ inp,
                                                          const int *__restrict__ offsets,
                                                          float *__restrict__ out, int *__restrict__ arg)
{
    const int w = wave_id(), l = lane_id();
    const float ident = IS_MAX ? -INFINITY : INFINITY;  // the reference's +-1e50 is +-inf in f32
    const bool pow2 = (C & (C - 1)) == 0 && C <= 64;
    for (int p = blockIdx.x * WAVES_PER_BLOCK + w; p < P; p += gridDim.x * WAVES_PER_BLOCK) {
        const int s = offsets[p], e = offsets[p + 1];
        if (pow2) {
            // feature rows (C = 16/32): lanes = (row slot, channel) so a wave load is 256 contiguous bytes
            const int c = l & (C - 1), rsub = l / C, rstep = 64 / C;
            float best = ident;
            int bi = -1;
            for (int r = s + rsub; r < e; r += rstep) {
                const float v = inp[(size_t)r * C + c];
                if (IS_MAX ? (v > best) : (v < best)) {
                    best = v;
                    bi = r;
                }
            }
            for (int d = 32; d >= C; d >>= 1) {
                const float ov = __shfl_xor(best, d, 64);
                const int oi = __shfl_xor(bi, d, 64);
                const bool better = IS_MAX ? (ov > best) : (ov < best);
                const bool tie = (ov == best) && (oi >= 0) && (bi < 0 || oi < bi);
                if (better || tie) {
                    best = ov;
                    bi = oi;
                }
            }
            if (l < C) {
                out[(size_t)p * C + c] = best;
                if (WITH_ARG) arg[(size_t)p * C + c] = bi;
            }
            continue;
        }
        for (int c = 0; c < C; c++) {
            float best = ident;
            int bi = -1;
            for (int r = s + l; r < e; r += 64) {  // ascending rows per lane: strict compare keeps the first
                const float v = inp[(size_t)r * C + c];
                if (IS_MAX ? (v > best) : (v < best)) {
                    best = v;
                    bi = r;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(best, d, 64);
                const int oi = __shfl_xor(bi, d, 64);
                const bool better = IS_MAX ? (ov > best) : (ov < best);
                // equal values: the smaller row wins; bi == -1 marks "nothing selected yet"
                const bool tie = (ov == best) && (oi >= 0) && (bi < 0 || oi < bi);
                if (better || tie) {
                    best = ov;
                    bi = oi;
                }
            }
            if (l == 0) {
                out[(size_t)p * C + c] = best;
                if (WITH_ARG) arg[(size_t)p * C + c] = bi;
            }
        }
    }
}

// Feature rows (C = 16 / 32 / 64, a power of two): one 512-thread block per proposal.  Lanes = (row slot, channel) so
// every wave load is 256 contiguous bytes; the 8 waves stride the proposal's rows, then combine through LDS with the
// same "strictly better, or equal and earlier row" rule, which reproduces the serial first-extremum semantics.
template <bool IS_MAX, bool WITH_ARG>
__global__ __launch_bounds__(512) void seg_extreme_block_kernel(int P, int C, const float *__restrict__ inp,
                                                                const int *__restrict__ offsets, float *__restrict__ out,
                                                                int *__restrict__ arg)
{
    __shared__ float s_v[8][64];
    __shared__ int s_i[8][64];
    const int w = wave_id(), l = lane_id();
    const float ident = IS_MAX ? -INFINITY : INFINITY;
    const int c = l & (C - 1), rsub = l / C, rstep = 64 / C;
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const int s = offsets[p], e = offsets[p + 1];
        float best = ident;
        int bi = -1;
        // ascending rows per lane; 8 loads in flight per lane (one dependent load per step paid a memory round trip for
        // every 8 * rstep rows of the proposal: 84 us for a 40k-point proposal)
        const int step = 8 * rstep;
        int r = s + w * rstep + rsub;
        for (; r + 7 * step < e; r += 8 * step) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = inp[(size_t)(r + u * step) * C + c];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (IS_MAX ? (v[u] > best) : (v[u] < best)) {
                    best = v[u];
                    bi = r + u * step;
                }
        }
        for (; r < e; r += step) {
            const float v = inp[(size_t)r * C + c];
            if (IS_MAX ? (v > best) : (v < best)) {
                best = v;
                bi = r;
            }
        }
        for (int d = 32; d >= C; d >>= 1) {
            const float ov = __shfl_xor(best, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            const bool better = IS_MAX ? (ov > best) : (ov < best);
            const bool tie = (ov == best) && (oi >= 0) && (bi < 0 || oi < bi);
            if (better || tie) {
                best = ov;
                bi = oi;
            }
        }
        s_v[w][l] = best;
        s_i[w][l] = bi;
        __syncthreads();
        if (w == 0 && l < C) {
            for (int k = 1; k < 8; k++) {
                const float ov = s_v[k][l];
                const int oi = s_i[k][l];
                const bool better = IS_MAX ? (ov > best) : (ov < best);
                const bool tie = (ov == best) && (oi >= 0) && (bi < 0 || oi < bi);
                if (better || tie) {
                    best = ov;
                    bi = oi;
                }
            }
            out[(size_t)p * C + c] = best;
            if (WITH_ARG) arg[(size_t)p * C + c] = bi;
        }
        __syncthreads();
    }
}

// dst[idx[i], :] += src[i, :]  (float atomics: the backward of a many-to-one row gather F[idx])
// out[i, :] = x[idx[i], :]: a lane moves 16 bytes (C % 4 == 0) -- the row gathers of the model (voxel -> point
// broadcast, engine row order in / out, proposal members) at copy speed; torch's index kernel works per element with
// 64-bit index arithmetic (54 us for 575k x 16 floats against 12 us here)
__global__ void gather_rows_kernel(const float *__restrict__ x, const long long *__restrict__ idx, long n, int C4,
                                   float *__restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C4) return;
    const long i = t / C4;
    const int c = (int)(t - i * C4);
    reinterpret_cast<float4 *>(out)[t] = reinterpret_cast<const float4 *>(x)[idx[i] * C4 + c];
}
__global__ void gather_rows_scalar_kernel(const float *__restrict__ x, const long long *__restrict__ idx, long n, int C,
                                          float *__restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    const long i = t / C;
    out[t] = x[idx[i] * C + (t - i * C)];
}

__global__ void scatter_add_rows_kernel(const float *__restrict__ src, const long long *__restrict__ idx, long n, int C,
                                        float *__restrict__ dst)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    const long i = t / C;
    const int c = (int)(t - i * C);
    atomicAdd(&dst[(size_t)idx[i] * C + c], src[t]);
}

// The same backward with a FIXED summation order (round 5: bit-reproducible training steps).  keys = the index list
// sorted ascending by a STABLE sort, order = the source row of every sorted position: the first position of a run of
// equal keys adds the run's rows up in list order (= ascending source row) and is the only writer of its destination
// row; rows no key names keep the caller's zeros.  Four gathers in flight per thread (a run is a chain of dependent
// round trips otherwise).  VEC = 4: C % 4 == 0, 16 bytes per lane.
template <int VEC>
__global__ void segment_sum_sorted_kernel(const float *__restrict__ src, const long long *__restrict__ keys,
                                          const long long *__restrict__ order, long n, int CQ, float *__restrict__ dst)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * CQ) return;
    const long p = t / CQ;
    const int c = (int)(t - p * CQ);
    const long long key = keys[p];
    if (p > 0 && keys[p - 1] == key) return;
    float acc[VEC];
#pragma unroll
    for (int u = 0; u < VEC; u++) acc[u] = 0.f;
    long q = p;
    while (q < n) {
        long long k[4], o[4];
        float v[4][VEC];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long qq = q + j < n ? q + j : n - 1;
            k[j] = q + j < n ? keys[qq] : -1;
            o[j] = order[qq];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (VEC == 4) {
                const float4 r = reinterpret_cast<const float4 *>(src)[o[j] * CQ + c];
                v[j][0] = r.x; v[j][1 % VEC] = r.y; v[j][2 % VEC] = r.z; v[j][3 % VEC] = r.w;
            } else {
                v[j][0] = src[o[j] * CQ + c];
            }
        }
        bool done = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (k[j] != key) done = true;
            if (!done) {
#pragma unroll
                for (int u = 0; u < VEC; u++) acc[u] += v[j][u];
            }
        }
        if (done) break;
        q += 4;
    }
    if (VEC == 4)
        reinterpret_cast<float4 *>(dst)[key * CQ + c] = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
    else
        dst[key * CQ + c] = acc[0];
}

__global__ void roipool_bp_kernel(int P, int C, float *__restrict__ d_feats, const int *__restrict__ maxidx,
                                  const float *__restrict__ d_out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)P * C) return;
    const int am = maxidx[t];
    if (am < 0) return;  // empty proposal: the reference would index row -1 (roipool.cu:45-46)
    const int c = (int)(t % C);
    atomicAdd(&d_feats[(size_t)am * C + c], d_out[t]);
}

// proposals own disjoint row ranges, so every d_feats element receives exactly one addend.  Element-parallel (a wave
// per proposal left the chip to the few largest proposals: 207 us): a thread owns one row x 4 channels and finds its
// proposal by bisection of the offsets (L2 resident).
__global__ __launch_bounds__(256) void avg_pool_bp_kernel(int P, int C, long S, float *__restrict__ d_feats,
                                                          const int *__restrict__ offsets,
                                                          const float *__restrict__ d_out)
{
    const int cq = (C + 3) / 4;   // 4-channel groups per row
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * cq) return;
    const long row = t / cq + offsets[0];
    if (row >= (long)offsets[P]) return;   // rows behind the last proposal receive nothing
    const int c0 = (int)(t % cq) * 4;
    int lo = 0, hi = P;           // largest p with offsets[p] <= row
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long)offsets[mid] <= row) lo = mid; else hi = mid;
    }
    const float n = (float)(offsets[lo + 1] - offsets[lo]);
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (c0 + u < C) d_feats[(size_t)row * C + c0 + u] += d_out[(size_t)lo * C + c0 + u] / n;
}

// same operator without the row count (the reference's signature): one wave per proposal
__global__ __launch_bounds__(256) void avg_pool_bp_wave_kernel(int P, int C, float *__restrict__ d_feats,
                                                               const int *__restrict__ offsets,
                                                               const float *__restrict__ d_out)
{
    const int w = wave_id(), l = lane_id();
    for (int p = blockIdx.x * WAVES_PER_BLOCK + w; p < P; p += gridDim.x * WAVES_PER_BLOCK) {
        const int s = offsets[p], e = offsets[p + 1];
        const float n = (float)(e - s);
        const long total = (long)(e - s) * C;
        for (long q = l; q < total; q += 64) {
            const int c = (int)(q % C);
            d_feats[(size_t)s * C + q] += d_out[(size_t)p * C + c] / n;
        }
    }
}

// ---- proposal voxelisation (the arithmetic of the reference's clusters_voxelization, general_model.py:152-193)
// Every float operation of the reference's torch expression chain is one correctly rounded f32 operation here, in the
// same order, so the integer voxel coordinates are identical.  hipcc contracts a * b + c into an FMA by default
// (-ffp-contract=fast), and HIP's __fmul_rn / __fadd_rn do not help (they are plain `*` / `+` defined in a header,
// i.e. under the default): the kernels switch contraction off and use the bare operators.
__global__ void pv_gather_kernel(int S, const long long *__restrict__ clusters_idx, const float *__restrict__ coords,
                                 float *__restrict__ xyz)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const long long pt = clusters_idx[(size_t)s * 2 + 1];
#pragma unroll
    for (int c = 0; c < 3; c++) xyz[(size_t)s * 3 + c] = coords[(size_t)pt * 3 + c];
}

// per proposal: min / max of the centred coordinates -> scale and shift   (params[p] = scale, shift x, y, z)
__global__ __launch_bounds__(256) void pv_params_kernel(int P, const int *__restrict__ offsets,
                                                        const float *__restrict__ xyz, const float *__restrict__ mean,
                                                        float scale_max, float ss, const float *__restrict__ rand6,
                                                        float *__restrict__ params)
{
#pragma clang fp contract(off)
    __shared__ float s_lo[4][3], s_hi[4][3];
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const int b = offsets[p], e = offsets[p + 1];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        const float m0 = mean[p * 3 + 0], m1 = mean[p * 3 + 1], m2 = mean[p * 3 + 2];
        for (int s = b + threadIdx.x; s < e; s += blockDim.x) {
            const float d0 = (xyz[(size_t)s * 3 + 0] - m0), d1 = (xyz[(size_t)s * 3 + 1] - m1),
                        d2 = (xyz[(size_t)s * 3 + 2] - m2);
            lo[0] = fminf(lo[0], d0); lo[1] = fminf(lo[1], d1); lo[2] = fminf(lo[2], d2);
            hi[0] = fmaxf(hi[0], d0); hi[1] = fmaxf(hi[1], d1); hi[2] = fmaxf(hi[2], d2);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lo[c] = wave_min_f(lo[c]);
            hi[c] = wave_max_f(hi[c]);
        }
        __syncthreads();
        if (lane_id() == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                s_lo[wave_id()][c] = lo[c];
                s_hi[wave_id()][c] = hi[c];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float l[3], h[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                l[c] = fminf(fminf(s_lo[0][c], s_lo[1][c]), fminf(s_lo[2][c], s_lo[3][c]));
                h[c] = fmaxf(fmaxf(s_hi[0][c], s_hi[1][c]), fmaxf(s_hi[2][c], s_hi[3][c]));
            }
            // c_scale = clamp(1 / max_c((hi - lo) / spatial_shape) - 0.01, max = scale)
            float r = ((h[0] - l[0]) / ss);
            r = fmaxf(r, ((h[1] - l[1]) / ss));
            r = fmaxf(r, ((h[2] - l[2]) / ss));
            float cs = ((1.f / r) - 0.01f);
            cs = cs > scale_max ? scale_max : cs;
            params[p * 4 + 0] = cs;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float ls = (l[c] * cs), hs = (h[c] * cs);
                const float room = (ss - (hs - ls));   // spatial_shape - extent
                float t1 = (room - 0.001f);
                t1 = t1 < 0.f ? 0.f : t1;                                // clamp(min = 0)
                float t2 = (room + 0.001f);
                t2 = t2 > 0.f ? 0.f : t2;                                // clamp(max = 0)
                const float sh = (-ls + (t1 * rand6[c]));
                params[p * 4 + 1 + c] = (sh + (t2 * rand6[3 + c]));
            }
        }
    }
}

__global__ void pv_emit_kernel(int S, const long long *__restrict__ clusters_idx, const float *__restrict__ xyz,
                               const float *__restrict__ mean, const float *__restrict__ params, int *__restrict__ out)
{
#pragma clang fp contract(off)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int p = (int)clusters_idx[(size_t)s * 2];
    const float cs = params[p * 4];
    int4 o;
    o.x = p;
    o.y = (int)(((xyz[(size_t)s * 3 + 0] - mean[p * 3 + 0]) * cs) + params[p * 4 + 1]);
    o.z = (int)(((xyz[(size_t)s * 3 + 1] - mean[p * 3 + 1]) * cs) + params[p * 4 + 2]);
    o.w = (int)(((xyz[(size_t)s * 3 + 2] - mean[p * 3 + 2]) * cs) + params[p * 4 + 3]);
    reinterpret_cast<int4 *>(out)[s] = o;
}

inline int grid_for(int P) { return max(1, min(ms3d_divup(P, WAVES_PER_BLOCK), 4096)); }
inline bool use_block_kernel(int C) { return (C & (C - 1)) == 0 && C >= 8 && C <= 64; }

}  // namespace

extern "C" {

int ms3d_sec_mean(int P, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    if (C == 3)
        seg_serial_sum3_kernel<OP_MEAN><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, inp, offsets, out);
    else
        seg_serial_sum_kernel<OP_MEAN><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, inp, offsets, out);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_sec_min(int P, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    if (use_block_kernel(C))
        seg_extreme_block_kernel<false, false><<<P < 4096 ? P : 4096, 512, 0, (hipStream_t)stream>>>(P, C, inp, offsets, out, nullptr);
    else
        seg_extreme_kernel<false, false><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, inp, offsets, out, nullptr);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_sec_max(int P, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    if (use_block_kernel(C))
        seg_extreme_block_kernel<true, false><<<P < 4096 ? P : 4096, 512, 0, (hipStream_t)stream>>>(P, C, inp, offsets, out, nullptr);
    else
        seg_extreme_kernel<true, false><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, inp, offsets, out, nullptr);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_roipool_fp(int P, int C, const float *feats, const int *offsets, float *out, int *maxidx,
                    ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    if (use_block_kernel(C))
        seg_extreme_block_kernel<true, true><<<P < 4096 ? P : 4096, 512, 0, (hipStream_t)stream>>>(P, C, feats, offsets, out, maxidx);
    else
        seg_extreme_kernel<true, true><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, feats, offsets, out, maxidx);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_roipool_bp(int P, int C, float *d_feats, const int *offsets, const int *maxidx, const float *d_out,
                    ms3d_stream_t stream)
{
    (void)offsets;
    if (P <= 0 || C <= 0) return 0;
    roipool_bp_kernel<<<ms3d_divup((long)P * C, 256), 256, 0, (hipStream_t)stream>>>(P, C, d_feats, maxidx, d_out);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_gather_rows(const float *x, const long long *idx, long n, int C, float *out, ms3d_stream_t stream)
{
    if (n <= 0 || C <= 0) return 0;
    if (C % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0)
        gather_rows_kernel<<<ms3d_divup(n * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, idx, n, C / 4, out);
    else
        gather_rows_scalar_kernel<<<ms3d_divup(n * C, 256), 256, 0, (hipStream_t)stream>>>(x, idx, n, C, out);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_scatter_add_rows(const float *src, const long long *idx, long n, int C, float *dst, ms3d_stream_t stream)
{
    if (n <= 0 || C <= 0) return 0;
    scatter_add_rows_kernel<<<ms3d_divup(n * C, 256), 256, 0, (hipStream_t)stream>>>(src, idx, n, C, dst);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_scatter_add_rows_sorted(const float *src, const long long *keys_sorted, const long long *order, long n, int C,
                                 float *dst, ms3d_stream_t stream)
{
    if (n <= 0 || C <= 0) return 0;
    if (C % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0)
        segment_sum_sorted_kernel<4><<<ms3d_divup(n * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(src, keys_sorted, order, n, C / 4, dst);
    else
        segment_sum_sorted_kernel<1><<<ms3d_divup(n * C, 256), 256, 0, (hipStream_t)stream>>>(src, keys_sorted, order, n, C, dst);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_global_avg_pool_fp(int P, int C, const float *feats, const int *offsets, float *out,
                            ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    seg_serial_sum_kernel<OP_SUM_DIV><<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, feats, offsets, out);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_global_avg_pool_bp(int P, int C, float *d_feats, const int *offsets, const float *d_out,
                            ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0) return 0;
    avg_pool_bp_wave_kernel<<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, C, d_feats, offsets, d_out);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_global_avg_pool_bp_rows(int P, int C, long n_rows, float *d_feats, const int *offsets, const float *d_out,
                                 ms3d_stream_t stream)
{
    if (P <= 0 || C <= 0 || n_rows <= 0) return 0;
    avg_pool_bp_kernel<<<ms3d_divup(n_rows * ((C + 3) / 4), 256), 256, 0, (hipStream_t)stream>>>(P, C, n_rows, d_feats, offsets,
                                                                                               d_out);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_proposal_voxel_coords(const long long *clusters_idx, int S, const int *offsets, int P, const float *coords,
                               float scale, int spatial_shape, const float *rand6, float *xyz_ws, float *mean_ws,
                               float *param_ws, int *out, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (S <= 0 || P <= 0) return 0;
    pv_gather_kernel<<<ms3d_divup(S, 256), 256, 0, stream>>>(S, clusters_idx, coords, xyz_ws);
    MS3D_LAUNCH_CHECK();
    seg_serial_sum3_kernel<OP_MEAN><<<grid_for(P), 256, 0, stream>>>(P, xyz_ws, offsets, mean_ws);
    MS3D_LAUNCH_CHECK();
    pv_params_kernel<<<P < 2048 ? P : 2048, 256, 0, stream>>>(P, offsets, xyz_ws, mean_ws, scale, (float)spatial_shape, rand6,
                                                             param_ws);
    MS3D_LAUNCH_CHECK();
    pv_emit_kernel<<<ms3d_divup(S, 256), 256, 0, stream>>>(S, clusters_idx, xyz_ws, mean_ws, param_ws, out);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
