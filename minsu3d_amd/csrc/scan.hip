// Device-wide exclusive prefix sum of int32 (three short launches: per-span scan, span-sum
// scan, carry add; one launch up to 8192 elements).  Used by the ball query (cell starts, list starts), the BFS output
// assembly and the coordinate engine.  Spans are contiguous so every load/store is coalesced.
#include "common.h"
#include "scan.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;                           // per thread per tile
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;    // 1024
constexpr int SCAN_MAX_SPANS = 1024;
constexpr int SCAN_ONE_BLOCK = 8 * SCAN_TILE;           // up to here one block scans the whole input

__device__ __forceinline__ int block_excl_scan(int v, int *total, int *s_wave)
{
    const int incl = wave_incl_scan(v);
    if (lane_id() == 63) s_wave[wave_id()] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const int t = s_wave[w];
        if (w < wave_id()) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_spans_kernel(const int *in, int *out /* may alias in */, int n, int span,
                                                                  int *__restrict__ span_sums, int *__restrict__ total_out)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    const long begin = (long)blockIdx.x * span;
    const long end = min((long)n, begin + span);
    int carry = 0;
    for (long t0 = begin; t0 < end; t0 += SCAN_TILE) {
        int v[SCAN_ITEMS];
        int local = 0;
        const long base = t0 + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = (base + k < end) ? in[base + k] : 0;
            local += v[k];
        }
        int tot;
        int ex = block_excl_scan(local, &tot, s_wave) + carry;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (base + k < end) out[base + k] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        span_sums[blockIdx.x] = carry;
        if (total_out) *total_out = carry;      // single-span launch: this block saw everything
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(int *__restrict__ span_sums, int nspans,
                                                                 int *__restrict__ total_out)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    int v[SCAN_ITEMS];
    int local = 0;
    const int base = threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < nspans) ? span_sums[base + k] : 0;
        local += v[k];
    }
    int tot;
    int ex = block_excl_scan(local, &tot, s_wave);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < nspans) span_sums[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0 && total_out) *total_out = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(int *__restrict__ out, int n, int span,
                                                                const int *__restrict__ span_sums)
{
    const long begin = (long)blockIdx.x * span;
    const long end = min((long)n, begin + span);
    const int add = span_sums[blockIdx.x];
    if (add == 0) return;
    for (long i = begin + threadIdx.x; i < end; i += SCAN_THREADS) out[i] += add;
}

}  // namespace

size_t ms3d_scan_workspace_bytes() { return ms3d_align(sizeof(int) * SCAN_MAX_SPANS); }

int ms3d_exclusive_scan_i32(const int *in, int *out, int n, int *total_out_dev, void *workspace, hipStream_t stream)
{
    int *span_sums = (int *)workspace;
    if (n <= 0) {
        if (total_out_dev) MS3D_CHECK(hipMemsetAsync(total_out_dev, 0, sizeof(int), stream));
        return 0;
    }
    if (n <= SCAN_ONE_BLOCK) {
        // short inputs (tile starts, per-cluster counts): one block walks all of it -- one launch instead of three
        scan_spans_kernel<<<1, SCAN_THREADS, 0, stream>>>(in, out, n, ms3d_divup(n, SCAN_TILE) * SCAN_TILE, span_sums, total_out_dev);
        MS3D_LAUNCH_CHECK();
        return 0;
    }
    int nspans = ms3d_divup(n, SCAN_TILE);
    if (nspans > SCAN_MAX_SPANS) nspans = SCAN_MAX_SPANS;
    int span = ms3d_divup(n, nspans);
    span = ms3d_divup(span, SCAN_TILE) * SCAN_TILE;  // whole tiles per span
    nspans = ms3d_divup(n, span);
    scan_spans_kernel<<<nspans, SCAN_THREADS, 0, stream>>>(in, out, n, span, span_sums, nullptr);
    MS3D_LAUNCH_CHECK();
    scan_sums_kernel<<<1, SCAN_THREADS, 0, stream>>>(span_sums, nspans, total_out_dev);
    MS3D_LAUNCH_CHECK();
    if (nspans > 1) {
        scan_add_kernel<<<nspans, SCAN_THREADS, 0, stream>>>(out, n, span, span_sums);
        MS3D_LAUNCH_CHECK();
    }
    return 0;
}
