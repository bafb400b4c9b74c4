// Device-wide exclusive prefix sum of int32 in ONE launch: a workgroup takes a contiguous span (taken in ticket order),
// sums it, publishes the sum in a 64-bit status word (flag | value, relaxed agent-scope atomics: the payload rides in the
// word, no fence), looks back over its predecessors' words for its prefix, and scans its span with that carry
// (decoupled look-back; the workgroup that finishes last clears the words for the next call on the stream).  Up to
// 8192 elements one block walks everything.  Round 3 took three launches (per-span scan, span-sum scan, carry add):
// ~24 scans x 3 per training step, most of them inside the ball-query / clustering chains.
// Used by the ball query (cell starts, list starts), the BFS output assembly and the coordinate engine.
#include <stdlib.h>
#include <map>
#include <mutex>
#include <unordered_map>

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

struct ScanState {            // one per stream, zero between calls
    unsigned long long status[SCAN_MAX_SPANS];
    int ticket, done;
};
constexpr unsigned long long SCAN_AGG = 1ull << 32, SCAN_PREFIX = 2ull << 32;

__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_kernel(const int *in, int *out /* may alias in */, int n, int span,
                                                                    int nspans, ScanState *__restrict__ st,
                                                                    int *__restrict__ total_out)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    __shared__ int s_bid, s_carry;
    if (threadIdx.x == 0) s_bid = atomicAdd(&st->ticket, 1);        // spans are taken in the order the blocks START
    __syncthreads();
    const int bid = s_bid;
    const long begin = (long)bid * span;
    const long end = min((long)n, begin + span);
    // pass A: the span's sum
    int local = 0;
    for (long i = begin + threadIdx.x; i < end; i += SCAN_THREADS) local += in[i];
    local = wave_sum(local);
    if (lane_id() == 0) s_wave[wave_id()] = local;
    __syncthreads();
    if (wave_id() == 0) {
        const int l = lane_id();
        int total = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; w++) total += s_wave[w];
        int prefix = 0;
        if (bid == 0) {
            if (l == 0)
                __hip_atomic_store(&st->status[0], SCAN_PREFIX | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (l == 0)
                __hip_atomic_store(&st->status[bid], SCAN_AGG | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // look-back, 64 predecessors per step (round 5; one thread walking back word by word paid a memory-side
            // round trip per predecessor: 35-75 us per scan inside the ball-query / clustering chains): the aggregates down
            // to the nearest span whose inclusive prefix is known are summed; every lower ticket is running or done, so the
            // wait ends
            for (int hi = bid - 1; hi >= 0;) {
                const int t = hi - l;
                const unsigned long long w = t >= 0 ? __hip_atomic_load(&st->status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                    : SCAN_PREFIX;
                const unsigned long long incl_m = __ballot((w >> 32) == (SCAN_PREFIX >> 32)), inval_m = __ballot((w >> 32) == 0ull);
                const int first = incl_m ? __ffsll((long long)incl_m) - 1 : 64;
                const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
                if (inval_m & need) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                prefix += wave_sum(l <= first ? (int)(unsigned)w : 0);
                if (first < 64) break;
                hi -= 64;
            }
            if (l == 0)
                __hip_atomic_store(&st->status[bid], SCAN_PREFIX | (unsigned)(prefix + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (l == 0) {
            s_carry = prefix;
            if (bid == nspans - 1 && total_out) *total_out = prefix + total;
        }
    }
    __syncthreads();
    int carry = s_carry;
    // pass B: the span again (L2), scanned with the carry; every element is read before the same thread overwrites it
    for (long t0 = begin; t0 < end; t0 += SCAN_TILE) {
        int v[SCAN_ITEMS];
        int loc = 0;
        const long base = t0 + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = (base + k < end) ? in[base + k] : 0;
            loc += v[k];
        }
        int tot;
        int ex = block_excl_scan(loc, &tot, s_wave) + carry;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (base + k < end) out[base + k] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    // the last block to get here leaves the state zero for the next call (all look-backs are over by then)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&st->done, 1) == nspans - 1) {
        for (int j = 0; j < nspans; j++)
            __hip_atomic_store(&st->status[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the look-back state of a stream (scans on one stream are ordered, scans on different streams run concurrently)
// Keyed by (device, stream): the default stream's handle is the same on every device, and the state lives in the memory
// of the device that was current when it was made.
ScanState *scan_state(hipStream_t stream)
{
    static std::mutex lock;
    static std::map<std::pair<int, hipStream_t>, ScanState *> states;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    auto it = states.find({dev, stream});
    if (it != states.end()) return it->second;
    ScanState *p = nullptr;
    if (hipMalloc((void **)&p, sizeof(ScanState)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(ScanState)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    states.emplace(std::make_pair(dev, stream), p);
    return p;
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
    static const bool chained = [] { const char *e = getenv("MS3D_SCAN_CHAINED"); return !e || atoi(e) != 0; }();
    if (chained && nspans > 1) {
        ScanState *st = scan_state(stream);
        if (!st) return 10003;   // MS3D_E_INTERNAL
        scan_chained_kernel<<<nspans, SCAN_THREADS, 0, stream>>>(in, out, n, span, nspans, st, total_out_dev);
        MS3D_LAUNCH_CHECK();
        return 0;
    }
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

// exported for tests / callers of the C ABI (include/minsu3d_hip.h)
extern "C" size_t ms3d_scan_i32_workspace_bytes(void) { return ms3d_scan_workspace_bytes(); }
extern "C" int ms3d_scan_i32(const int *in, int *out, int n, int *total_out_dev, void *workspace, void *stream)
{
    return ms3d_exclusive_scan_i32(in, out, n, total_out_dev, workspace, (hipStream_t)stream);
}
