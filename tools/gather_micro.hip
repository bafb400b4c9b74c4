// Throw-away roofline probe for the graph kernels: how fast can a wave-per-list kernel stream index lists (coalesced
// 4 B/lane) and gather one 4-byte word per index from a small (L2-resident) array?  Variants: U lists slices in flight,
// optional atomicMin on a fraction of the gathered words.   build: hipcc --offload-arch=gfx950 -O3 tools/gather_micro.hip -o tools/bin/gather_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int U, int MODE>
__global__ __launch_bounds__(256) void k(int nlists, int len, const int *__restrict__ idx, int *table, int *out, int thr)
{
    const int l = threadIdx.x & 63, waves = blockDim.x >> 6;
    int acc = 0;
    for (int p = blockIdx.x * waves + (threadIdx.x >> 6); p < nlists; p += gridDim.x * waves) {
        const int *lst = idx + (long)p * len;
        for (int t0 = 0; t0 < len; t0 += 64 * U) {
            int j[U], c[U];
#pragma unroll
            for (int u = 0; u < U; u++) j[u] = (t0 + 64 * u + l < len) ? lst[t0 + 64 * u + l] : -1;
#pragma unroll
            for (int u = 0; u < U; u++) c[u] = j[u] >= 0 ? table[j[u]] : 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (MODE == 1 && j[u] >= 0 && c[u] > thr && (j[u] & 15) == 0) atomicMin(&table[j[u]], p);
                if (MODE == 2 && j[u] >= 0 && c[u] > thr && (j[u] & 15) == 0)
                    c[u] = __hip_atomic_load(&table[j[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += c[u];
            }
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}
int main()
{
    const int N = 232761, len = 196, nl = 232761;
    std::vector<int> h((size_t)nl * len);
    srand(1);
    for (auto &v : h) v = rand() % N;
    int *idx, *table, *out;
    hipMalloc(&idx, h.size() * 4); hipMalloc(&table, N * 4); hipMalloc(&out, 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(U, MODE, GRID)                                                                                  \
    {                                                                                                       \
        hipMemset(table, 0x7f, N * 4);                                                                      \
        k<U, MODE><<<GRID, 256>>>(nl, len, idx, table, out, 1000); hipDeviceSynchronize();                   \
        hipMemset(table, 0x7f, N * 4);                                                                      \
        hipEventRecord(a); k<U, MODE><<<GRID, 256>>>(nl, len, idx, table, out, 1000); hipEventRecord(b);     \
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);                                    \
        printf("U=%d mode=%d grid=%d: %.3f ms  %.1f G gathers/s\n", U, MODE, GRID, ms, (double)nl * len / ms / 1e6); \
    }
    RUN(1, 0, 2048) RUN(2, 0, 2048) RUN(4, 0, 2048) RUN(4, 0, 4096) RUN(4, 0, 8192) RUN(4, 1, 2048) RUN(4, 2, 2048) RUN(1, 1, 2048)
    return 0;
}
