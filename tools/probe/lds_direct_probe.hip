// Probe: does global_load_lds_dwordx4 (__builtin_amdgcn_global_load_lds, 16 B) put lane i of a wave at LDS base + 16 i?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_direct_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint4 *src, uint4 *out)
{
    __shared__ uint4 buf[256];
    const int w = threadIdx.x >> 6;
    // each wave copies 64 x 16 B: wave-uniform LDS base, per-lane global address
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + threadIdx.x),
                                     (__attribute__((address_space(3))) void *)(buf + w * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = buf[255 - threadIdx.x];
}
int main()
{
    uint4 *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
    uint4 h[256]; for (int i = 0; i < 256; i++) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256>>>(s, o);
    hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) if (h[i].x != (unsigned)(255 - i) || h[i].w != (unsigned)(255 - i) * 4) bad++;
    printf("bad=%d first=%u %u %u %u\n", bad, h[0].x, h[0].y, h[0].z, h[0].w);
    return 0;
}
