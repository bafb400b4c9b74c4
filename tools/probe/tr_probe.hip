// What does ds_read_b64_tr_b16 deliver?  lds[i] = i (16-bit); lane l reads at a chosen element offset; print what it gets.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
    for (int i = 0; i < 4; i++) out[threadIdx.x * 4 + i] = v[i];
}
int main() {
    short *out; int *addr; int h[64]; short r[256];
    hipMalloc(&out, 512); hipMalloc(&addr, 256);
    for (int pat = 0; pat < 2; pat++) {
        // pattern 0: lane l -> element 4*l (each lane its own 4 consecutive elements)
        // pattern 1: row-major [row][16 ch] tile: lane (cl = l & 15, q = l >> 4) -> row 4q + (cl & 3), channels 4*(cl >> 2)..+3
        for (int l = 0; l < 64; l++) h[l] = pat == 0 ? 4 * l : ((4 * (l >> 4) + (l & 3)) * 16 + 4 * ((l & 15) >> 2));
        hipMemcpy(addr, h, 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(out, addr);
        hipMemcpy(r, out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; l++) printf("lane %2d addr %4d -> %4d %4d %4d %4d%s", l, h[l], r[4*l], r[4*l+1], r[4*l+2], r[4*l+3], (l & 1) ? "\n" : "   |   ");
    }
    return 0;
}
