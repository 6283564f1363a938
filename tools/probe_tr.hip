// Probe of ds_read_b64_tr_b16 on gfx950: which LDS halfwords does lane l receive, given per-lane byte addresses?
// LDS halfword i holds the value i.  Pattern 0: addr = 8 * lane.  Pattern 1: a [k][R=64] bf16 image, lane -> row (k0 + (lane%16)/4),
// column 16 * (lane/16 % 2) + 4 * (lane % 4), the upper half-wave k0 + 4.  Prints the 4 halfwords of every lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const unsigned *addr, unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds + addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xFFFF; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xFFFF; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned *da; unsigned short *dout;
    hipMalloc(&da, 256); hipMalloc(&dout, 512);
    for (int pat = 0; pat < 2; ++pat) {
        std::vector<unsigned> a(64);
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) a[l] = 8 * l;
            else { const int g = l / 16, i = l % 16; const int k = i / 4 + 4 * (g / 2), c = 16 * (g % 2) + 4 * (i % 4); a[l] = (k * 64 + c) * 2; }
        }
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dout);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4u -> %4u %4u %4u %4u\n", l, a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
    return 0;
}
