// ubench_addtid.hip -- what does an M0-addressed LDS read cost the issuing wave?  (s_add_u32 m0 / ds_read_addtid_b32 pairs as
// in round_loop_lds, 16 per group, one lgkmcnt(0) per group; against the same reads through a VGPR address: v_add + ds_read_b32)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_addtid tools/ubench_addtid.hip && /tmp/ubench_addtid
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *cyc, float *sink, int iters, const unsigned *offs) {
    __shared__ float tab[32 * 64 * 4];
    for (int i = threadIdx.x; i < 32 * 64 * 4; i += 256) tab[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)tab + (threadIdx.x >> 6) * 8192u;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
    float acc = 0.f;
    unsigned o[16];
    for (int i = 0; i < 16; ++i) o[i] = __builtin_amdgcn_readfirstlane(offs[i]);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float d[16];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(d[i]) : "s"(o[i]), "s"(sbase) : "m0", "scc", "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const unsigned vb = base + lane * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = *(const __attribute__((address_space(3))) float *)(uintptr_t)(vb + o[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += d[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
    if (acc == 1.2345f) sink[0] = acc;
}
int main() {
    unsigned long long *cyc; float *sink; unsigned *offs;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 64); hipMalloc(&offs, 64);
    unsigned h[16]; for (int i = 0; i < 16; ++i) h[i] = (i * 7 % 21) * 256;
    hipMemcpy(offs, h, 64, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int wgs : {1, 256, 256 * 5}) {
        k<0><<<wgs, 256>>>(cyc, sink, iters, offs);
        k<1><<<wgs, 256>>>(cyc, sink, iters, offs);
        hipDeviceSynchronize();
        unsigned long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
        printf("workgroups %4d (4 waves each): addtid %.1f cycles per read and wave, vgpr-address %.1f\n", wgs, (double)hc[0] / iters / 16, (double)hc[1] / iters / 16);
    }
    return 0;
}
