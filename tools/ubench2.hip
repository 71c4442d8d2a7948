// ubench2.hip -- barrier / skeleton cost probes for the producer-consumer similarity kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ROUNDS 20000
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *cyc, int rounds, const uint4 *__restrict__ src, const float *__restrict__ w, size_t stride, float *sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    float facc = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4 *p = src + blockIdx.x * 64 + (threadIdx.x & 63);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 cur = p[0];
    for (int r = 0; r < rounds; ++r) {
        if (MODE >= 1) {  // one prefetched 1-KiB vector load per wave per round, consumed a round later
            uint4 nxt = p[(size_t)((r * 8 + wave) & 1023) * stride];
            acc.x += cur.x; acc.y ^= cur.y;
            cur = nxt;
        }
        if (MODE >= 2) {  // plus a scalar 32-byte load
            const float *q = w + (size_t)((r * 8 + wave) & 4095) * 8;
            facc += q[0] + q[3] + q[7];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (float)(acc.x + acc.y) + facc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *cyc; uint4 *src; float *w, *sink;
    const size_t stride = 10048;
    hipMalloc(&cyc, 1024 * 8); hipMalloc(&src, 1024 * stride * 16); hipMalloc(&w, 4096 * 8 * 4); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(src, 1, 1024 * stride * 16); hipMemset(w, 0, 4096 * 8 * 4);
    for (int threads : {256, 512}) for (int mode = 0; mode < 3; ++mode) for (int grid : {1, 157}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<grid, threads>>>(cyc, ROUNDS, src, w, stride, sink);
            if (mode == 1) k<1><<<grid, threads>>>(cyc, ROUNDS, src, w, stride, sink);
            if (mode == 2) k<2><<<grid, threads>>>(cyc, ROUNDS, src, w, stride, sink);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("threads %d mode %d grid %3d: %.1f cycles/round\n", threads, mode, grid, s / grid / ROUNDS);
    }
    return 0;
}
