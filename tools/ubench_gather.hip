// ubench_gather.hip -- what does the W row load of the similarity kernel cost when the 64 lanes are the column's
// VALID rows (a compacted round: 64 lanes spread over 64/density consecutive floats) instead of 64 consecutive rows?
// Same instruction (global_load_dword v, v_off, s[base:base+1]), 16 rows in flight, rows 8 KB apart, L2-resident matrix.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k(const float *w, int rows, int iters, const uint32_t *offs, float *sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const uint32_t joff = offs[(size_t)(wave % 1024) * 64 + lane];
    uint32_t r = (uint32_t)((wave * 37) % rows);
    float v[16], acc = 0.f;
    auto ld = [&](uint32_t row) {
        const uint64_t base = (uint64_t)w + (uint64_t)row * 8192u;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        const uint64_t sb = ((uint64_t)hi << 32) | lo;
        float d;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(joff), "s"(sb) : "memory");
        return d;
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = ld(r);
        r = r + 1 < (uint32_t)rows ? r + 1 : 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("s_waitcnt vmcnt(15)" : "+v"(v[i])::"memory");
            acc += v[i];
            v[i] = ld(r);
            r = r + 1 < (uint32_t)rows ? r + 1 : 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i];
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int rows = 2048, iters = 1000;
    float *w, *sink;
    uint32_t *doffs;
    hipMalloc(&w, (size_t)rows * 8192);
    hipMalloc(&sink, 64);
    hipMalloc(&doffs, 1024 * 64 * 4);
    hipMemset(w, 0, (size_t)rows * 8192);
    for (double density : {1.0, 0.9, 0.72, 0.5, 0.3}) {
        std::vector<uint32_t> offs(1024 * 64);
        srand(3);
        for (int wv = 0; wv < 1024; ++wv) {
            int row = (wv * 64) % 1500;  // the round's first row (any alignment)
            if (density == 1.0) row &= ~63;
            for (int l = 0; l < 64; ++l) {
                while (density < 1.0 && rand() / (double)RAND_MAX >= density) ++row;
                offs[wv * 64 + l] = 4u * (uint32_t)(row % 2048);
                ++row;
            }
        }
        hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
        for (int wps : {4, 5}) {
            const int grid = 256 * wps;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            k<<<grid, 256>>>(w, rows, 10, doffs, sink);
            hipEventRecord(a);
            k<<<grid, 256>>>(w, rows, iters, doffs, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double loads = (double)grid * 4 * (iters + 1) * 16;
            printf("density %.2f waves/SIMD %d: %.3f ms  %.3e wave-loads/s  %.2f CU-cycles per wave-load at 2.0 GHz\n", density, wps, ms,
                   loads / (ms * 1e-3), ms * 1e-3 * 2.0e9 * 256 / loads);
        }
    }
    return 0;
}
