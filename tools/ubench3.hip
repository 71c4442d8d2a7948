// ubench3.hip -- LDS throughput probes with the producer/consumer access patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 2000
// mode 0: 16 conflict-free b64 gathers; 1: 8 b128 writes; 2: both + 16 pk_mul; 3: 8 b128 reads + 16 pk_add (consumer)
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *cyc, float *sink, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = i * 0.001f;
    __syncthreads();
    unsigned char *slice = smem + wave * 10752;          // 21 entries x 512 B
    float4 *ring = reinterpret_cast<float4 *>(smem + 8 * 10752) + wave * 8 * 64 + lane;
    unsigned x = seed * (threadIdx.x + 1) * 2654435761u;
    f32x2 acc = {0.f, 0.f};
    f32x2 w = {1.5f, 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        f32x2 tv[16];
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                x = x * 1664525u + 1013904223u;
                const unsigned e = (x >> 24) % 21u;
                tv[s] = *reinterpret_cast<const f32x2 *>(slice + e * 512 + lane * 8);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) tv[s] = f32x2{(float)it, (float)s};
        }
        if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) tv[s] = tv[s] * w;
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int p = 0; p < 8; ++p) ring[p * 64] = make_float4(tv[2 * p].x, tv[2 * p].y, tv[2 * p + 1].x, tv[2 * p + 1].y);
        }
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc += tv[s];
        }
        if (MODE == 3) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 v = ring[p * 64];
                acc += f32x2{v.x, v.y};
                acc += f32x2{v.z, v.w};
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + (float)x;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE>
void run(int waves, unsigned long long *cyc, float *sink) {
    const int lds = 150000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, 64 * waves, lds>>>(cyc, sink, 12345u); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 8 + w];
    printf("mode %d waves %d: %.1f cycles per iteration (16 steps) per wave -> %.2f cycles/step/CU aggregate\n", MODE, waves,
           s / (256.0 * waves) / ITERS, s / (256.0 * waves) / ITERS / 16.0 / waves);
}
int main() {
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    for (int waves : {1, 2, 3, 4, 6, 8}) { run<0>(waves, cyc, sink); run<1>(waves, cyc, sink); run<2>(waves, cyc, sink); run<3>(waves, cyc, sink); }
    return 0;
}
