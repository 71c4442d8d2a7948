// ubench3.hip -- LDS throughput probes with the producer/consumer access patterns (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 4000
// MODE bit0: 16 conflict-free b64 gathers; bit1: 16 pk_mul; bit2: 8 b128 ring writes; bit3: consumer (8 b128 reads + 16 pk_add)
template <int MODE, int ACTIVE>
__global__ __launch_bounds__(512) void k(unsigned long long *cyc, float *sink, const unsigned *codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = i * 0.001f;
    __syncthreads();
    unsigned char *slice = smem;                                   // 21 entries x 512 B, shared
    f32x4 *ring = reinterpret_cast<f32x4 *>(smem + 16384) + wave * 8 * 64 + lane;  // 8 KB per wave
    unsigned cw[8];
    for (int i = 0; i < 8; ++i) cw[i] = codes[(wave * 8 + i) * 64 + lane];
    f32x2 acc = {0.f, 0.f};
    const f32x2 w = {1.5f, 0.5f};
    f32x2 tv[16];
    for (int s = 0; s < 16; ++s) tv[s] = f32x2{(float)s, 1.0f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < ACTIVE)
    for (int it = 0; it < ITERS; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                tv[2 * s] = *reinterpret_cast<const f32x2 *>(slice + (cw[s] & 0xFFFFu));
                tv[2 * s + 1] = *reinterpret_cast<const f32x2 *>(slice + (cw[s] >> 16));
            }
        }
        if (MODE & 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                f32x2 in = tv[s], out;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(out) : "v"(in), "v"(w));
                tv[s] = out;
            }
        }
        if (MODE & 4) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                f32x4 v = {tv[2 * p].x, tv[2 * p].y, tv[2 * p + 1].x, tv[2 * p + 1].y};
                asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(ring + p * 64)), "v"(v) : "memory");
            }
        }
        if (MODE & 8) {
            f32x4 v[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("ds_read_b128 %0, %1" : "=v"(v[p]) : "v"((unsigned)(size_t)(ring + p * 64)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                acc += f32x2{v[p].x, v[p].y};
                acc += f32x2{v[p].z, v[p].w};
            }
        }
        if ((MODE & 1) && !(MODE & 4)) {  // keep the gathers alive
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a = tv[s].x, b = tv[s].y;
                asm volatile("" ::"v"(a), "v"(b));
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + tv[3].x;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int ACTIVE>
void run(int waves, unsigned long long *cyc, float *sink, const unsigned *codes) {
    const int lds = 150000;
    hipFuncSetAttribute((const void *)k<MODE, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { k<MODE, ACTIVE><<<256, 64 * waves, lds>>>(cyc, sink, codes); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 8 + w];
    const double per = s / (256.0 * waves) / ITERS;
    printf("active %2d mode %2d waves %d: %7.1f cycles per 16-step unit per wave -> %.2f cycles/step aggregate per CU\n", ACTIVE, MODE, waves, per, per / 16.0 / waves);
}
int main() {
    unsigned long long *cyc; float *sink; unsigned *codes;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&codes, 64 * 64 * 4);
    std::vector<unsigned> hc(64 * 64);
    unsigned x = 12345;
    for (int i = 0; i < 64 * 64; ++i) {
        const int lane = i & 63;
        x = x * 1664525u + 1013904223u; unsigned a = ((x >> 20) % 21) * 512 + lane * 8;
        x = x * 1664525u + 1013904223u; unsigned b = ((x >> 20) % 21) * 512 + lane * 8;
        hc[i] = a | (b << 16);
    }
    hipMemcpy(codes, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    for (int waves : {1, 7}) {
        run<1, 64>(waves, cyc, sink, codes); run<4, 64>(waves, cyc, sink, codes); run<7, 64>(waves, cyc, sink, codes); run<8, 64>(waves, cyc, sink, codes);
        run<1, 40>(waves, cyc, sink, codes); run<4, 40>(waves, cyc, sink, codes); run<7, 40>(waves, cyc, sink, codes); run<8, 40>(waves, cyc, sink, codes);
        run<1, 32>(waves, cyc, sink, codes); run<4, 32>(waves, cyc, sink, codes); run<7, 32>(waves, cyc, sink, codes); run<8, 32>(waves, cyc, sink, codes);
    }
    return 0;
}
