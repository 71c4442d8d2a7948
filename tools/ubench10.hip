// ubench10.hip -- is the numerator kernel's round limited by the LDS instruction rate at 2 waves per SIMD?
// One consumer wave (28 ds_read_b128 + 112 dependent v_add_f32 per round) and NP producer waves that each do, per
// round, STEPS x (v_perm_b32 + ds_read_b32 gather) + 1 W read, STEPS / 2 v_pk_mul_f32, STEPS x ds_write_addtid_b32,
// then s_barrier.  NP x STEPS = 112 in both shapes: 7 x 16 (8 waves per workgroup) and 14 x 8 (15 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ROUNDS 4000
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NP, int STEPS, int MODE, int CM>
__global__ __launch_bounds__((MODE & 16) ? 640 : 64 * (NP + 2 - (NP & 1))) void k(unsigned long long *cyc, float *sink, const uint32_t *codes) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 38 * 1024 / 4 * 4; i += blockDim.x) lds[i] = 1.0f + (i & 7) * 0.125f;
    __syncthreads();
    if (!(MODE & 16) && wave > NP) return;
    if ((MODE & 16) && (wave == 4 || wave == 8)) return;  // the consumer keeps its SIMD to itself (waves k and k + 4 share one)
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const float4 *ring = reinterpret_cast<const float4 *>(lds) + 256 + lane;  // 28 quads at 4 KB
        auto chain7 = [&](const float4 *v) {  // 28 dependent adds
#define Q4(p) "v"(v[p].x), "v"(v[p].y), "v"(v[p].z), "v"(v[p].w)
            asm volatile(
                "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t"
                "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0\n\t"
                "v_add_f32 %0, %9, %0\n\tv_add_f32 %0, %10, %0\n\tv_add_f32 %0, %11, %0\n\tv_add_f32 %0, %12, %0\n\t"
                "v_add_f32 %0, %13, %0\n\tv_add_f32 %0, %14, %0\n\tv_add_f32 %0, %15, %0\n\tv_add_f32 %0, %16, %0\n\t"
                "v_add_f32 %0, %17, %0\n\tv_add_f32 %0, %18, %0\n\tv_add_f32 %0, %19, %0\n\tv_add_f32 %0, %20, %0\n\t"
                "v_add_f32 %0, %21, %0\n\tv_add_f32 %0, %22, %0\n\tv_add_f32 %0, %23, %0\n\tv_add_f32 %0, %24, %0\n\t"
                "v_add_f32 %0, %25, %0\n\tv_add_f32 %0, %26, %0\n\tv_add_f32 %0, %27, %0\n\tv_add_f32 %0, %28, %0"
                : "+v"(acc) : Q4(0), Q4(1), Q4(2), Q4(3), Q4(4), Q4(5), Q4(6));
#undef Q4
        };
        if (CM == 1) {  // all 28 reads, one wait, 112 adds
            for (int r = 0; r < ROUNDS; ++r) {
                float4 v[28];
#pragma unroll
                for (int q = 0; q < 28; ++q) v[q] = ring[((r & 1) * 28 + q) * 64];
                __builtin_amdgcn_sched_barrier(0);
                chain7(v); chain7(v + 7); chain7(v + 14); chain7(v + 21);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else if (CM == 2) {  // a whole round behind: the adds never wait, 7 reads are issued in front of every 28 adds
            float4 a[28], b[28];
#pragma unroll
            for (int q = 0; q < 28; ++q) a[q] = b[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            auto half = [&](float4 *cur, float4 *nxt, int r) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
                    for (int q = 0; q < 7; ++q) nxt[blk * 7 + q] = ring[((r & 1) * 28 + blk * 7 + q) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    chain7(cur + blk * 7);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            };
            for (int r = 0; r < ROUNDS; r += 2) {
                half(a, b, r);
                half(b, a, r + 1);
            }
        } else
        for (int r = 0; r < ROUNDS; ++r) {
            float4 v[28];
            if (!(MODE & 8)) {
#pragma unroll
            for (int q = 0; q < 28; ++q) v[q] = ring[((r & 1) * 28 + q) * 64];
            }
            if (!(MODE & 9))
#pragma unroll
            for (int q = 0; q < 28; ++q) {
                asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0"
                             : "+v"(acc) : "v"(v[q].x), "v"(v[q].y), "v"(v[q].z), "v"(v[q].w));
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        const int P = (MODE & 16) ? (wave < 4 ? wave - 1 : (wave < 8 ? wave - 2 : 6)) : wave - 1;
        uint32_t cw[STEPS / 4];
        for (int i = 0; i < STEPS / 4; ++i) cw[i] = (codes[lane + 64 * i + P] & 0x1f1f1f1fu) * 4u & 0x7c7c7c7cu;
        uint32_t selv[4];
        for (int k2 = 0; k2 < 4; ++k2) asm volatile("v_mov_b32 %0, %1" : "=v"(selv[k2]) : "s"(0x03020400u + ((uint32_t)k2 << 8)));
        const uint32_t vlane = lane * 4u;
        const uint32_t ring_m0 = 4096u + (uint32_t)(P * (STEPS / 4)) * 1024u;
        for (int r = 0; r < ROUNDS; ++r) {
            float tv[STEPS];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const uint32_t addr = __builtin_amdgcn_perm(cw[s >> 2], vlane, selv[s & 3]);
                if (MODE & 2) asm volatile("v_mov_b32 %0, %1" : "=v"(tv[s]) : "v"(addr));
                else asm volatile("ds_read_b32 %0, %1 offset:61440" : "=v"(tv[s]) : "v"(addr));
            }
            float4 wq;
            asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(wq) : "v"((uint32_t)((lane & 3) * 16)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x2 x[STEPS / 2];
#pragma unroll
            for (int s = 0; s < STEPS / 2; ++s) x[s] = f32x2{tv[2 * s], tv[2 * s + 1]} * ((s & 1) ? f32x2{wq.z, wq.w} : f32x2{wq.x, wq.y});
            const uint32_t m0v = ring_m0 + (uint32_t)(r & 1) * 28672u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "m0");
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const float val = (s & 1) ? x[s >> 1].y : x[s >> 1].x;
                // quad (s / 4), piece (s % 4): offsets are compile-time
                if (MODE & 4) asm volatile("" ::"v"(val));
                else asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(val), "i"((s >> 2) * 1024 + (s & 3) * 256) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NP, int STEPS, int MODE, int CM>
void run(const char *name, unsigned long long *cyc, float *sink, const uint32_t *codes) {
    const int threads = (MODE & 16) ? 640 : 64 * (NP + 2 - (NP & 1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<NP, STEPS, MODE, CM>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    for (int rep = 0; rep < 2; ++rep) { k<NP, STEPS, MODE, CM><<<160, threads, 156 * 1024>>>(cyc, sink, codes); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(160);
    (void)hipMemcpy(h.data(), cyc, 160 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-50s %7.0f cycles per round of 112 steps = %.2f per step  (%s)\n", name, s / 160 / ROUNDS, s / 160 / ROUNDS / 112, hipGetErrorString(hipGetLastError()));
}
int main() {
    unsigned long long *cyc; float *sink; uint32_t *codes;
    (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 160 * 1024 * 4); (void)hipMalloc(&codes, 8192);
    std::vector<uint32_t> hc(2048); for (int i = 0; i < 2048; ++i) hc[i] = 0x9e3779b9u * (i + 1);
    (void)hipMemcpy(codes, hc.data(), 8192, hipMemcpyHostToDevice);
    run<7, 16, 0, 0>("7 producers x 16 steps + consumer (8 waves)", cyc, sink, codes);
    run<14, 8, 0, 0>("14 producers x 8 steps + consumer (15 waves)", cyc, sink, codes);
    run<7, 16, 1, 0>("7 x 16, consumer reads but does not add", cyc, sink, codes);
    run<7, 16, 8, 0>("7 x 16, consumer only meets the barrier", cyc, sink, codes);
    run<7, 16, 2, 0>("7 x 16, no gathers (permutes only)", cyc, sink, codes);
    run<7, 16, 4, 0>("7 x 16, no ring stores", cyc, sink, codes);
    run<7, 16, 6, 0>("7 x 16, no gathers, no stores", cyc, sink, codes);
    run<7, 16, 14, 0>("7 x 16, no gathers, no stores, idle consumer", cyc, sink, codes);
    run<7, 16, 6, 1>("consumer: 28 reads, wait, 112 adds; idle LDS", cyc, sink, codes);
    run<7, 16, 0, 1>("consumer: 28 reads, wait, 112 adds; full producers", cyc, sink, codes);
    run<7, 16, 6, 2>("consumer a round behind; idle LDS", cyc, sink, codes);
    run<7, 16, 0, 2>("consumer a round behind; full producers", cyc, sink, codes);
    run<7, 16, 16, 1>("consumer alone on its SIMD (10 waves, 2 idle); reads upfront", cyc, sink, codes);
    run<7, 16, 16, 2>("consumer alone on its SIMD; a round behind", cyc, sink, codes);
    run<7, 16, 22, 1>("consumer alone on its SIMD; idle LDS", cyc, sink, codes);
    return 0;
}
