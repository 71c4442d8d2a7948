// ubench_wform.hip -- the similarity kernel's W stream by ADDRESS FORM of the load (round 4).
// Every wave walks rows of an L2-resident float matrix through a list of 32-bit row offsets read by scalar loads, one
// coalesced 256-byte global_load_dword per row, 16 loads in flight, hand-issued with the kernel's own vmcnt counting --
// exactly the loop of round_loop_lds (msastat_simx.hip) without its LDS table read.  Forms:
//   0  saddr: row base in an SGPR pair by s_add_u32 / s_addc_u32, lane offset in a VGPR   (what the kernel shipped in round 3)
//   1  saddr constant (the matrix base), the row offset added to the lane offset on the VALU: v_add_u32   (1 VALU, no SALU)
//   2  vaddr64: per-lane base pair + row offset by v_add_co_u32 / v_addc_co_u32                         (2 VALU, no SALU)
//   3  vaddr64 by v_mad_u64_u32 (entry x 1 + base pair)                                                   (1 VALU, wide)
//   4  vaddr64 by v_lshl_add_u64 from a 64-bit list entry in an SGPR pair                                 (1 VALU)
// WORK = 1 adds the kernel's three VALU instructions per step (v_mul_f32, two v_pk_add_f32).
// Prints, per line: ms, TB/s, CU-cycles per wave-load at the NOMINAL clock and at the clock the chip really ran the
// kernel at (s_memtime against s_memrealtime, measured inside the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_wform tools/ubench_wform.hip && tools/ubench_wform
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) uint32_t *c32;
typedef const __attribute__((address_space(4))) uint64_t *c64;

template <int FORM, int WORK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k(const float *w, const uint32_t *list32,
                                                                                    const uint64_t *list64, int nlist, int iters,
                                                                                    float *sink, unsigned long long *clocks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t joff = (uint32_t)(lane * 4) + (uint32_t)((wave % 32) * 256);
    uint64_t wuni;
    {
        const uint64_t v = (uint64_t)w;
        wuni = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    }
    const uint64_t vbase = wuni + joff;  // per-lane base (forms 2-4)
    const uint32_t one = 1u;
    int t = (wave * 48) % (nlist - (iters + 2) * 16 - 16);
    t &= ~15;
    uint32_t eA[16], eB[16];
    uint64_t qA[16], qB[16];
    auto sload = [&](uint32_t(&e)[16], uint64_t(&q)[16], int at) {
        if constexpr (FORM == 4) {
            c64 p = (c64)(uint64_t)(list64 + at);
#pragma unroll
            for (int i = 0; i < 16; ++i) q[i] = p[i];
        } else {
            c32 p = (c32)(uint64_t)(list32 + at);
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = p[i];
        }
    };
    auto wrow = [&](float &dst, uint32_t o, uint64_t q) {
        if constexpr (FORM == 0) {
            const uint64_t row = wuni + o;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(joff), "s"(row) : "memory");
        } else if constexpr (FORM == 1) {
            uint32_t vo;
            asm volatile("v_add_u32 %1, %2, %3\n\tglobal_load_dword %0, %1, %4" : "=v"(dst), "=&v"(vo) : "s"(o), "v"(joff), "s"(wuni) : "memory");
        } else if constexpr (FORM == 2) {
            const uint64_t a = vbase + (uint64_t)o;  // (v_add_co_u32 + v_addc_co_u32 by the compiler: no 64-bit operand halves in inline asm)
            asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(a) : "memory");
        } else if constexpr (FORM == 3) {
            uint64_t a;
            asm volatile("v_mad_u64_u32 %1, vcc, %2, %3, %4\n\tglobal_load_dword %0, %1, off"
                         : "=v"(dst), "=&v"(a)
                         : "s"(o), "v"(one), "v"(vbase)
                         : "memory", "vcc");
        } else {
            uint64_t a;
            asm volatile("v_lshl_add_u64 %1, %2, 0, %3\n\tglobal_load_dword %0, %1, off" : "=v"(dst), "=&v"(a) : "s"(q), "v"(vbase) : "memory");
        }
    };
    float wv[16];
    f2 an = {1.0f, 1.0f}, ad = {1.0f, 1.0f};
    float acc = 0.f;
    const float dmul = 1.0f + 1e-7f * (float)lane;
    sload(eA, qA, t);
    sload(eB, qB, t + 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) wrow(wv[i], eA[i], qA[i]);
    auto consume_reload = [&](uint32_t(&e)[16], uint64_t(&q)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("s_waitcnt vmcnt(15)" : "+v"(wv[i])::"memory");
            const float wi = wv[i];
            if constexpr (WORK) {
                const float x = wi * dmul;
                const f2 xn = {x, x}, xd = {wi, wi};
                wrow(wv[i], e[i], q[i]);
                an += xn;
                ad += xd;
            } else {
                acc += wi;
                wrow(wv[i], e[i], q[i]);
            }
        }
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sload(eA, qA, t + 32);
        __builtin_amdgcn_sched_barrier(0);
        consume_reload(eB, qB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sload(eB, qB, t + 48);
        __builtin_amdgcn_sched_barrier(0);
        consume_reload(eA, qA);
        t += 32;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += wv[i];
    acc += an.x + an.y + ad.x + ad.y;
    if (acc == 12345.678f) sink[0] = acc;
    if (wave == 0 && lane == 0) {
        clocks[0] = __builtin_readcyclecounter() - c0;
        clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;  // 100 MHz
    }
}

template <int FORM, int WORK>
void run(const float *w, const uint32_t *l32, const uint64_t *l64, int nlist, int waves_per_simd, float *sink, unsigned long long *clk_d) {
    const int iters = 2000;
    const int grid = 256 * waves_per_simd;  // 4 waves per workgroup
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<FORM, WORK><<<grid, 256>>>(w, l32, l64, nlist, 10, sink, clk_d);
    hipEventRecord(a);
    k<FORM, WORK><<<grid, 256>>>(w, l32, l64, nlist, iters, sink, clk_d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long clk[2];
    hipMemcpy(clk, clk_d, sizeof(clk), hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // shader cycles per ns
    const double loads = (double)grid * 4 * (iters + 1) * 16;
    int nominal = 0;
    hipDeviceGetAttribute(&nominal, hipDeviceAttributeClockRate, 0);
    const char *names[] = {"saddr row base (s_add/s_addc)   ", "saddr const + v_add_u32 offset   ", "vaddr64 v_add_co/v_addc_co       ",
                           "vaddr64 v_mad_u64_u32            ", "vaddr64 v_lshl_add_u64 (64-b list)"};
    printf("form %d %s work %d waves/SIMD %d: %.3f ms  %.1f TB/s  %.2f CU-cycles per wave-load at %.2f GHz nominal, %.2f at the measured %.2f GHz\n",
           FORM, names[FORM], WORK, waves_per_simd, ms, loads * 256 / ms * 1e-9, ms * 1e-3 * nominal * 1e3 * 256 / loads,
           nominal * 1e-6, ms * 1e-3 * ghz * 1e9 * 256 / loads, ghz);
}

int main() {
    const int rows = 2048, rowbytes = 8192;
    const int nlist = 1 << 16;
    float *w, *sink;
    uint32_t *l32;
    uint64_t *l64;
    unsigned long long *clk;
    hipMalloc(&w, (size_t)rows * rowbytes + 65536);
    hipMalloc(&sink, 64);
    hipMalloc(&l32, nlist * 4);
    hipMalloc(&l64, nlist * 8);
    hipMalloc(&clk, 16);
    hipMemset(w, 0, (size_t)rows * rowbytes + 65536);
    std::vector<uint32_t> h32(nlist);
    std::vector<uint64_t> h64(nlist);
    // (rows ascending with gaps, as a compacted list of valid rows: ~72 % of the rows)
    uint32_t r = 0;
    for (int i = 0; i < nlist; ++i) {
        h32[i] = r * rowbytes;
        h64[i] = (uint64_t)r * rowbytes;
        r += 1 + ((i * 7) % 18 < 7 ? 1 : 0);
        if (r >= (uint32_t)rows) r -= rows;
    }
    hipMemcpy(l32, h32.data(), nlist * 4, hipMemcpyHostToDevice);
    hipMemcpy(l64, h64.data(), nlist * 8, hipMemcpyHostToDevice);
    for (int wps : {5, 4}) {
        run<0, 0>(w, l32, l64, nlist, wps, sink, clk);
        run<1, 0>(w, l32, l64, nlist, wps, sink, clk);
        run<2, 0>(w, l32, l64, nlist, wps, sink, clk);
        run<3, 0>(w, l32, l64, nlist, wps, sink, clk);
        run<4, 0>(w, l32, l64, nlist, wps, sink, clk);
        run<0, 1>(w, l32, l64, nlist, wps, sink, clk);
        run<1, 1>(w, l32, l64, nlist, wps, sink, clk);
        run<2, 1>(w, l32, l64, nlist, wps, sink, clk);
        run<3, 1>(w, l32, l64, nlist, wps, sink, clk);
        run<4, 1>(w, l32, l64, nlist, wps, sink, clk);
    }
    return 0;
}
