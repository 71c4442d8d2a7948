// ubench4.hip -- probes for the split similarity design (gfx950):
//   (A) "den wave": exec-masked scalar-operand add chain fed by SMEM (validity ballots + W row),
//       double-buffered on lgkmcnt(0);   (B) num path: b32 gathers + pk_mul + b128 ring writes, and the
//       v_add_f32 consumer reading b128 (4 steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One row pass = NG groups of 16 steps.  VARIANT 0: masks+W from SMEM; 1: masks from SMEM, W constant VGPR;
// 2: no loads at all (register-only chain).
template <int VARIANT>
__global__ __launch_bounds__(128) void den_kernel(const unsigned *__restrict__ masks, const float *__restrict__ w,
                                                  int m, int rows, unsigned long long *cyc, float *sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned *mp = masks + (size_t)(blockIdx.x * 2 + wave) * m;
    float den = 0.f;
    const int ng2 = m / 32;  // iterations of 2 groups
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < rows; ++j) {
        const float *wp = w + (size_t)j * m;
        const unsigned vj = __builtin_amdgcn_readfirstlane(mp[j]);
        if (VARIANT == 2) {
            asm volatile(
                "s_mov_b64 s[10:11], exec\n s_mov_b32 exec_hi, 0\n s_mov_b32 s8, %1\n s_mov_b32 s9, %2\n"
                "s_mov_b32 s36, -1\n s_mov_b32 s37, 0x55555555\n s_mov_b32 s38, 0x0f0f0f0f\n s_mov_b32 s39, 0x00ff00ff\n"
                "s_mov_b32 s52, 0.5\n s_mov_b32 s53, 1.0\n s_mov_b32 s54, 2.0\n s_mov_b32 s55, 4.0\n"
                "1:\n"
                "s_and_b32 exec_lo, s8, s36\n v_add_f32 %0, s52, %0\n s_and_b32 exec_lo, s8, s37\n v_add_f32 %0, s53, %0\n"
                "s_and_b32 exec_lo, s8, s38\n v_add_f32 %0, s54, %0\n s_and_b32 exec_lo, s8, s39\n v_add_f32 %0, s55, %0\n"
                "s_and_b32 exec_lo, s8, s36\n v_add_f32 %0, s52, %0\n s_and_b32 exec_lo, s8, s37\n v_add_f32 %0, s53, %0\n"
                "s_and_b32 exec_lo, s8, s38\n v_add_f32 %0, s54, %0\n s_and_b32 exec_lo, s8, s39\n v_add_f32 %0, s55, %0\n"
                "s_and_b32 exec_lo, s8, s36\n v_add_f32 %0, s52, %0\n s_and_b32 exec_lo, s8, s37\n v_add_f32 %0, s53, %0\n"
                "s_and_b32 exec_lo, s8, s38\n v_add_f32 %0, s54, %0\n s_and_b32 exec_lo, s8, s39\n v_add_f32 %0, s55, %0\n"
                "s_and_b32 exec_lo, s8, s36\n v_add_f32 %0, s52, %0\n s_and_b32 exec_lo, s8, s37\n v_add_f32 %0, s53, %0\n"
                "s_and_b32 exec_lo, s8, s38\n v_add_f32 %0, s54, %0\n s_and_b32 exec_lo, s8, s39\n v_add_f32 %0, s55, %0\n"
                "s_sub_u32 s9, s9, 1\n s_cmp_lg_u32 s9, 0\n s_cbranch_scc1 1b\n"
                "s_mov_b64 exec, s[10:11]\n"
                : "+v"(den) : "s"(vj), "s"(ng2 * 2)
                : "s8", "s9", "s10", "s11", "s36", "s37", "s38", "s39", "s52", "s53", "s54", "s55", "scc", "memory");
            continue;
        }
#define ST(MR, WR) "s_and_b32 exec_lo, s8, s" #MR "\n v_add_f32 %0, " WR ", %0\n"
#define GROUP_A_S ST(36,"s52") ST(37,"s53") ST(38,"s54") ST(39,"s55") ST(40,"s56") ST(41,"s57") ST(42,"s58") ST(43,"s59") ST(44,"s60") ST(45,"s61") ST(46,"s62") ST(47,"s63") ST(48,"s64") ST(49,"s65") ST(50,"s66") ST(51,"s67")
#define GROUP_B_S ST(68,"s84") ST(69,"s85") ST(70,"s86") ST(71,"s87") ST(72,"s88") ST(73,"s89") ST(74,"s90") ST(75,"s91") ST(76,"s92") ST(77,"s93") ST(78,"s94") ST(79,"s95") ST(80,"s96") ST(81,"s97") ST(82,"s98") ST(83,"s99")
#define GROUP_A_V ST(36,"%5") ST(37,"%5") ST(38,"%5") ST(39,"%5") ST(40,"%5") ST(41,"%5") ST(42,"%5") ST(43,"%5") ST(44,"%5") ST(45,"%5") ST(46,"%5") ST(47,"%5") ST(48,"%5") ST(49,"%5") ST(50,"%5") ST(51,"%5")
#define GROUP_B_V ST(68,"%5") ST(69,"%5") ST(70,"%5") ST(71,"%5") ST(72,"%5") ST(73,"%5") ST(74,"%5") ST(75,"%5") ST(76,"%5") ST(77,"%5") ST(78,"%5") ST(79,"%5") ST(80,"%5") ST(81,"%5") ST(82,"%5") ST(83,"%5")
        const float wconst = 0.25f;
        if (VARIANT == 0) {
            asm volatile(
                "s_mov_b64 s[10:11], exec\n s_mov_b32 exec_hi, 0\n s_mov_b32 s8, %1\n s_mov_b32 s9, %2\n"
                "s_mov_b64 s[12:13], %3\n s_mov_b64 s[14:15], %4\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n s_load_dwordx16 s[52:67], s[14:15], 0x0\n"
                "1:\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[68:83], s[12:13], 0x40\n s_load_dwordx16 s[84:99], s[14:15], 0x40\n"
                GROUP_A_S
                "s_add_u32 s12, s12, 0x80\n s_addc_u32 s13, s13, 0\n s_add_u32 s14, s14, 0x80\n s_addc_u32 s15, s15, 0\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n s_load_dwordx16 s[52:67], s[14:15], 0x0\n"
                GROUP_B_S
                "s_sub_u32 s9, s9, 1\n s_cmp_lg_u32 s9, 0\n s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_mov_b64 exec, s[10:11]\n"
                : "+v"(den) : "s"(vj), "s"(ng2), "s"(mp), "s"(wp), "v"(wconst)
                : "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15",
                  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
                  "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
                  "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
                  "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
                  "scc", "memory");
        } else {
            asm volatile(
                "s_mov_b64 s[10:11], exec\n s_mov_b32 exec_hi, 0\n s_mov_b32 s8, %1\n s_mov_b32 s9, %2\n"
                "s_mov_b64 s[12:13], %3\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n"
                "1:\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[68:83], s[12:13], 0x40\n"
                GROUP_A_V
                "s_add_u32 s12, s12, 0x80\n s_addc_u32 s13, s13, 0\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n"
                GROUP_B_V
                "s_sub_u32 s9, s9, 1\n s_cmp_lg_u32 s9, 0\n s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_mov_b64 exec, s[10:11]\n"
                : "+v"(den) : "s"(vj), "s"(ng2), "s"(mp), "s"(wp), "v"(wconst)
                : "s8", "s9", "s10", "s11", "s12", "s13",
                  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
                  "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
                  "scc", "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = den;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 2 + wave] = t1 - t0;
}

#define ITERS 4000
// MODE bit0: 16 conflict-free b32 gathers; bit1: 8 pk_mul; bit2: 4 b128 ring writes; bit3: consumer (4 b128 reads + 16 v_add chain)
template <int MODE>
__global__ __launch_bounds__(512) void num_kernel(unsigned long long *cyc, float *sink, const unsigned *codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 30000; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = i * 0.001f;
    __syncthreads();
    unsigned char *slice = smem;  // 29 entries x 256 B
    f32x4 *ring = reinterpret_cast<f32x4 *>(smem + 16384) + wave * 4 * 64 + lane;  // 4 KB per wave
    unsigned cw[16];
    for (int i = 0; i < 16; ++i) cw[i] = codes[(wave * 16 + i) * 64 + lane];
    float acc = 0.f;
    const f32x2 w = {1.5f, 0.5f};
    float tv[16];
    for (int s = 0; s < 16; ++s) tv[s] = (float)s;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < 16; ++s) tv[s] = *reinterpret_cast<const float *>(slice + cw[s]);
        }
        if (MODE & 2) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f32x2 in = {tv[2 * s], tv[2 * s + 1]}, out;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(out) : "v"(in), "v"(w));
                tv[2 * s] = out.x; tv[2 * s + 1] = out.y;
            }
        }
        if (MODE & 4) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 v = {tv[4 * p], tv[4 * p + 1], tv[4 * p + 2], tv[4 * p + 3]};
                asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(ring + p * 64)), "v"(v) : "memory");
            }
        }
        if (MODE & 8) {
            f32x4 v[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) asm volatile("ds_read_b128 %0, %1" : "=v"(v[p]) : "v"((unsigned)(size_t)(ring + p * 64)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < 4; ++p) { acc += v[p].x; acc += v[p].y; acc += v[p].z; acc += v[p].w; }
        }
        if ((MODE & 1) && !(MODE & 4)) {
#pragma unroll
            for (int s = 0; s < 16; ++s) { float a = tv[s]; asm volatile("" ::"v"(a)); }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + tv[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run_num(int waves, unsigned long long *cyc, float *sink, const unsigned *codes) {
    const int lds = 100000;
    hipFuncSetAttribute((const void *)num_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { num_kernel<MODE><<<256, 64 * waves, lds>>>(cyc, sink, codes); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 8 + w];
    const double per = s / (256.0 * waves) / ITERS;
    printf("num mode %2d waves %d: %7.1f ticks per 16-step unit per wave -> %.2f ticks/step aggregate per CU\n", MODE, waves, per, per / 16.0 / waves);
}

template <int VARIANT>
void run_den(int grid, int nwaves, const unsigned *masks, const float *w, int m, int rows, unsigned long long *cyc, float *sink) {
    for (int rep = 0; rep < 2; ++rep) { den_kernel<VARIANT><<<grid, 64 * nwaves>>>(masks, w, m, rows, cyc, sink); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0; for (int b = 0; b < grid; ++b) for (int wv = 0; wv < nwaves; ++wv) { s += h[b * 2 + wv]; if (h[b * 2 + wv] > mx) mx = h[b * 2 + wv]; }
    const double steps = (double)rows * (m / 32) * 32;
    printf("den variant %d grid %3d waves %d: avg %.2f max %.2f ticks per step\n", VARIANT, grid, nwaves, s / (grid * nwaves) / steps, mx / steps);
}

int main() {
    const int m = 1984, rows = 256, grid = 256;
    unsigned long long *cyc; float *sink, *w; unsigned *masks, *codes;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&masks, (size_t)grid * 2 * m * 4 + 256); hipMalloc(&w, (size_t)(rows + 1) * m * 4 + 256);
    hipMalloc(&codes, 128 * 64 * 4);
    std::vector<unsigned> hm((size_t)grid * 2 * m + 64);
    unsigned x = 777;
    for (auto &v : hm) { x = x * 1664525u + 1013904223u; v = x | (x >> 3); }
    hipMemcpy(masks, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)(rows + 1) * m + 64);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.25f + (i % 7) * 0.125f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> hc(128 * 64);
    for (int i = 0; i < 128 * 64; ++i) { x = x * 1664525u + 1013904223u; hc[i] = ((x >> 20) % 29) * 256 + (i & 63) * 4; }
    hipMemcpy(codes, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);

    for (int nw : {1, 2}) {
        run_den<2>(grid, nw, masks, w, m, rows, cyc, sink);
        run_den<1>(grid, nw, masks, w, m, rows, cyc, sink);
        run_den<0>(grid, nw, masks, w, m, rows, cyc, sink);
    }
    run_den<0>(160, 2, masks, w, m, rows, cyc, sink);
    for (int waves : {1, 4, 5, 6}) {
        run_num<1>(waves, cyc, sink, codes); run_num<4>(waves, cyc, sink, codes); run_num<5>(waves, cyc, sink, codes);
        run_num<7>(waves, cyc, sink, codes);
    }
    run_num<8>(1, cyc, sink, codes);
    return 0;
}
