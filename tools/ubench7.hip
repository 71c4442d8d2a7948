// ubench7.hip -- "den wave" candidates (gfx950): den += valid(lane, j, k) ? W[j][k] : 0, sequential in k.
//   masks: 64-bit ballots per row k (SMEM, s_load_dwordx16 = 8 rows), W: broadcast ds_read_b128 from LDS.
//   Per step: s_and_b64 vcc, Vj, Vk ; v_cndmask_b32 t, 0, W, vcc ; v_add_f32 den, t, den   (selects run ahead of the chain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VARIANT 0: registers only (no loads); 1: SMEM masks double-buffered (16 steps per buffer) + LDS W reads
template <int VARIANT>
__global__ __launch_bounds__(64) void den_kernel(const unsigned long long *__restrict__ masks, int m, int rows,
                                                 unsigned long long *cyc, float *sink) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0.25f + (i % 7) * 0.125f;
    __syncthreads();
    const unsigned long long *mp = masks + (size_t)blockIdx.x * m;
    float den = 0.f;
    const int groups2 = m / 32;  // loop iterations: 2 groups of 16 steps
    unsigned waddr = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < rows; ++j) {
        const unsigned long long vj = __builtin_amdgcn_readfirstlane((unsigned)(mp[j])) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mp[j] >> 32)) << 32);
#define SEL(MR, WV, T) "s_and_b64 vcc, s[8:9], s[" #MR ":" #MR "+1]\n v_cndmask_b32 " T ", 0, " WV ", vcc\n"
#define ADD(T) "v_add_f32 %0, " T ", %0\n"
        if (VARIANT == 0) {
            asm volatile(
                "s_mov_b64 s[8:9], %1\n s_mov_b32 s10, %2\n"
                "s_mov_b64 s[36:37], -1\n s_mov_b64 s[38:39], 0x55555555\n s_mov_b64 s[40:41], 0x0f0f0f0f\n s_mov_b64 s[42:43], 0x00ff00ff\n"
                "1:\n"
                SEL(36, "%3", "v40") SEL(38, "%3", "v41") SEL(40, "%3", "v42") SEL(42, "%3", "v43")
                SEL(36, "%3", "v44") SEL(38, "%3", "v45") SEL(40, "%3", "v46") SEL(42, "%3", "v47")
                ADD("v40") ADD("v41") ADD("v42") ADD("v43") ADD("v44") ADD("v45") ADD("v46") ADD("v47")
                SEL(36, "%3", "v40") SEL(38, "%3", "v41") SEL(40, "%3", "v42") SEL(42, "%3", "v43")
                SEL(36, "%3", "v44") SEL(38, "%3", "v45") SEL(40, "%3", "v46") SEL(42, "%3", "v47")
                ADD("v40") ADD("v41") ADD("v42") ADD("v43") ADD("v44") ADD("v45") ADD("v46") ADD("v47")
                "s_sub_u32 s10, s10, 1\n s_cmp_lg_u32 s10, 0\n s_cbranch_scc1 1b\n"
                : "+v"(den) : "s"(vj), "s"(groups2 * 2), "v"(0.25f)
                : "s8", "s9", "s10", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "vcc", "scc",
                  "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
        } else {
            // buffers: A masks s[36:67] (16 rows), W v[48:63]; B masks s[68:99], W v[64:79]
#define SELS(MB, WB, TB)                                                                                             \
    SEL(MB + 0, "v" #WB "+0", "v" #TB "+0")
            asm volatile(
                "s_mov_b64 s[8:9], %1\n s_mov_b32 s10, %2\n s_mov_b64 s[12:13], %3\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n s_load_dwordx16 s[52:67], s[12:13], 0x40\n"
                "ds_read_b128 v[48:51], %4\n ds_read_b128 v[52:55], %4 offset:16\n ds_read_b128 v[56:59], %4 offset:32\n ds_read_b128 v[60:63], %4 offset:48\n"
                "1:\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[68:83], s[12:13], 0x80\n s_load_dwordx16 s[84:99], s[12:13], 0xc0\n"
                "ds_read_b128 v[64:67], %4 offset:64\n ds_read_b128 v[68:71], %4 offset:80\n ds_read_b128 v[72:75], %4 offset:96\n ds_read_b128 v[76:79], %4 offset:112\n"
                SEL(36, "v48", "v80") SEL(38, "v49", "v81") SEL(40, "v50", "v82") SEL(42, "v51", "v83")
                SEL(44, "v52", "v84") SEL(46, "v53", "v85") SEL(48, "v54", "v86") SEL(50, "v55", "v87")
                ADD("v80") ADD("v81") ADD("v82") ADD("v83") ADD("v84") ADD("v85") ADD("v86") ADD("v87")
                SEL(52, "v56", "v80") SEL(54, "v57", "v81") SEL(56, "v58", "v82") SEL(58, "v59", "v83")
                SEL(60, "v60", "v84") SEL(62, "v61", "v85") SEL(64, "v62", "v86") SEL(66, "v63", "v87")
                ADD("v80") ADD("v81") ADD("v82") ADD("v83") ADD("v84") ADD("v85") ADD("v86") ADD("v87")
                "s_add_u32 s12, s12, 0x100\n s_addc_u32 s13, s13, 0\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_load_dwordx16 s[36:51], s[12:13], 0x0\n s_load_dwordx16 s[52:67], s[12:13], 0x40\n"
                "ds_read_b128 v[48:51], %4\n ds_read_b128 v[52:55], %4 offset:16\n ds_read_b128 v[56:59], %4 offset:32\n ds_read_b128 v[60:63], %4 offset:48\n"
                SEL(68, "v64", "v80") SEL(70, "v65", "v81") SEL(72, "v66", "v82") SEL(74, "v67", "v83")
                SEL(76, "v68", "v84") SEL(78, "v69", "v85") SEL(80, "v70", "v86") SEL(82, "v71", "v87")
                ADD("v80") ADD("v81") ADD("v82") ADD("v83") ADD("v84") ADD("v85") ADD("v86") ADD("v87")
                SEL(84, "v72", "v80") SEL(86, "v73", "v81") SEL(88, "v74", "v82") SEL(90, "v75", "v83")
                SEL(92, "v76", "v84") SEL(94, "v77", "v85") SEL(96, "v78", "v86") SEL(98, "v79", "v87")
                ADD("v80") ADD("v81") ADD("v82") ADD("v83") ADD("v84") ADD("v85") ADD("v86") ADD("v87")
                "s_sub_u32 s10, s10, 1\n s_cmp_lg_u32 s10, 0\n s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "+v"(den) : "s"(vj), "s"(groups2), "s"(mp), "v"(waddr)
                : "s8", "s9", "s10", "s12", "s13",
                  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
                  "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
                  "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
                  "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
                  "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
                  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "vcc", "scc", "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 64 + threadIdx.x] = den;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VARIANT>
void run(const char *name, int grid, const unsigned long long *masks, int m, int rows, unsigned long long *cyc, float *sink) {
    for (int rep = 0; rep < 2; ++rep) { den_kernel<VARIANT><<<grid, 64, 16384>>>(masks, m, rows, cyc, sink); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0; for (auto v : h) { s += v; if (v > mx) mx = v; }
    const double steps = (double)rows * (m / 32) * 32;
    printf("%-40s grid %3d: avg %.2f max %.2f ticks per step\n", name, grid, s / grid / steps, mx / steps);
}
int main() {
    const int m = 1984, rows = 128, grid = 256;
    unsigned long long *cyc, *masks; float *sink;
    (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 256 * 64 * 4);
    (void)hipMalloc(&masks, (size_t)grid * m * 8 + 4096);
    std::vector<unsigned long long> hm((size_t)grid * m + 512);
    unsigned long long x = 777;
    for (auto &v : hm) { x = x * 6364136223846793005ull + 1442695040888963407ull; v = x | (x >> 3); }
    (void)hipMemcpy(masks, hm.data(), hm.size() * 8, hipMemcpyHostToDevice);
    run<0>("select + add, registers only", grid, masks, m, rows, cyc, sink);
    run<1>("select + add, SMEM masks + LDS W", grid, masks, m, rows, cyc, sink);
    run<1>("select + add, SMEM masks + LDS W", 160, masks, m, rows, cyc, sink);
    return 0;
}
