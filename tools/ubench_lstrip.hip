// ubench_lstrip.hip -- the round loop of a strip-sharing similarity kernel in isolation (tools/gen_lstrip_loop.py writes
// the loop): the G waves of a workgroup (one column each, lane = row) share every 64 x K block of W through LDS --
// copied there once per workgroup by global_load_lds_dword, read per step with ds_read_addtid_b32 -- and keep the lane's
// distance-table column in VGPRs (relative addressing through M0).  Checks the sums against a host loop (bit for bit)
// and reports CU-cycles per valid step, to be compared with the production loop's 5.4 (one W wave-load per step).
//   python3 tools/gen_lstrip_loop.py 4 64 pk LS_4_64_PK > /tmp/lstrip_loops.h; ... (see tools/ubench_lstrip.sh)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_lstrip tools/ubench_lstrip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v32f __attribute__((ext_vector_type(32)));
constexpr int NT = 21;

#define CLOBBERS                                                                                                              \
    "memory", "m0", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
        "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", \
        "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "v32", "v33", "v34", "v35", \
        "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"

#include "/tmp/lstrip_loops.h"

// lo / ix: the column's lists (LDS offset of the W row inside the two-strip ring; 0x2000 | table row), cum: entries in
// front of every strip.  wt: W in strip layout [block][row][64].
#define LS_KERNEL(NAME, G, K, INC, PK, WPE)                                                                                       \
    __global__ __launch_bounds__(64 * G) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void NAME(                                \
        const float *__restrict__ wt, size_t block_floats, int nblocks, const uint32_t *__restrict__ lo,                          \
        const uint32_t *__restrict__ ix, const uint32_t *__restrict__ cum, size_t ldk, int nstr, const float *__restrict__ tabs,  \
        float *__restrict__ out, int rep_) {                                                                                      \
        extern __shared__ float lds[];                                                                                            \
        const int lane = threadIdx.x & 63;                                                                                        \
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                                                 \
        const int col = __builtin_amdgcn_readfirstlane((int)blockIdx.x) * G + wave;                                               \
        const int rep = rep_ & 0xFFFF;                                                                                            \
        const int lcol = (rep_ >> 16) ? col % 16 : col;                                                                           \
        const float *wsrc = wt + (size_t)(blockIdx.x % nblocks) * block_floats + (size_t)wave * (K / G) * 64;                     \
        const uint32_t joff = 4u * lane;                                                                                          \
        v32f TA;                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 32; ++i) TA[i] = i < NT ? tabs[(size_t)i * 64 + lane] : 0.0f;                       \
        f2 n = {0, 0}, d = {0, 0};                                                                                                \
        float nl = 0, nh = 0, dl = 0, dh = 0;                                                                                     \
        const uint64_t s_wsrc = (uint64_t)wsrc, s_lo = (uint64_t)(lo + (size_t)lcol * ldk), s_ix = (uint64_t)(ix + (size_t)lcol * ldk), \
                       s_cum = (uint64_t)(cum + (size_t)lcol * (nstr + 1));                                                        \
        const uint32_t ldsw = (uint32_t)wave * (K / G) * 256u;                                                                    \
        _Pragma("unroll 1") for (int r = 0; r < rep; ++r) {                                                                       \
            asm volatile(INC : [n] "+v"(n), [d] "+v"(d), [nl] "+v"(nl), [nh] "+v"(nh), [dl] "+v"(dl), [dh] "+v"(dh)           \
                             : [joff] "v"(joff), [wsrc] "s"(s_wsrc), [ldsw] "s"(ldsw), [nstr] "s"(nstr), [lo] "s"(s_lo),          \
                               [ix] "s"(s_ix), [cum] "s"(s_cum), "{v[64:95]}"(TA)                                                 \
                             : CLOBBERS);                                                                                         \
        }                                                                                                                         \
        if (!PK) n = f2{nl, nh}, d = f2{dl, dh};                                                                                  \
        float *o = out + (size_t)col * 4 * 64 + lane;                                                                             \
        o[0] = n.x, o[64] = n.y, o[128] = d.x, o[192] = d.y;                                                                      \
        if (lds[0] == 12345.678f) o[0] = 0;                                                                                       \
    }

LS_KERNEL(ls_4_64_pk, 4, 64, LS_4_64_PK, 1, 4)
LS_KERNEL(ls_4_64_fma, 4, 64, LS_4_64_FMA, 0, 4)
LS_KERNEL(ls_8_64_pk, 8, 64, LS_8_64_PK, 1, 4)
LS_KERNEL(ls_8_64_fma, 8, 64, LS_8_64_FMA, 0, 4)
LS_KERNEL(ls_4_64_fma_nofill, 4, 64, LS_4_64_FMA_NOFILL, 0, 4)
LS_KERNEL(ls_4_64_fma_nobar, 4, 64, LS_4_64_FMA_NOBAR, 0, 4)
LS_KERNEL(ls_4_32_fma, 4, 32, LS_4_32_FMA, 0, 4)
LS_KERNEL(ls_4_64_pkg, 4, 64, LS_4_64_PKG, 1, 4)
LS_KERNEL(ls_4_64_fmag, 4, 64, LS_4_64_FMAG, 0, 4)
LS_KERNEL(ls_8_64_pkg6, 8, 64, LS_8_64_PKG, 1, 6)
LS_KERNEL(ls_8_64_pkg5, 8, 64, LS_8_64_PKG, 1, 5)

typedef void (*kern_t)(const float *, size_t, int, const uint32_t *, const uint32_t *, const uint32_t *, size_t, int, const float *,
                       float *, int);

void run(kern_t kern, const char *name, int G, int K, int waves_per_simd, double pvalid, bool check, bool exact = true, int rep = 8, bool flat = false) {
    const int m = 2048, nstr = flat ? 1 : m / K, nblocks = 32;
    const size_t rows = m + 2 * K, block_floats = rows * 64;
    const int ncols = 1024 * waves_per_simd;  // one wave per column
    const size_t ldk = m + 64;
    std::vector<float> hw((size_t)nblocks * block_floats), htab((size_t)NT * 64);
    std::vector<uint32_t> hlo((size_t)ncols * ldk, 0), hix((size_t)ncols * ldk, 0x2000), hcum((size_t)ncols * (nstr + 1), 0);
    std::vector<uint16_t> hk((size_t)ncols * ldk, 0);
    srand(7);
    for (auto &x : hw) x = (float)(rand() % 100000) / 131072.0f + 0.001f;
    for (auto &x : htab) x = (float)(rand() % 2000) / 97.0f;
    double valid = 0;
    for (int c = 0; c < ncols; ++c) {
        uint32_t cnt = 0;
        for (int k = 0; k < m; ++k) {
            if (!flat && k % K == 0) hcum[(size_t)c * (nstr + 1) + k / K] = cnt;
            if ((rand() / (double)RAND_MAX) < pvalid) {
                hlo[(size_t)c * ldk + cnt] = (uint32_t)(k % (2 * K)) * 256u;
                hix[(size_t)c * ldk + cnt] = 0x2000u | (uint32_t)(1 + rand() % 20);
                hk[(size_t)c * ldk + cnt] = (uint16_t)k;
                ++cnt;
            }
        }
        hcum[(size_t)c * (nstr + 1) + nstr] = cnt;
        valid += cnt;
    }
    float *dw, *dtab, *dout;
    uint32_t *dlo, *dix, *dcum;
    hipMalloc(&dw, hw.size() * 4);
    hipMalloc(&dtab, htab.size() * 4);
    hipMalloc(&dlo, hlo.size() * 4);
    hipMalloc(&dix, hix.size() * 4);
    hipMalloc(&dcum, hcum.size() * 4);
    hipMalloc(&dout, (size_t)ncols * 4 * 64 * 4);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, htab.data(), htab.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dlo, hlo.data(), hlo.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dix, hix.data(), hix.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcum, hcum.data(), hcum.size() * 4, hipMemcpyHostToDevice);
    const size_t dyn = (size_t)2 * K * 256 + 256;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = ncols / G;
    kern<<<grid, 64 * G, dyn>>>(dw, block_floats, nblocks, dlo, dix, dcum, ldk, nstr, dtab, dout, 1);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        kern<<<grid, 64 * G, dyn>>>(dw, block_floats, nblocks, dlo, dix, dcum, ldk, nstr, dtab, dout, rep);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double steps = valid * (rep & 0xFFFF);
    printf("%-16s G %d K %3d waves/SIMD %d pvalid %.2f: %.3f ms  %.3e steps  %.2f CU-cycles per valid step at 2.0 GHz  (%s)\n", name, G, K,
           waves_per_simd, pvalid, best, steps, best * 1e-3 * 2.0e9 * 256 / steps, hipGetErrorString(err));
    if (check) {
        kern<<<grid, 64 * G, dyn>>>(dw, block_floats, nblocks, dlo, dix, dcum, ldk, nstr, dtab, dout, 1);
        hipDeviceSynchronize();
        std::vector<float> ho((size_t)ncols * 4 * 64);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int c = 0; c < ncols; c += 61) {
            const float *ws = hw.data() + (size_t)((c / G) % nblocks) * block_floats;
            const uint32_t cnt = hcum[(size_t)c * (nstr + 1) + nstr];
            for (int l = 0; l < 64; ++l) {
                volatile float sn = 0, sd = 0;
                for (uint32_t t = 0; t < cnt; ++t) {
                    const int k = hk[(size_t)c * ldk + t];
                    const float wv = exact ? ws[(size_t)k * 64 + l] : 1.0f;
                    volatile float x = wv * htab[(size_t)(hix[(size_t)c * ldk + t] & 0xFF) * 64 + l];
                    sn = sn + x;
                    sd = sd + wv;
                }
                const float gn = ho[(size_t)c * 256 + l], gd = ho[(size_t)c * 256 + 128 + l];
                if (memcmp((const void *)&gn, (const void *)&sn, 4) || memcmp((const void *)&gd, (const void *)&sd, 4)) {
                    if (bad < 5) printf("  MISMATCH col %d lane %d: num %.9g vs %.9g, den %.9g vs %.9g\n", c, l, gn, (float)sn, gd, (float)sd);
                    ++bad;
                }
            }
        }
        printf("  check: %ld mismatches\n", bad);
    }
    hipFree(dw);
    hipFree(dtab);
    hipFree(dlo);
    hipFree(dix);
    hipFree(dcum);
    hipFree(dout);
}

int main() {
    run(ls_4_64_pk, "pk", 4, 64, 4, 0.72, true);
    run(ls_4_64_fma, "fma", 4, 64, 4, 0.72, true);
    run(ls_8_64_pk, "pk", 8, 64, 4, 0.72, true);
    run(ls_8_64_fma, "fma", 8, 64, 4, 0.72, true);
    run(ls_4_32_fma, "fma", 4, 32, 4, 0.72, true);
    run(ls_4_64_fma, "fma", 4, 64, 3, 0.72, false);
    run(ls_4_64_fma, "fma", 4, 64, 2, 0.72, false);
    run(ls_4_64_fma, "fma", 4, 64, 4, 1.0, false);
    run(ls_4_64_fma, "fma", 4, 64, 4, 0.4, true);
    run(ls_4_64_fma, "fma 16 lists", 4, 64, 4, 0.72, false, true, 8 | 0x10000);
    run(ls_4_64_fma, "fma flat", 4, 64, 4, 0.72, false, true, 8, true);
    run(ls_4_64_fma, "fma flat 16 l", 4, 64, 4, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_pk, "pk flat 16 l", 4, 64, 4, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_fma, "fma flat 16 l", 4, 64, 3, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_fma, "fma flat 16 l", 4, 64, 2, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_fma, "fma flat 16 l", 4, 64, 1, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_pkg, "pk grouped", 4, 64, 4, 0.72, true);
    run(ls_4_64_pkg, "pkg flat 16 l", 4, 64, 4, 0.72, false, true, 8 | 0x10000, true);
    run(ls_4_64_fmag, "fmag flat 16 l", 4, 64, 4, 0.72, false, true, 8 | 0x10000, true);
    run(ls_8_64_pkg5, "pkg5 flat 16 l", 8, 64, 5, 0.72, false, true, 8 | 0x10000, true);
    run(ls_8_64_pkg6, "pkg6 flat 16 l", 8, 64, 6, 0.72, false, true, 8 | 0x10000, true);
    run(ls_8_64_pkg6, "pkg6 16 l", 8, 64, 6, 0.72, true, true, 8 | 0x10000, false);
    run(ls_4_64_fma_nofill, "fma nofill", 4, 64, 4, 0.72, false);
    run(ls_4_64_fma_nobar, "fma nobarrier", 4, 64, 4, 0.72, false);
    return 0;
}
