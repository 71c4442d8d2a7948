// ubench_tile.hip -- the partner loop of the column-tile similarity kernel in isolation (tools/gen_tile_loop.py writes
// the loop): C columns per wave share every W row through a register; the lane's distance-table columns live in VGPRs
// and are read with relative addressing (M0 written straight from the 16-bit entry); a slot whose row takes no part
// is skipped by a scalar branch.  Checks the sums against a host loop (bit for bit) and reports valid slots per
// second chip-wide and CU-cycles per valid slot.
//   for v in branch neutral nobranch plainfma noidx noload purevalu bfesgpr purefma mfma mfmaadd mfma2; do
//       python3 tools/gen_tile_loop.py 3 16 $v LOOP3_${v^^} >> /tmp/tile_loops.h; done; python3 tools/gen_tile_loop.py 2 16 branch LOOP2_BRANCH >> /tmp/tile_loops.h
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_tile tools/ubench_tile.hip     (results: profiles/r03_ubench_tile.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NT = 21;

#define CLOBBERS                                                                                                              \
    "memory", "m0", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", \
        "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", \
        "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", \
        "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "v46", "v47",       \
        "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"

// the lane's table columns: v[64 + 21 c + a]
typedef float v32f __attribute__((ext_vector_type(32)));

#include "/tmp/tile_loops.h"
// one kernel per generated loop (C columns, variant V); REP repeats the loop over the same rows
#define TILE_KERNEL(NAME, C, INC, OUTS, PLAIN)                                                                                   \
    __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void NAME(                                        \
        const float *__restrict__ wt, size_t strip_bytes, int nstrips, const uint16_t *__restrict__ ent, size_t ldk, int nch2,    \
        const float *__restrict__ tabs, float *__restrict__ out, int rep) {                                                       \
        const int lane = threadIdx.x;                                                                                             \
        const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);                                                         \
        const char *wrow = (const char *)wt + (size_t)(wave % nstrips) * strip_bytes + (size_t)((wave * 5) % 8) * 16 * 256;       \
        const uint32_t joff = 4u * lane;                                                                                          \
        v32f TA, TB;                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 32; ++i) {                                                                          \
            TA[i] = i < NT * C ? tabs[(size_t)i * 64 + lane] : 0.0f;                                                              \
            TB[i] = 32 + i < NT * C ? tabs[(size_t)(32 + i) * 64 + lane] : 0.0f;                                                  \
        }                                                                                                                         \
        f2 n0 = {0, 0}, n1 = {0, 0}, n2 = {0, 0}, d0 = {0, 0}, d1 = {0, 0}, d2 = {0, 0};                                          \
        f4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, q0 = m0, q1 = m0, q2 = m0;                                                       \
        const float one = 1.0f;                                                                                                   \
        float n0l = 0, n0h = 0, n1l = 0, n1h = 0, n2l = 0, n2h = 0, d0l = 0, d0h = 0, d1l = 0, d1h = 0, d2l = 0, d2h = 0;         \
        const uint64_t s_wrow = (uint64_t)wrow, s_e0 = (uint64_t)(ent + (size_t)(wave * C + 0) * ldk),                            \
                       s_e1 = (uint64_t)(ent + (size_t)(wave * C + 1) * ldk),                                                     \
                       s_e2 = (uint64_t)(ent + (size_t)(wave * C + (C > 2 ? 2 : 0)) * ldk);                                       \
        _Pragma("unroll 1") for (int r = 0; r < rep; ++r) {                                                                       \
            asm volatile(                                                                                                         \
            INC                                                                                                                   \
            : OUTS                                                                                                                \
            : [joff] "v"(joff), [wrow] "s"(s_wrow), [nch2] "s"(nch2), [e0] "s"(s_e0), [e1] "s"(s_e1), [e2] "s"(s_e2),             \
              [one] "v"(one), "{v[64:95]}"(TA), "{v[96:127]}"(TB)                                                                 \
            : CLOBBERS);                                                                                                          \
        }                                                                                                                         \
        if (PLAIN >= 2) n0 = f2{m0.x, m0.y}, n1 = f2{m1.x, m1.y}, n2 = f2{m2.x, m2.y};                                          \
        if (PLAIN == 3) d0 = f2{d0l, d0h}, d1 = f2{d1l, d1h}, d2 = f2{d2l, d2h};                                                  \
        if (PLAIN == 4) d0 = f2{q0.x, q0.y}, d1 = f2{q1.x, q1.y}, d2 = f2{q2.x, q2.y};                                            \
        if (PLAIN == 1) n0 = f2{n0l, n0h}, n1 = f2{n1l, n1h}, n2 = f2{n2l, n2h}, d0 = f2{d0l, d0h}, d1 = f2{d1l, d1h}, d2 = f2{d2l, d2h}; \
        float *o = out + (size_t)wave * 16 * 64 + lane;                                                                           \
        o[0 * 64] = n0.x, o[1 * 64] = n0.y, o[2 * 64] = d0.x, o[3 * 64] = d0.y;                                                   \
        o[4 * 64] = n1.x, o[5 * 64] = n1.y, o[6 * 64] = d1.x, o[7 * 64] = d1.y;                                                   \
        o[8 * 64] = n2.x, o[9 * 64] = n2.y, o[10 * 64] = d2.x, o[11 * 64] = d2.y;                                                 \
    }
#define OUT_PK [n0] "+v"(n0), [n1] "+v"(n1), [n2] "+v"(n2), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2)
#define OUT_PL                                                                                                              \
    [n0l] "+v"(n0l), [n0h] "+v"(n0h), [n1l] "+v"(n1l), [n1h] "+v"(n1h), [n2l] "+v"(n2l), [n2h] "+v"(n2h), [d0l] "+v"(d0l), \
        [d0h] "+v"(d0h), [d1l] "+v"(d1l), [d1h] "+v"(d1h), [d2l] "+v"(d2l), [d2h] "+v"(d2h)
TILE_KERNEL(k3_branch, 3, LOOP3_BRANCH, OUT_PK, 0)
TILE_KERNEL(k2_branch, 2, LOOP2_BRANCH, OUT_PK, 0)
TILE_KERNEL(k3_neutral, 3, LOOP3_NEUTRAL, OUT_PK, 0)
TILE_KERNEL(k3_nobranch, 3, LOOP3_NOBRANCH, OUT_PK, 0)
TILE_KERNEL(k3_plainfma, 3, LOOP3_PLAINFMA, OUT_PL, 1)
TILE_KERNEL(k3_noidx, 3, LOOP3_NOIDX, OUT_PK, 0)
TILE_KERNEL(k3_noload, 3, LOOP3_NOLOAD, OUT_PK, 0)
#define OUT_M1 [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2)
#define OUT_M1A \
    [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [d0l] "+v"(d0l), [d0h] "+v"(d0h), [d1l] "+v"(d1l), [d1h] "+v"(d1h), [d2l] "+v"(d2l), [d2h] "+v"(d2h)
#define OUT_M2 [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2)
TILE_KERNEL(k3_mfma, 3, LOOP3_MFMA, OUT_M1, 2)
TILE_KERNEL(k3_mfmaadd, 3, LOOP3_MFMAADD, OUT_M1A, 3)
TILE_KERNEL(k3_mfma2, 3, LOOP3_MFMA2, OUT_M2, 4)
TILE_KERNEL(k3_purevalu, 3, LOOP3_PUREVALU, OUT_PK, 0)
TILE_KERNEL(k3_bfesgpr, 3, LOOP3_BFESGPR, OUT_PK, 0)
TILE_KERNEL(k3_purefma, 3, LOOP3_PUREFMA, OUT_PL, 1)

typedef void (*kern_t)(const float *, size_t, int, const uint16_t *, size_t, int, const float *, float *, int);

void run(kern_t kern, const char *name, int C, int waves_per_simd, double pvalid, bool check, int rep = 8, int mode = 0x2000) {
    const int rows = 2048 + 256, nstrips = 32, K = 16;
    const size_t strip_bytes = (size_t)rows * 256;
    const int nwaves = 1024 * waves_per_simd;
    const int nch2 = 2048 / (2 * K) - 4;          // double chunks per wave
    const size_t ldk = 2048 + 256;
    std::vector<float> hw((size_t)nstrips * rows * 64), htab((size_t)NT * C * 64);
    std::vector<uint16_t> hent((size_t)nwaves * C * ldk, 0);
    srand(7);
    for (auto &x : hw) x = (float)(rand() % 100000) / 131072.0f + 0.001f;
    for (auto &x : htab) x = (float)(rand() % 2000) / 97.0f;
    for (int l = 0; l < 64; ++l)
        for (int c = 0; c < C; ++c) htab[(size_t)(NT * c) * 64 + l] = 0.0f;  // index 0: the zero row (neutralised variant)
    double valid = 0;
    for (size_t i = 0; i < hent.size(); ++i) {
        const bool v = (rand() / (double)RAND_MAX) < pvalid;
        hent[i] = v ? (uint16_t)(mode | (1 + rand() % 20)) : 0;
    }
    for (int w = 0; w < nwaves; ++w)
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < nch2 * 2 * K; ++k) valid += hent[(size_t)(w * C + c) * ldk + k] != 0;
    float *dw, *dtab, *dout;
    uint16_t *dent;
    hipMalloc(&dw, hw.size() * 4);
    hipMalloc(&dtab, htab.size() * 4);
    hipMalloc(&dent, hent.size() * 2);
    hipMalloc(&dout, (size_t)nwaves * 16 * 64 * 4);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, htab.data(), htab.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dent, hent.data(), hent.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<<<nwaves, 64>>>(dw, strip_bytes, nstrips, dent, ldk, nch2, dtab, dout, 1);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        kern<<<nwaves, 64>>>(dw, strip_bytes, nstrips, dent, ldk, nch2, dtab, dout, rep);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    valid *= rep;
    printf("%-12s C %d waves/SIMD %d pvalid %.2f: %.3f ms  %.3e valid slots  %.3e slots/s  %.2f CU-cycles per valid slot at 2.0 GHz  (%s)\n", name, C,
           waves_per_simd, pvalid, best, valid, valid / (best * 1e-3), best * 1e-3 * 2.0e9 * 256 / valid, hipGetErrorString(err));
    if (check) {
        kern<<<nwaves, 64>>>(dw, strip_bytes, nstrips, dent, ldk, nch2, dtab, dout, 1);
        hipDeviceSynchronize();
        std::vector<float> ho((size_t)nwaves * 16 * 64);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int w = 0; w < nwaves; w += 97) {
            const float *ws = hw.data() + (size_t)(w % nstrips) * rows * 64 + (size_t)((w * 5) % 8) * 16 * 64;
            for (int c = 0; c < C; ++c)
                for (int l = 0; l < 64; ++l) {
                    volatile float sn = 0, sd = 0;
                    for (int k = 0; k < nch2 * 2 * K; ++k) {
                        const uint16_t e = hent[(size_t)(w * C + c) * ldk + k];
                        if (!e) continue;
                        const float wv = ws[(size_t)k * 64 + l];
                        volatile float x = wv * htab[(size_t)(NT * c + (e & 0xFF)) * 64 + l];
                        sn = sn + x;
                        sd = sd + wv;
                    }
                    const float gn = ho[(size_t)w * 16 * 64 + (4 * c + 0) * 64 + l], gd = ho[(size_t)w * 16 * 64 + (4 * c + 2) * 64 + l];
                    if (memcmp((const void *)&gn, (const void *)&sn, 4) || memcmp((const void *)&gd, (const void *)&sd, 4)) {
                        if (bad < 5) printf("  MISMATCH wave %d col %d lane %d: num %.9g vs %.9g, den %.9g vs %.9g\n", w, c, l, gn, (float)sn, gd, (float)sd);
                        ++bad;
                    }
                }
        }
        printf("  check: %ld mismatches\n", bad);
    }
    hipFree(dw);
    hipFree(dtab);
    hipFree(dent);
    hipFree(dout);
}

int main() {
    run(k3_branch, "branch", 3, 4, 0.72, true);
    run(k3_mfma, "mfma", 3, 4, 0.72, true, 8, 0x1000);
    run(k3_mfma, "mfma", 3, 3, 0.72, false, 8, 0x1000);
    run(k3_mfma, "mfma", 3, 4, 1.0, false, 8, 0x1000);
    run(k3_mfmaadd, "mfmaadd", 3, 4, 0.72, true, 8, 0x1000);
    run(k3_mfmaadd, "mfmaadd", 3, 3, 0.72, false, 8, 0x1000);
    run(k3_mfmaadd, "mfmaadd", 3, 4, 1.0, false, 8, 0x1000);
    run(k3_mfma2, "mfma2", 3, 4, 0.72, true, 8, 0x1000);
    run(k3_mfma2, "mfma2", 3, 4, 1.0, false, 8, 0x1000);
    return 0;
}
