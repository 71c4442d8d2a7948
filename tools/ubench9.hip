// ubench9.hip -- chains fed through DPP: 4 (or 2) lanes per column each prepare the term of their own step, and
// the chain add reads the term of lane i of the quad through quad_perm, so that no product passes through LDS.
//   mode 0: 16 dependent v_add_f32_dpp                                  (latency of a DPP add in a chain)
//   mode 1: numerator block of 16 steps: 4 v_perm + 4 ds_read_b32 (gather) + 1 ds_read_b128 (W) + 2 v_pk_mul + 16 adds
//   mode 2: denominator block, 2 lanes per column: 8 v_cndmask (SGPR-pair masks) + 2 ds_read_b128 (W) + 16 adds
//   mode 3: denominator block, 4 lanes per column: 4 v_cndmask + 1 ds_read_b128 + 16 adds
// Each mode runs 4 waves per workgroup (one per SIMD) on every CU; reads are issued one block ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 2000
#define A4(p0, p1, p2, p3, t)                                                                            \
    "v_add_f32_dpp v0, " t ", v0 quad_perm:[" p0 "," p0 "," p0 "," p0 "] row_mask:0xf bank_mask:0xf\n" \
    "v_add_f32_dpp v0, " t ", v0 quad_perm:[" p1 "," p1 "," p1 "," p1 "] row_mask:0xf bank_mask:0xf\n" \
    "v_add_f32_dpp v0, " t ", v0 quad_perm:[" p2 "," p2 "," p2 "," p2 "] row_mask:0xf bank_mask:0xf\n" \
    "v_add_f32_dpp v0, " t ", v0 quad_perm:[" p3 "," p3 "," p3 "," p3 "] row_mask:0xf bank_mask:0xf\n"
#define QUAD(t) A4("0", "1", "2", "3", t)
#define PAIR2(ta, tb)                                                                                  \
    "v_add_f32_dpp v0, " ta ", v0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n"                    \
    "v_add_f32_dpp v0, " ta ", v0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n"                    \
    "v_add_f32_dpp v0, " tb ", v0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n"                    \
    "v_add_f32_dpp v0, " tb ", v0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n"
template <int MODE>
__global__ void k(unsigned long long *cyc, float *sink, const uint32_t *codes) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f + (i & 7) * 0.125f;
    __syncthreads();
    const uint32_t cw = codes[lane] & 0x0f0f0f0fu;            // four one-byte codes < 16
    const uint32_t vlane = lane * 4u + (wave << 16 >> 3);     // byte 0 = lane*4; byte 1 takes the code; wave slice
    const uint32_t waddr = 24576u + (lane & 3) * 16u;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0)
        asm volatile("v_mov_b32 v0, 0\n v_mov_b32 v1, 1.0\n s_mov_b32 s10, %1\n1:\n" QUAD("v1") QUAD("v1") QUAD("v1") QUAD("v1")
                     "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n v_mov_b32 %0, v0\n"
                     : "=v"(acc) : "s"(ITERS - 1) : "s10", "v0", "v1", "scc");
    if (MODE == 1)
        // set A: addr v[2:5] D v[6:9] W v[10:13]; set B: addr v[14:17] D v[18:21] W v[22:25]
        asm volatile(
            "v_mov_b32 v0, 0\n s_mov_b32 s10, %1\n s_mov_b32 s11, 0x07060004\n s_mov_b32 s12, 0x07060104\n s_mov_b32 s13, 0x07060204\n s_mov_b32 s14, 0x07060304\n"
            "v_perm_b32 v2, %2, %3, s11\n v_perm_b32 v3, %2, %3, s12\n v_perm_b32 v4, %2, %3, s13\n v_perm_b32 v5, %2, %3, s14\n"
            "ds_read_b32 v6, v2\n ds_read_b32 v7, v3\n ds_read_b32 v8, v4\n ds_read_b32 v9, v5\n ds_read_b128 v[10:13], %4\n"
            "1:\n"
            "v_perm_b32 v14, %2, %3, s11\n v_perm_b32 v15, %2, %3, s12\n v_perm_b32 v16, %2, %3, s13\n v_perm_b32 v17, %2, %3, s14\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_read_b32 v18, v14\n ds_read_b32 v19, v15\n ds_read_b32 v20, v16\n ds_read_b32 v21, v17\n ds_read_b128 v[22:25], %4 offset:64\n"
            "v_pk_mul_f32 v[6:7], v[6:7], v[10:11]\n v_pk_mul_f32 v[8:9], v[8:9], v[12:13]\n"
            QUAD("v6") QUAD("v7") QUAD("v8") QUAD("v9")
            "v_perm_b32 v2, %2, %3, s11\n v_perm_b32 v3, %2, %3, s12\n v_perm_b32 v4, %2, %3, s13\n v_perm_b32 v5, %2, %3, s14\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_read_b32 v6, v2\n ds_read_b32 v7, v3\n ds_read_b32 v8, v4\n ds_read_b32 v9, v5\n ds_read_b128 v[10:13], %4\n"
            "v_pk_mul_f32 v[18:19], v[18:19], v[22:23]\n v_pk_mul_f32 v[20:21], v[20:21], v[24:25]\n"
            QUAD("v18") QUAD("v19") QUAD("v20") QUAD("v21")
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v0\n"
            : "=v"(acc) : "s"(ITERS / 2 - 1), "v"(cw), "v"(vlane), "v"(waddr)
            : "s10", "s11", "s12", "s13", "s14", "scc", "memory", "v0", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",
              "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25");
    if (MODE == 2)
        asm volatile(
            "v_mov_b32 v0, 0\n s_mov_b32 s10, %1\n s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5f5f5f5f\n"
            "ds_read_b128 v[10:13], %2\n ds_read_b128 v[14:17], %2 offset:32\n"
            "1:\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_cndmask_b32_e64 v2, 0, v10, s[20:21]\n v_cndmask_b32_e64 v3, 0, v11, s[22:23]\n v_cndmask_b32_e64 v4, 0, v12, s[20:21]\n v_cndmask_b32_e64 v5, 0, v13, s[22:23]\n"
            "v_cndmask_b32_e64 v6, 0, v14, s[20:21]\n v_cndmask_b32_e64 v7, 0, v15, s[22:23]\n v_cndmask_b32_e64 v8, 0, v16, s[20:21]\n v_cndmask_b32_e64 v9, 0, v17, s[22:23]\n"
            "ds_read_b128 v[10:13], %2 offset:64\n ds_read_b128 v[14:17], %2 offset:96\n"
            "s_load_dwordx16 s[36:51], %3, 0x0\n"
            PAIR2("v2", "v3") PAIR2("v4", "v5") PAIR2("v6", "v7") PAIR2("v8", "v9")
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v0\n"
            : "=v"(acc) : "s"(ITERS - 1), "v"(waddr), "s"(codes)
            : "s10", "s20", "s21", "s22", "s23", "scc", "memory", "v0", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
              "v16", "v17", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
    if (MODE == 3)
        asm volatile(
            "v_mov_b32 v0, 0\n s_mov_b32 s10, %1\n s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5f5f5f5f\n"
            "ds_read_b128 v[10:13], %2\n"
            "1:\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_cndmask_b32_e64 v2, 0, v10, s[20:21]\n v_cndmask_b32_e64 v3, 0, v11, s[22:23]\n v_cndmask_b32_e64 v4, 0, v12, s[20:21]\n v_cndmask_b32_e64 v5, 0, v13, s[22:23]\n"
            "ds_read_b128 v[10:13], %2 offset:64\n"
            "s_load_dwordx8 s[36:43], %3, 0x0\n"
            QUAD("v2") QUAD("v3") QUAD("v4") QUAD("v5")
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v0\n"
            : "=v"(acc) : "s"(ITERS - 1), "v"(waddr), "s"(codes)
            : "s10", "s20", "s21", "s22", "s23", "scc", "memory", "v0", "v2", "v3", "v4", "v5", "v10", "v11", "v12", "v13", "s36", "s37", "s38", "s39", "s40",
              "s41", "s42", "s43");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE>
void run(const char *name, unsigned long long *cyc, float *sink, const uint32_t *codes, int waves) {
    for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, 64 * waves, 32768>>>(cyc, sink, codes); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 4 + w];
    printf("%-72s %d waves/CU: %.2f ticks per step\n", name, waves, s / 256 / waves / ITERS / 16);
}
int main() {
    unsigned long long *cyc; float *sink; uint32_t *codes;
    (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&sink, 256 * 256 * 4); (void)hipMalloc(&codes, 4096);
    std::vector<uint32_t> hc(1024); for (int i = 0; i < 1024; ++i) hc[i] = 0x9e3779b9u * (i + 1);
    (void)hipMemcpy(codes, hc.data(), 4096, hipMemcpyHostToDevice);
    for (int waves : {1, 4}) {
        run<0>("16 dependent v_add_f32_dpp", cyc, sink, codes, waves);
        run<1>("numerator, 4 lanes/column: 4 perm + 4 gather + W + 2 pk_mul + 16 adds", cyc, sink, codes, waves);
        run<2>("denominator, 2 lanes/column: 8 cndmask + 2 W reads + s_load + 16 adds", cyc, sink, codes, waves);
        run<3>("denominator, 4 lanes/column: 4 cndmask + 1 W read + s_load + 16 adds", cyc, sink, codes, waves);
    }
    return 0;
}
