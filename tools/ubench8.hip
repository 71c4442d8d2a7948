// ubench8.hip -- den chain as select + fma: t = mask ? 1.0 : 0 (v_cndmask_b32_e64, SGPR-pair mask), den = fma(W, t, den)
// with W an SGPR operand.  Exact (W * 1 and W * 0 are exact).  Registers only: per-step issue cost of the pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 4000
template <int MODE>
__global__ void k(unsigned long long *cyc, float *sink) {
    float den = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0)  // selects run 8 steps ahead of the chain
        asm volatile(
            "s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5555\n s_mov_b32 s24, 0.5\n s_mov_b32 s25, 2.0\n s_mov_b32 s10, %1\n"
            "1:\n"
            "v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n v_cndmask_b32_e64 v41, 0, 1.0, s[22:23]\n"
            "v_cndmask_b32_e64 v42, 0, 1.0, s[20:21]\n v_cndmask_b32_e64 v43, 0, 1.0, s[22:23]\n"
            "v_cndmask_b32_e64 v44, 0, 1.0, s[20:21]\n v_cndmask_b32_e64 v45, 0, 1.0, s[22:23]\n"
            "v_cndmask_b32_e64 v46, 0, 1.0, s[20:21]\n v_cndmask_b32_e64 v47, 0, 1.0, s[22:23]\n"
            "v_fma_f32 %0, s24, v40, %0\n v_fma_f32 %0, s25, v41, %0\n v_fma_f32 %0, s24, v42, %0\n v_fma_f32 %0, s25, v43, %0\n"
            "v_fma_f32 %0, s24, v44, %0\n v_fma_f32 %0, s25, v45, %0\n v_fma_f32 %0, s24, v46, %0\n v_fma_f32 %0, s25, v47, %0\n"
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n"
            : "+v"(den) : "s"(ITERS - 1) : "s10", "s20", "s21", "s22", "s23", "s24", "s25", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "scc");
    if (MODE == 1)  // interleaved select / fma
        asm volatile(
            "s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5555\n s_mov_b32 s24, 0.5\n s_mov_b32 s25, 2.0\n s_mov_b32 s10, %1\n"
            "v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n"
            "1:\n"
            "v_cndmask_b32_e64 v41, 0, 1.0, s[22:23]\n v_fma_f32 %0, s24, v40, %0\n v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n v_fma_f32 %0, s25, v41, %0\n"
            "v_cndmask_b32_e64 v41, 0, 1.0, s[22:23]\n v_fma_f32 %0, s24, v40, %0\n v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n v_fma_f32 %0, s25, v41, %0\n"
            "v_cndmask_b32_e64 v41, 0, 1.0, s[22:23]\n v_fma_f32 %0, s24, v40, %0\n v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n v_fma_f32 %0, s25, v41, %0\n"
            "v_cndmask_b32_e64 v41, 0, 1.0, s[22:23]\n v_fma_f32 %0, s24, v40, %0\n v_cndmask_b32_e64 v40, 0, 1.0, s[20:21]\n v_fma_f32 %0, s25, v41, %0\n"
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n"
            : "+v"(den) : "s"(ITERS - 1) : "s10", "s20", "s21", "s22", "s23", "s24", "s25", "v40", "v41", "scc");
    if (MODE == 2)  // masked W directly: t = mask ? W : 0 (W in a VGPR), den += t
        asm volatile(
            "s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5555\n v_mov_b32 v50, 0.5\n s_mov_b32 s10, %1\n"
            "1:\n"
            "v_cndmask_b32_e64 v40, 0, v50, s[20:21]\n v_cndmask_b32_e64 v41, 0, v50, s[22:23]\n"
            "v_cndmask_b32_e64 v42, 0, v50, s[20:21]\n v_cndmask_b32_e64 v43, 0, v50, s[22:23]\n"
            "v_cndmask_b32_e64 v44, 0, v50, s[20:21]\n v_cndmask_b32_e64 v45, 0, v50, s[22:23]\n"
            "v_cndmask_b32_e64 v46, 0, v50, s[20:21]\n v_cndmask_b32_e64 v47, 0, v50, s[22:23]\n"
            "v_add_f32 %0, v40, %0\n v_add_f32 %0, v41, %0\n v_add_f32 %0, v42, %0\n v_add_f32 %0, v43, %0\n"
            "v_add_f32 %0, v44, %0\n v_add_f32 %0, v45, %0\n v_add_f32 %0, v46, %0\n v_add_f32 %0, v47, %0\n"
            "s_sub_u32 s10, s10, 1\n s_cbranch_scc0 1b\n"
            : "+v"(den) : "s"(ITERS - 1) : "s10", "s20", "s21", "s22", "s23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50", "scc");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 64 + threadIdx.x] = den;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, unsigned long long *cyc, float *sink) {
    for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, 64>>>(cyc, sink); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-58s %.2f ticks per step\n", name, s / 256 / ITERS / 8);
}
int main() {
    unsigned long long *cyc; float *sink;
    (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 256 * 64 * 4);
    run<0>("cndmask(1.0) x8 then fma(W sgpr) x8", cyc, sink);
    run<1>("cndmask / fma interleaved", cyc, sink);
    run<2>("cndmask(W vgpr) x8 then v_add x8", cyc, sink);
    return 0;
}
