// ubench6.hip -- dependent v_pk_add_f32 chain forms (gfx950): operand position of the accumulator,
// the s_nop the compiler inserts between dependent packed-fp32 ops, fillers in that slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define ITERS 2000
template <int MODE>
__global__ void k(unsigned long long *cyc, float *sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f; lds[threadIdx.x + 64] = 2.0f; lds[threadIdx.x + 128] = 3.f; lds[threadIdx.x + 192] = 4.f;
    __syncthreads();
    f32x2 acc = {0.f, 0.f}, acc2 = {0.f, 0.f};
    float a1 = 0.f, a2 = 0.f, t = 0.f;
    f32x2 w = {0.37f + threadIdx.x * 1e-3f, 0.11f};
    f32x4 rd = {0.f, 0.f, 0.f, 0.f};
    unsigned addr = threadIdx.x * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile(R16("v_pk_add_f32 %0, %1, %0\n") : "+v"(acc) : "v"(w));
        if (MODE == 1) asm volatile(R16("v_pk_add_f32 %0, %1, %0\n s_nop 0\n") : "+v"(acc) : "v"(w));
        if (MODE == 2) asm volatile(R16("v_pk_add_f32 %0, %0, %1\n") : "+v"(acc) : "v"(w));
        if (MODE == 3) asm volatile(R16("v_pk_add_f32 %0, %0, %1\n s_nop 0\n") : "+v"(acc) : "v"(w));
        if (MODE == 4) asm volatile(R16("v_pk_add_f32 %0, %2, %0\n v_mov_b32 %1, %3\n") : "+v"(acc), "+v"(t) : "v"(w), "v"(w.x));
        if (MODE == 5) asm volatile(R4("v_pk_add_f32 %0, %2, %0\n ds_read_b128 %1, %3\n v_pk_add_f32 %0, %2, %0\n s_nop 0\n"
                                       "v_pk_add_f32 %0, %2, %0\n ds_read_b128 %1, %3\n v_pk_add_f32 %0, %2, %0\n s_nop 0\n")
                                    "s_waitcnt lgkmcnt(0)\n" : "+v"(acc), "+v"(rd) : "v"(w), "v"(addr));
        if (MODE == 6) asm volatile(R16("v_add_f32 %0, %2, %0\n v_add_f32 %1, %3, %1\n") : "+v"(a1), "+v"(a2) : "v"(w.x), "v"(w.y));
        if (MODE == 7) asm volatile(R4("v_pk_add_f32 %1, %2, %0\n s_nop 0\n v_pk_add_f32 %0, %2, %1\n s_nop 0\n"
                                       "v_pk_add_f32 %1, %2, %0\n s_nop 0\n v_pk_add_f32 %0, %2, %1\n s_nop 0\n") : "+v"(acc), "+v"(acc2) : "v"(w));
        if (MODE == 8) asm volatile(R16("v_pk_add_f32 %0, %1, %0\n s_nop 1\n") : "+v"(acc) : "v"(w));
        if (MODE == 9) asm volatile(R16("v_add_f32 %0, %1, %0\n") : "+v"(a1) : "v"(w.x));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[(blockIdx.x * 64 + threadIdx.x) * 2] = acc.x + acc2.x + a1 + t + rd.x;
    sink[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = acc.y + acc2.y + a2;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, unsigned long long *cyc, float *sink) {
    for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, 64, 4096>>>(cyc, sink); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256);
    std::vector<float> hs(4);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs.data(), sink + 2 * 5, 16, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-52s %.2f ticks per step   (lane5: %.9g %.9g)\n", name, s / 256 / ITERS / 16, hs[0], hs[1]);
}
int main() {
    unsigned long long *cyc; float *sink;
    (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 256 * 64 * 8);
    run<0>("pk_add acc=src1", cyc, sink);
    run<1>("pk_add acc=src1 + s_nop 0", cyc, sink);
    run<8>("pk_add acc=src1 + s_nop 1", cyc, sink);
    run<2>("pk_add acc=src0", cyc, sink);
    run<3>("pk_add acc=src0 + s_nop 0", cyc, sink);
    run<4>("pk_add acc=src1 + independent v_mov", cyc, sink);
    run<5>("pk_add acc=src1, ds_read_b128 / s_nop alternating", cyc, sink);
    run<7>("pk_add ping-pong acc (src1) + s_nop 0", cyc, sink);
    run<6>("v_add num + v_add den (src1), 2 chains", cyc, sink);
    run<9>("v_add acc=src1 single chain", cyc, sink);
    return 0;
}
