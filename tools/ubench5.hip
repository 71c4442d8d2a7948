// ubench5.hip -- chain-latency probes (gfx950): does a partial EXEC shorten the dependent-add latency?
// how expensive are select-then-add and fma_mix formulations of the masked denominator chain?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define ITERS 2000

// MODE 0: v_add_f32 chain; 1: v_pk_add_f32 chain; 2: cndmask(off-chain, SGPR-pair mask, s_and_b64 first) + add;
// 3: v_fma_mix_f32 chain (f16 validity operand); 4: v_fma_f32 chain; 5: s_and exec + v_add (as ubench4 variant 2);
// 6: cndmask with vcc written by s_and_b64 vcc
template <int MODE>
__global__ void k(unsigned long long *cyc, float *sink, unsigned long long execmask) {
    float acc = 0.f;
    f32x2 acc2 = {0.f, 0.f};
    float w = 0.25f, t = 0.f;
    unsigned vf = 0x3c003c00u;  // two f16 ones
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_mov_b64 s[10:11], exec\n s_mov_b64 s[14:15], -1\n s_mov_b64 s[16:17], %0\n s_mov_b64 exec, %0\n" ::"s"(execmask) : "s10", "s11", "s14", "s15", "s16", "s17");
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile(R16("v_add_f32 %0, %1, %0\n") : "+v"(acc) : "v"(w));
        if (MODE == 1) asm volatile(R16("v_pk_add_f32 %0, %1, %0\n") : "+v"(acc2) : "v"(f32x2{w, w}));
        if (MODE == 2)
            asm volatile(R16("s_and_b64 s[12:13], s[14:15], s[16:17]\n v_cndmask_b32_e64 %1, 0, %2, s[12:13]\n v_add_f32 %0, %1, %0\n")
                         : "+v"(acc), "+v"(t) : "v"(w) : "s12", "s13");
        if (MODE == 3) asm volatile(R16("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]\n") : "+v"(acc) : "v"(w), "v"(vf));
        if (MODE == 4) asm volatile(R16("v_fma_f32 %0, %1, %2, %0\n") : "+v"(acc) : "v"(w), "v"(1.0f));
        if (MODE == 5) asm volatile(R16("s_and_b64 exec, s[14:15], s[16:17]\n v_add_f32 %0, %1, %0\n") : "+v"(acc) : "v"(w));
        if (MODE == 6)
            asm volatile(R16("s_and_b64 vcc, s[14:15], s[16:17]\n v_cndmask_b32_e32 %1, 0, %2, vcc\n v_add_f32 %0, %1, %0\n")
                         : "+v"(acc), "+v"(t) : "v"(w) : "vcc");
        if (MODE == 7)  // two VALU per step, second independent (issue-limit probe)
            asm volatile(R16("v_mov_b32 %1, %2\n v_add_f32 %0, %2, %0\n") : "+v"(acc), "+v"(t) : "v"(w));
    }
    t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, s[10:11]\n" ::: "memory");
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2.x + acc2.y + t;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, unsigned long long *cyc, float *sink, unsigned long long execmask) {
    for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, 64>>>(cyc, sink, execmask); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-44s exec %016llx: %.2f ticks per step\n", name, execmask, s / 256 / ITERS / 16);
}

int main() {
    unsigned long long *cyc; float *sink;
    (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 256 * 64 * 4);
    for (unsigned long long e : {~0ull, 0xffffffffull, 0xffffull, 0xffffffff00000000ull}) {
        run<0>("v_add_f32 chain", cyc, sink, e);
        run<1>("v_pk_add_f32 chain", cyc, sink, e);
        run<4>("v_fma_f32 chain", cyc, sink, e);
        run<3>("v_fma_mix_f32 chain", cyc, sink, e);
        run<7>("v_mov + v_add chain", cyc, sink, e);
    }
    run<2>("s_and_b64 + cndmask_e64 + add", cyc, sink, ~0ull);
    run<6>("s_and_b64 vcc + cndmask_e32 + add", cyc, sink, ~0ull);
    run<2>("s_and_b64 + cndmask_e64 + add", cyc, sink, 0xffffffffull);
    run<6>("s_and_b64 vcc + cndmask_e32 + add", cyc, sink, 0xffffffffull);
    run<5>("s_and_b64 exec + add", cyc, sink, ~0ull);
    return 0;
}
