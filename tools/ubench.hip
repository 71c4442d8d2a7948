// ubench.hip -- single-wave instruction-timing probes for the similarity chain design (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 0: two independent dependent chains of v_add_f32 (num, den)
__global__ void k_add2(float *out, unsigned long long *cyc, float x, float w) {
    float num = out[threadIdx.x], den = num * 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(num), "+v"(den) : "v"(x), "v"(w));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = num + den;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 1: one dependent chain of v_pk_add_f32
__global__ void k_pkadd(float *out, unsigned long long *cyc, float x, float w) {
    f32x2 acc = {out[threadIdx.x], 1.0f}, xv = {x, w};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(xv));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = acc.x + acc.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 2: single dependent chain of v_add_f32
__global__ void k_add1(float *out, unsigned long long *cyc, float x, float w) {
    float num = out[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(num) : "v"(x));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = num;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 3: 8 independent v_add_f32 chains (issue rate of one wave)
__global__ void k_add8(float *out, unsigned long long *cyc, float x, float w) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = out[threadIdx.x] + k;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            asm volatile(
                "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "v"(x));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[threadIdx.x + blockIdx.x * 64] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 4: consumer loop: ds_read_b128 (two steps of {x,w}) + 2 dependent v_pk_add_f32
__global__ void k_consume(float *out, unsigned long long *cyc, float x, float w) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 64];  // 64 KB: 64 reads of b128 per lane
    for (int i = threadIdx.x; i < 64 * 4 * 64; i += 64) buf[i] = x * (i & 7);
    __syncthreads();
    f32x2 acc = {0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS / 4; ++i) {
        const float4 *p = reinterpret_cast<const float4 *>(buf) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            float4 v = p[u * 64];
            f32x2 a = {v.x, v.y}, b = {v.z, v.w};
            acc += a;
            acc += b;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = acc.x + acc.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 5: monolithic step: sdwa-add address, ds_read_b64 gather, pk_mul (SGPR w), pk_add
__global__ void k_mono(float *out, unsigned long long *cyc, float x, float w, const unsigned *codes) {
    __shared__ __attribute__((aligned(16))) f32x2 tab[29 * 32];
    for (int i = threadIdx.x; i < 29 * 32; i += 64) tab[i] = f32x2{x * (i % 29), 1.0f};
    __syncthreads();
    f32x2 acc = {0.f, 0.f};
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tab) + (threadIdx.x % 20) * 256;
    unsigned cw = codes[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x2 t = *reinterpret_cast<const f32x2 *>(tb + ((cw >> (8 * s)) & 0xF8u));
                f32x2 ww = {w, w};
                acc += t * ww;
            }
            cw = cw * 1664525u + 1013904223u;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = acc.x + acc.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *out;
    unsigned long long *cyc;
    unsigned *codes;
    hipMalloc(&out, 64 * 1024 * sizeof(float));
    hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
    hipMalloc(&codes, 64 * sizeof(unsigned));
    hipMemset(out, 0, 64 * 1024 * sizeof(float));
    std::vector<unsigned> hc(64);
    for (int i = 0; i < 64; ++i) hc[i] = 0x9E3779B9u * (i + 1);
    hipMemcpy(codes, hc.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice);
    const char *names[] = {"add x2 chains (num,den) per step", "pk_add chain per step", "add x1 chain per op",
                           "8 indep adds per op", "consumer: b128 read + 2 pk_add, per step", "monolithic step"};
    const double per[] = {16.0 * ITERS, 16.0 * ITERS, 16.0 * ITERS, 16.0 * ITERS, 128.0 * (ITERS / 4), 16.0 * ITERS};
    for (int grid : {1, 256, 1024}) {
        for (int k = 0; k < 6; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (k) {
                    case 0: k_add2<<<grid, 64>>>(out, cyc, 1.0f, 0.5f); break;
                    case 1: k_pkadd<<<grid, 64>>>(out, cyc, 1.0f, 0.5f); break;
                    case 2: k_add1<<<grid, 64>>>(out, cyc, 1.0f, 0.5f); break;
                    case 3: k_add8<<<grid, 64>>>(out, cyc, 1.0f, 0.5f); break;
                    case 4: k_consume<<<grid, 64>>>(out, cyc, 1.0f, 0.5f); break;
                    case 5: k_mono<<<grid, 64>>>(out, cyc, 1.0f, 0.5f, codes); break;
                }
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double s = 0, mx = 0;
            for (auto v : h) { s += v; mx = v > mx ? v : mx; }
            printf("grid %4d  %-42s  avg %.2f  max %.2f  memtime ticks per unit\n", grid, names[k], s / grid / per[k], mx / per[k]);
        }
    }
    return 0;
}
