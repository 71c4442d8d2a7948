// probe: global_load_lds_dword semantics on gfx950 (LDS address = M0 + offset + lane*4 ?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* w, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -1.f;
    __syncthreads();
    unsigned off = (threadIdx.x & 15) * 4;
    unsigned m0v = 1024 + (threadIdx.x >> 6) * 512;   // wave 0 -> byte 1024, wave 1 -> byte 1536
    m0v = __builtin_amdgcn_readfirstlane(m0v);
    const float* wp = w + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 100;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2 offset:16" :: "s"(m0v), "v"(off), "s"(wp) : "m0", "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
    float *w, *out; hipMalloc(&w, 4096); hipMalloc(&out, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i; hipMemcpy(w, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 128, 8192>>>(w, out); hipDeviceSynchronize();
    hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; ++i) if (h[i] != -1.f) printf("lds[%d]=%g\n", i, h[i]);
    return 0;
}
