// addtid_probe.hip -- where does ds_write_addtid_b32 put lane L of wave W?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, int m0_per_wave, int m0_base) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = -1;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int val = wave * 1000 + lane;
    const int m0 = __builtin_amdgcn_readfirstlane(m0_base + wave * m0_per_wave);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tds_write_addtid_b32 %0 offset:64\n\ts_waitcnt lgkmcnt(0)" ::"v"(val), "s"(m0) : "m0", "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) out[i] = lds[i];
}
int main() {
    int *out; (void)hipMalloc(&out, 8192 * 4);
    std::vector<int> h(8192);
    for (int waves : {1, 4})
        for (int per : {0, 4096})
            for (int base : {0, 256, 4096, 16384}) {
                k<<<1, 64 * waves, 32768>>>(out, per, base); (void)hipDeviceSynchronize();
                (void)hipMemcpy(h.data(), out, 8192 * 4, hipMemcpyDeviceToHost);
                printf("waves %d  M0 = %5d + %4d * wave:", waves, base, per);
                for (int w = 0; w < waves; ++w) {
                    int at0 = -1, at63 = -1, cnt = 0;
                    for (int i = 0; i < 8192; ++i) { if (h[i] == w * 1000) at0 = i * 4; if (h[i] == w * 1000 + 63) at63 = i * 4; if (h[i] / 1000 == w && h[i] >= 0) ++cnt; }
                    printf("  w%d: lane0 @%6d lane63 @%6d (%d found)", w, at0, at63, cnt);
                }
                printf("\n");
            }
    return 0;
}
