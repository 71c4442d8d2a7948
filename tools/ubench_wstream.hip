// ubench_wstream.hip -- what does the W stream of the similarity kernel cost the vector-memory pipeline?
// Every wave walks rows of a float matrix (row stride 8 KB, as wlow at m = 2000) with one coalesced buffer load
// per row, VEC dwords per lane (VEC = 1: 256 B per row and wave, the kernel's pattern; 2: 512 B; 4: 1 KB), rows
// 16 in flight, SGPR row offsets.  The matrix (2 MB or 16 MB) is shared by all waves: L2 / Infinity-Cache hits.
// Reports bytes per clock and CU and cycles of the CU per wave-load.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_wstream tools/ubench_wstream.hip && tools/ubench_wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VEC, int POL = 0>
__global__ __launch_bounds__(256) void k(const float *w, uint32_t wbytes, int rows, int iters, float *sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, (int)wbytes, 0x00027000);
    typedef float fv __attribute__((ext_vector_type(VEC)));
    float acc = 0.f;
    // strips of 64 * VEC columns; the wave starts at a row of its own
    const uint32_t joff = (uint32_t)(lane * 4 * VEC) + (uint32_t)((wave % (2048 / (64 * VEC))) * 256 * VEC);
    uint32_t r = (uint32_t)((wave * 37) % rows);
    fv v[16];
    auto ld = [&](uint32_t row) {
        const uint32_t so = row * 8192u;
        if constexpr (VEC == 1 && POL == 4) {  // global_load_dword, 64-bit per-lane address (no buffer descriptor)
            const float *p = (const float *)((const char *)w + so);
            return fv{*(const __attribute__((address_space(1))) float *)((const char *)p + joff)};
        } else if constexpr (VEC == 1 && POL == 5) {  // global_load_dword v, v_offset32, s[base:base+1]
            const uint64_t base = (uint64_t)w + so;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            const uint64_t sb = ((uint64_t)hi << 32) | lo;
            float d;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(joff), "s"(sb) : "memory");
            return fv{d};
        } else if constexpr (VEC == 1) return fv{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, joff, so, POL))};
        else if constexpr (POL == 4) {  // global loads of 2 / 4 dwords per lane
            const char *p = (const char *)w + so;
            return *(const __attribute__((address_space(1))) fv *)(p + joff);
        } else if constexpr (VEC == 2) return __builtin_bit_cast(fv, __builtin_amdgcn_raw_buffer_load_b64(rsrc, joff, so, 0));
        else return __builtin_bit_cast(fv, __builtin_amdgcn_raw_buffer_load_b128(rsrc, joff, so, 0));
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = ld(r);
        r = r + 1 < (uint32_t)rows ? r + 1 : 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (POL == 5) asm volatile("s_waitcnt vmcnt(15)" : "+v"(v[i][0])::"memory");
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += v[i][e];
            v[i] = ld(r);
            r = r + 1 < (uint32_t)rows ? r + 1 : 0;
        }
    }
    if constexpr (POL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i][0];
    if (acc == 12345.678f) sink[0] = acc;
}

template <int VEC, int POL = 0>
void run(const float *w, size_t wbytes, int rows, int waves_per_simd, float *sink) {
    const int iters = 2000 / VEC;
    const int grid = 256 * waves_per_simd;  // 4 waves per workgroup
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<VEC, POL><<<grid, 256>>>(w, (uint32_t)wbytes, rows, 10, sink);
    hipEventRecord(a);
    k<VEC, POL><<<grid, 256>>>(w, (uint32_t)wbytes, rows, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double loads = (double)grid * 4 * (iters + 1) * 16;
    const double bytes = loads * 256 * VEC;
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double cyc = ms * 1e-3 * clk * 1e3;
    if (POL) printf("[policy %d: 1 glc, 2 slc, 3 glc+slc, 4 global_load vaddr64, 5 global_load saddr+voffset] ", POL);
    printf("VEC %d rows %5d waves/SIMD %d: %.3f ms  %.1f TB/s  %.1f B/clk/CU  %.2f CU-cycles per wave-load (at %.2f GHz nominal)\n", VEC, rows,
           waves_per_simd, ms, bytes / ms * 1e-9, bytes / 256 / cyc, cyc * 256 / loads, clk * 1e-6);
}

int main() {
    const size_t rowsmax = 2048;
    float *w, *sink;
    hipMalloc(&w, rowsmax * 8192);
    hipMalloc(&sink, 64);
    hipMemset(w, 0, rowsmax * 8192);
    for (int rows : {256, 2048})
        for (int wps : {4, 6, 8}) {
            run<1>(w, rowsmax * 8192, rows, wps, sink);
            run<2>(w, rowsmax * 8192, rows, wps, sink);
            run<4>(w, rowsmax * 8192, rows, wps, sink);
        }
    for (int rows : {256, 2048}) {
        run<1, 1>(w, rowsmax * 8192, rows, 6, sink);
        run<1, 2>(w, rowsmax * 8192, rows, 6, sink);
        run<1, 3>(w, rowsmax * 8192, rows, 6, sink);
        run<1, 4>(w, rowsmax * 8192, rows, 6, sink);
        run<1, 5>(w, rowsmax * 8192, rows, 6, sink);
        run<1, 5>(w, rowsmax * 8192, rows, 4, sink);
        run<2, 4>(w, rowsmax * 8192, rows, 6, sink);
        run<4, 4>(w, rowsmax * 8192, rows, 6, sink);
    }
    return 0;
}
