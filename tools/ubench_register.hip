// ubench_register.hip -- what does hipHostRegister of a pageable buffer cost, and how fast are a linear and a pitched
// (rows of n bytes -> device pitch ld) copy from it afterwards?   hipcc -O2 -o tools/ubench_register tools/ubench_register.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (auto shape : {std::pair<int, int>{2000, 10000}, {1000, 4000}, {5000, 5000}}) {
        const int m = shape.first, n = shape.second, ld = (n + 63) / 64 * 64;
        const size_t bytes = (size_t)m * n;
        unsigned char *h = (unsigned char *)malloc(bytes), *d;
        memset(h, 65, bytes);
        hipMalloc(&d, (size_t)m * ld);
        double t0 = now();
        hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
        double t1 = now();
        printf("%d x %d (%.1f MB): hipHostRegister %.3f ms (%s)\n", m, n, bytes / 1048576.0, t1 - t0, hipGetErrorString(e));
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now();
            for (int i = 0; i < 10; ++i) hipMemcpy2DAsync(d, ld, h, n, n, m, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            t1 = now();
            if (rep == 2) printf("   pitched copy %.3f ms  %.1f GB/s\n", (t1 - t0) / 10, bytes / ((t1 - t0) / 10) / 1e6);
            t0 = now();
            for (int i = 0; i < 10; ++i) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            t1 = now();
            if (rep == 2) printf("   linear copy  %.3f ms  %.1f GB/s\n", (t1 - t0) / 10, bytes / ((t1 - t0) / 10) / 1e6);
        }
        // one upload at a time with a wait each (what a trim does)
        t0 = now();
        for (int i = 0; i < 10; ++i) { hipMemcpy2DAsync(d, ld, h, n, n, m, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
        t1 = now();
        printf("   pitched copy + wait, one at a time %.3f ms\n", (t1 - t0) / 10);
        t0 = now();
        hipHostUnregister(h);
        t1 = now();
        printf("   hipHostUnregister %.3f ms\n", t1 - t0);
        // pageable, no registration: what the runtime does by itself
        t0 = now();
        for (int i = 0; i < 5; ++i) { hipMemcpy2DAsync(d, ld, h, n, n, m, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
        t1 = now();
        printf("   pageable pitched copy (runtime staging) %.3f ms\n", (t1 - t0) / 5);
        hipFree(d);
        free(h);
    }
    return 0;
}
