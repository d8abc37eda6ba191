// how long pinned host memory takes to get (hipHostMalloc vs hipHostRegister of touched pages) and what it buys for copies
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(0);
    void *d; hipMalloc(&d, 512u << 20);
    for (size_t mb : {64, 256}) {
        const size_t n = mb << 20;
        double t0 = now(); void *h; hipHostMalloc(&h, n, hipHostMallocDefault); double t1 = now();
        memset(h, 1, n); double t2 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t3 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t4 = now();
        hipHostFree(h); double t5 = now();
        printf("%zu MB: hipHostMalloc %.4f s, first touch %.4f, H2D %.4f (%.1f GB/s), D2H %.4f (%.1f GB/s), free %.4f\n", mb, t1 - t0, t2 - t1, t3 - t2, n / (t3 - t2) / 1e9, t4 - t3, n / (t4 - t3) / 1e9, t5 - t4);
        char *p = (char *)malloc(n); memset(p, 1, n);
        t0 = now(); hipMemcpy(d, p, n, hipMemcpyHostToDevice); t1 = now(); hipMemcpy(p, d, n, hipMemcpyDeviceToHost); t2 = now();
        printf("%zu MB pageable: H2D %.4f (%.1f GB/s), D2H %.4f (%.1f GB/s)\n", mb, t1 - t0, n / (t1 - t0) / 1e9, t2 - t1, n / (t2 - t1) / 1e9);
        t0 = now(); hipHostRegister(p, n, hipHostRegisterDefault); t1 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice); t2 = now(); hipMemcpy(p, d, n, hipMemcpyDeviceToHost); t3 = now();
        hipHostUnregister(p); t4 = now();
        printf("%zu MB registered: register %.4f s, H2D %.4f (%.1f GB/s), D2H %.4f (%.1f GB/s), unregister %.4f\n", mb, t1 - t0, t2 - t1, n / (t2 - t1) / 1e9, t3 - t2, n / (t3 - t2) / 1e9, t4 - t3);
        free(p);
    }
    return 0;
}
