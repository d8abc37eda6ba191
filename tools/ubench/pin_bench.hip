// how long does it take to get 36 MB of pinned host memory?  hipHostMalloc vs transparent-huge-page backed memory + hipHostRegister
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double t0 = now(); hipFree(0); printf("runtime init %.1f ms\n", (now() - t0) * 1e3);
    const size_t N = (size_t)36 << 20;
    for (int rep = 0; rep < 3; ++rep) {
        void *p = nullptr; t0 = now(); hipHostMalloc(&p, N, hipHostMallocDefault); const double a = now() - t0;
        t0 = now(); memset(p, 1, N); const double b = now() - t0;
        t0 = now(); hipHostFree(p); const double c = now() - t0;
        printf("hipHostMalloc %.1f ms, first touch %.1f ms, free %.1f ms\n", a * 1e3, b * 1e3, c * 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        t0 = now();
        void *p = mmap(nullptr, N + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char *q = (char *)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        madvise(q, N, MADV_HUGEPAGE);
        const double a = now() - t0;
        t0 = now(); for (size_t i = 0; i < N; i += 4096) q[i] = 1; const double b = now() - t0;
        t0 = now(); hipError_t e = hipHostRegister(q, N, hipHostRegisterDefault); const double c = now() - t0;
        void *d = nullptr; hipMalloc(&d, N);
        t0 = now(); hipMemcpy(d, q, N, hipMemcpyHostToDevice); const double h = now() - t0;
        t0 = now(); hipHostUnregister(q); munmap(p, N + (2 << 20)); const double f = now() - t0;
        printf("mmap+madvise %.2f ms, touch %.1f ms, hipHostRegister %.1f ms (%s), H2D %.1f ms = %.1f GB/s, unregister+unmap %.1f ms\n", a * 1e3, b * 1e3, c * 1e3, hipGetErrorString(e), h * 1e3, N / h / 1e9, f * 1e3);
        hipFree(d);
    }
    return 0;
}
