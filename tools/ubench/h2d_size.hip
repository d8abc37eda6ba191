// tools/ubench/h2d_size.hip — host-to-device copy rate by size (page-locked source, one stream, events): is there a step above 32 MiB?  Also from SIX rotating source
// buffers (what the decode's readers use) instead of one.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
int main()
{
    (void)hipSetDevice(0);
    const size_t cap = (size_t)72 << 20;
    void *h[6], *d[6];
    for (int k = 0; k < 6; ++k) { if (hipHostMalloc(&h[k], cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&d[k], cap) != hipSuccess) return 1; memset(h[k], k + 1, cap); }
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const double mb[] = {8, 16, 24, 30, 31.9, 32, 32.1, 33, 33.7, 36, 40, 48, 64};
    for (int rot = 0; rot < 2; ++rot)
        for (double m : mb) {
            const size_t n = (size_t)(m * 1048576.0);
            std::vector<float> ms;
            for (int r = 0; r < 24; ++r) {
                const int k = rot ? r % 6 : 0;
                (void)hipEventRecord(a, st); (void)hipMemcpyAsync(d[k], h[k], n, hipMemcpyHostToDevice, st); (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
                float x = 0; (void)hipEventElapsedTime(&x, a, b); if (r >= 6) ms.push_back(x);
            }
            std::sort(ms.begin(), ms.end());
            printf("%s, %.1f MiB: median %.3f ms = %.1f GB/s (min %.3f, max %.3f)\n", rot ? "six buffers in turn" : "one buffer", m, ms[ms.size() / 2], n / ms[ms.size() / 2] / 1e6, ms.front(), ms.back());
        }
    return 0;
}
