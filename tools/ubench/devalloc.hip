// tools/ubench/devalloc.hip — what device allocations cost on this box: hipMalloc / hipMemsetAsync / hipFree of 0.25 .. 4 GB, alone and while a
// kernel-launching thread runs beside them (does an allocation stall launches?).  hipcc --offload-arch=gfx950 -O2 devalloc.hip -o devalloc
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(int *p, int n) { int v = 0; for (int i = 0; i < n; ++i) v += i ^ (v >> 3); if (v == 123456789) *p = v; }
int main()
{
    (void)hipSetDevice(0);
    void *warm; (void)hipMalloc(&warm, 1 << 20);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int pass = 0; pass < 2; ++pass) {
        std::atomic<bool> stop{false};
        std::atomic<long> launches{0};
        std::thread bg;
        if (pass == 1) bg = std::thread([&] {
            (void)hipSetDevice(0);
            hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            int *d; (void)hipMalloc(&d, 64);
            while (!stop.load()) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s2, d, 2000); (void)hipStreamSynchronize(s2); ++launches; }
        });
        for (double gb : {0.25, 1.0, 2.0, 4.0}) {
            const size_t n = (size_t)(gb * (1u << 30));
            void *p = nullptr;
            const long l0 = launches.load();
            double t0 = now(); hipError_t e = hipMalloc(&p, n); double t1 = now();
            (void)hipMemsetAsync(p, 0, n, st); (void)hipStreamSynchronize(st); double t2 = now();
            (void)hipMemsetAsync(p, 0, n, st); (void)hipStreamSynchronize(st); double t3 = now();
            (void)hipFree(p); double t4 = now();
            printf("%s %.2f GB: hipMalloc %.1f ms (rc %d), first memset %.1f ms, second memset %.1f ms, hipFree %.1f ms; launches by the other thread meanwhile: %ld in %.1f ms\n",
                   pass ? "with a launching thread" : "alone", gb, (t1 - t0) * 1e3, (int)e, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, launches.load() - l0, (t4 - t0) * 1e3);
        }
        if (pass == 1) {
            const long l0 = launches.load(); const double t0 = now();
            std::this_thread::sleep_for(std::chrono::milliseconds(200));
            printf("the launching thread alone: %ld launches in %.1f ms\n", launches.load() - l0, (now() - t0) * 1e3);
            stop = true; bg.join();
        }
    }
    return 0;
}
