// tools/ubench/pin_cost.hip — what page-locked host buffers cost a short-lived process, by how they are made: <n> buffers of <MB> each, written once, one
// host-to-device copy of each timed, then _exit; the caller's clock around the process gives what leaving costs.
//   pin_cost <mode> <n> <MB>      mode 0: hipHostMalloc   1: mmap + MADV_HUGEPAGE + hipHostRegister   2: mmap (4 KiB pages) + hipHostRegister
//                                 3: one mmap + MADV_HUGEPAGE region for all buffers, ONE hipHostRegister   4: nothing pinned (pageable memory through hipMemcpy)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const double t0 = now();
    const int mode = argc > 1 ? atoi(argv[1]) : 0, n = argc > 2 ? atoi(argv[2]) : 6;
    const size_t each = (size_t)(argc > 3 ? atoi(argv[3]) : 32) << 20;
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    void *dev = nullptr; (void)hipMalloc(&dev, each);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const double t1 = now();
    std::vector<void *> h(n, nullptr);
    double t_map = 0, t_touch = 0, t_reg = 0;
    auto region = [&](size_t bytes, bool huge) -> void * {
        const size_t al = (size_t)2 << 20;
        char *p = (char *)mmap(nullptr, bytes + al, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        char *q = (char *)(((uintptr_t)p + al - 1) / al * al);
        if (huge) (void)madvise(q, bytes, MADV_HUGEPAGE);
        return q;
    };
    if (mode == 3) {
        double a = now();
        char *q = (char *)region(each * n, true);
        t_map += now() - a; a = now();
        memset(q, 1, each * n);
        t_touch += now() - a; a = now();
        if (hipHostRegister(q, each * n, hipHostRegisterDefault) != hipSuccess) { fprintf(stderr, "register failed\n"); return 1; }
        t_reg += now() - a;
        for (int k = 0; k < n; ++k) h[k] = q + each * k;
    } else for (int k = 0; k < n; ++k) {
        double a = now();
        if (mode == 0) { if (hipHostMalloc(&h[k], each, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "hipHostMalloc failed\n"); return 1; } }
        else h[k] = region(each, mode == 1);
        t_map += now() - a; a = now();
        memset(h[k], 1, each);
        t_touch += now() - a; a = now();
        if (mode == 1 || mode == 2) if (hipHostRegister(h[k], each, hipHostRegisterDefault) != hipSuccess) { fprintf(stderr, "register failed\n"); return 1; }
        t_reg += now() - a;
    }
    const double t2 = now();
    // one copy of every buffer, twice (the second pass is the steady state)
    double c1 = 0, c2 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const double a = now();
        for (int k = 0; k < n; ++k) { if (mode == 4) (void)hipMemcpy(dev, h[k], each, hipMemcpyHostToDevice); else (void)hipMemcpyAsync(dev, h[k], each, hipMemcpyHostToDevice, st); }
        (void)hipStreamSynchronize(st);
        (pass ? c2 : c1) = now() - a;
    }
    const double t3 = now();
    long thp = 0;
    if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) { char line[256]; while (fgets(line, sizeof line, f)) if (!strncmp(line, "AnonHugePages:", 14)) thp = atol(line + 14); fclose(f); }
    fprintf(stderr, "mode %d, %d x %zu MB: runtime up %.3f s; buffers %.3f s (map/alloc %.3f, first touch %.3f, register %.3f); copies %.1f / %.1f GB/s; AnonHugePages %ld kB; in main %.3f s\n",
            mode, n, each >> 20, t1 - t0, t2 - t1, t_map, t_touch, t_reg, n * each / c1 / 1e9, n * each / c2 / 1e9, thp, t3 - t0);
    _exit(0);
}
