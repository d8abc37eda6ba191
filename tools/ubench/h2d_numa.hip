// tools/ubench/h2d_numa.hip — host-to-device copy rate by the NUMA node of the page-locked source (mmap + mbind + touch + hipHostRegister), alone and while other
// threads read a large file from the page cache (pread), which on a two-socket host crosses the sockets' link for the pages of the other node.
//   h2d_numa [file to read beside the copies] [readers]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static long mbind_node(void *p, size_t len, int node)
{
    unsigned long mask[16] = {0}; mask[node / 64] = 1ul << (node % 64);
    return syscall(SYS_mbind, p, len, 2 /* MPOL_BIND */, mask, 1024ul, 0u);
}
int main(int argc, char **argv)
{
    (void)hipSetDevice(0);
    char bus[64] = {0}; (void)hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    int gpu_node = -2; if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &gpu_node) != 1) gpu_node = -2; fclose(f); }
    printf("GPU %s: numa_node %d; this thread runs on cpu %d\n", bus, gpu_node, sched_getcpu());
    const size_t N = (size_t)32 << 20;
    void *dev = nullptr; (void)hipMalloc(&dev, N);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::atomic<bool> stop{false};
    const int n_readers = argc > 2 ? atoi(argv[2]) : 5;
    for (int load = 0; load < (argc > 1 ? 2 : 1); ++load) {
        std::vector<std::thread> th;
        stop = false;
        if (load) for (int k = 0; k < n_readers; ++k) th.emplace_back([&, k]() {
            const int fd = open(argv[1], O_RDONLY); if (fd < 0) return;
            const off_t fs = lseek(fd, 0, SEEK_END); char *buf = nullptr; if (getenv("PINNED_READERS")) { (void)hipSetDevice(0); (void)hipHostMalloc((void **)&buf, N, hipHostMallocDefault); } else buf = (char *)aligned_alloc(4096, N); off_t at = (off_t)k * (fs / n_readers);
            while (!stop.load()) { if (pread(fd, buf, N, at) <= 0) at = 0; at += N; if (at + (off_t)N > fs) at = 0; }
            close(fd);
        });
        for (int node = 0; node < 2; ++node) {
            void *p = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            const long rc = mbind_node(p, N, node);
            memset(p, 3, N);
            if (hipHostRegister(p, N, hipHostRegisterDefault) != hipSuccess) { printf("node %d: register failed\n", node); continue; }
            std::vector<float> ms;
            for (int r = 0; r < 30; ++r) { (void)hipEventRecord(a, st); (void)hipMemcpyAsync(dev, p, N, hipMemcpyHostToDevice, st); (void)hipEventRecord(b, st); (void)hipEventSynchronize(b); float x = 0; (void)hipEventElapsedTime(&x, a, b); if (r >= 6) ms.push_back(x); }
            std::sort(ms.begin(), ms.end());
            printf("%s source bound to node %d (mbind rc %ld): 32 MiB copy median %.3f ms = %.1f GB/s (min %.3f, max %.3f)\n", load ? "beside the readers," : "alone,", node, rc, ms[ms.size() / 2],
                   N / ms[ms.size() / 2] / 1e6, ms.front(), ms.back());
            (void)hipHostUnregister(p); munmap(p, N);
        }
        stop = true; for (auto &t : th) t.join();
    }
    return 0;
}
