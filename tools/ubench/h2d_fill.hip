// tools/ubench/h2d_fill.hip — does the copy engine read a page-locked buffer slower when the CPU has just FILLED it the way a decode feeder does (pread from
// the page cache: the kernel's copy leaves the lines modified in the CPU's caches)?  32 MB buffer, host-to-device copy timed by events, per way of filling:
//   old      nothing written since the last copy
//   pread    pread() of 32 MB from a file in the page cache
//   memcpy   memcpy from ordinary memory
//   nt       a copy with non-temporal stores (the lines bypass the caches)
//   pread+nt pread into a 256 KB bounce buffer, non-temporal copy from there, chunk by chunk
//   flush    pread, then clflushopt over the buffer
// each alone and with <threads> other threads pread()ing into their own buffers meanwhile (the other feeders).   h2d_fill [threads=5] [reps=20]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <immintrin.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static const size_t N = (size_t)32 << 20;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void nt_copy(void *dst, const void *src, size_t n)
{
    __m256i *d = (__m256i *)dst; const __m256i *s = (const __m256i *)src;
    for (size_t i = 0; i < n / 32; i += 4) {
        const __m256i a = _mm256_loadu_si256(s + i), b = _mm256_loadu_si256(s + i + 1), c = _mm256_loadu_si256(s + i + 2), e = _mm256_loadu_si256(s + i + 3);
        _mm256_stream_si256(d + i, a); _mm256_stream_si256(d + i + 1, b); _mm256_stream_si256(d + i + 2, c); _mm256_stream_si256(d + i + 3, e);
    }
    _mm_sfence();
}
int main(int argc, char **argv)
{
    const int n_other = argc > 1 ? atoi(argv[1]) : 5, reps = argc > 2 ? atoi(argv[2]) : 20;
    const char *path = "/tmp/h2d_fill.dat";
    const size_t FILE_N = (size_t)512 << 20;
    {   // a file in the page cache
        std::vector<char> junk(N);
        for (size_t i = 0; i < N; ++i) junk[i] = (char)(i * 2654435761u >> 13);
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        for (size_t o = 0; o < FILE_N; o += N) if (write(fd, junk.data(), N) != (ssize_t)N) return 1;
        close(fd);
    }
    (void)hipSetDevice(0);
    void *host = nullptr, *dev = nullptr; char *plain = (char *)aligned_alloc(4096, N), *bounce = (char *)aligned_alloc(4096, 256 << 10);
    if (hipHostMalloc(&host, N, hipHostMallocDefault) != hipSuccess || hipMalloc(&dev, N) != hipSuccess) return 1;
    memset(plain, 7, N); memset(host, 1, N);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int fd = open(path, O_RDONLY);
    size_t file_at = 0;
    auto next_off = [&]() { file_at = (file_at + N) % FILE_N; return (off_t)file_at; };
    auto copy_ms = [&]() { (void)hipEventRecord(e0, st); (void)hipMemcpyAsync(dev, host, N, hipMemcpyHostToDevice, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return (double)ms; };
    struct Case { const char *name; int how; };
    const Case cases[] = {{"old", 0}, {"pread", 1}, {"memcpy", 2}, {"nt", 3}, {"pread+nt (256 KB bounce)", 4}, {"pread + clflushopt", 5}};
    std::atomic<bool> stop{false};
    for (int others = 0; others <= n_other; others += n_other ? n_other : 1) {
        std::vector<std::thread> th;
        stop = false;
        for (int k = 0; k < others; ++k) th.emplace_back([&, k]() {
            char *b = (char *)aligned_alloc(4096, N); const int f = open(path, O_RDONLY); size_t at = (size_t)k * N;
            while (!stop.load()) { if (pread(f, b, N, (off_t)at) < 0) break; at = (at + N) % FILE_N; }
            close(f); free(b);
        });
        for (int w = 0; w < 3; ++w) copy_ms();
        for (const Case &c : cases) {
            std::vector<double> ms, fill;
            for (int r = 0; r < reps; ++r) {
                const double t0 = now();
                switch (c.how) {
                case 1: if (pread(fd, host, N, next_off()) != (ssize_t)N) return 1; break;
                case 2: memcpy(host, plain, N); break;
                case 3: nt_copy(host, plain, N); break;
                case 4: { const off_t o = next_off(); for (size_t at = 0; at < N; at += 256 << 10) { if (pread(fd, bounce, 256 << 10, o + (off_t)at) < 0) return 1; nt_copy((char *)host + at, bounce, 256 << 10); } break; }
                case 5: if (pread(fd, host, N, next_off()) != (ssize_t)N) return 1; for (size_t at = 0; at < N; at += 64) _mm_clflushopt((char *)host + at); _mm_sfence(); break;
                default: break;
                }
                fill.push_back((now() - t0) * 1e3);
                ms.push_back(copy_ms());
            }
            std::sort(ms.begin(), ms.end()); std::sort(fill.begin(), fill.end());
            printf("%d other readers | %-26s fill %.2f ms, host-to-device copy median %.3f ms = %.1f GB/s (min %.3f, max %.3f)\n", others, c.name, fill[fill.size() / 2], ms[ms.size() / 2],
                   N / ms[ms.size() / 2] / 1e6, ms.front(), ms.back());
        }
        stop = true;
        for (auto &t : th) t.join();
        if (!n_other) break;
    }
    unlink(path);
    return 0;
}
