// tools/ubench/exit_cost.hip — what a process pays at exit for the device memory it holds: allocates <GB> of device memory in <pieces> pieces, touches it
// (or not), optionally frees it, prints the time since start and leaves with _exit; the caller's clock around the process gives the rest.
//   exit_cost <GB> <pieces> <touch 0|1> <free 0|1> [pinned host MB, in 32 MB buffers] [streams]
#include <hip/hip_runtime.h>
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
    const double gb = argc > 1 ? atof(argv[1]) : 0;
    const int pieces = argc > 2 ? atoi(argv[2]) : 1, touch = argc > 3 ? atoi(argv[3]) : 1, do_free = argc > 4 ? atoi(argv[4]) : 0;
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    const double t1 = now();
    std::vector<void *> p;
    const size_t each = (size_t)(gb * 1e9 / (pieces > 0 ? pieces : 1));
    for (int k = 0; k < pieces && each; ++k) { void *q = nullptr; if (hipMalloc(&q, each) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; } p.push_back(q); }
    const int pinned_mb = argc > 5 ? atoi(argv[5]) : 0, n_streams = argc > 6 ? atoi(argv[6]) : 0;
    std::vector<void *> h; std::vector<hipStream_t> st;
    for (int k = 0; k < pinned_mb / 32; ++k) { void *q = nullptr; if (hipHostMalloc(&q, (size_t)32 << 20, hipHostMallocDefault) == hipSuccess) { memset(q, 1, (size_t)32 << 20); h.push_back(q); } }
    for (int k = 0; k < n_streams; ++k) { hipStream_t x; if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) == hipSuccess) { st.push_back(x); if (!p.empty() && !h.empty()) (void)hipMemcpyAsync(p[0], h[k % h.size()], 1 << 20, hipMemcpyHostToDevice, x); } }
    if (touch) for (void *q : p) (void)hipMemsetAsync(q, 1, each, nullptr);
    (void)hipDeviceSynchronize();
    const double t2 = now();
    if (do_free) for (void *q : p) (void)hipFree(q);
    const double t3 = now();
    fprintf(stderr, "runtime up %.3f s, alloc%s %.3f s, free %.3f s, in main %.3f s\n", t1 - t0, touch ? " + touch" : "", t2 - t1, t3 - t2, t3 - t0);
    _exit(0);
}
