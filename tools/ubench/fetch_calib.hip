// tools/ubench/fetch_calib.hip — calibration of rocprofv3 FETCH_SIZE for the two load shapes the
// scatter kernel uses: 12 B/lane struct loads (global_load_dwordx3, the run stream) and 16 B/lane.
// Run under `rocprofv3 --pmc FETCH_SIZE`; each kernel reads exactly N bytes once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct iv3 { int32_t a, b, c; };
__global__ void k_calib_read12(const iv3* p, size_t n, int* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; int acc = 0;
    for (; i < n; i += st) { iv3 v = p[i]; acc += v.a + v.b + v.c; }
    if (acc == 0x7fffffff) *out = acc;
}
__global__ void k_calib_read16(const int4* p, size_t n, int* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; int acc = 0;
    for (; i < n; i += st) { int4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x7fffffff) *out = acc;
}
int main() {
    const size_t bytes = (size_t)12 << 30;           // 12 GiB, far beyond the 256 MiB Infinity Cache
    void* p; int* o; hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 1, bytes);
    for (int r = 0; r < 2; ++r) {
        k_calib_read12<<<8192, 256>>>((const iv3*)p, bytes / 12, o);
        k_calib_read16<<<8192, 256>>>((const int4*)p, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes\n", bytes);
    return 0;
}
