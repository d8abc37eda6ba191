// pd_format.hip — per-site rows on the device (include/pandepth_amd.h: pd_format_sites).  The reference writes one line per
// base, "<contig>\t<0-based index>\t<depth>\n" (PD:4278-4281); the host used to read the cells back (4 B each) and format
// them on its threads.  Here the text itself is produced in HBM — lengths, an exclusive scan of the lengths, the bytes — and
// what crosses PCIe is the text the gzip stage consumes.  HBM-bound byte work: 4 B read + ~15 B written per cell.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pd_kernels.h"

namespace pdk {
namespace {
constexpr int FWG = 256;                 // threads per workgroup
constexpr int FPER = 16;                 // consecutive cells per thread
constexpr uint32_t FBLK = FWG * FPER;    // cells per workgroup

__device__ __forceinline__ uint32_t dec_digits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}
// writes v in decimal, `nd` digits, ending just before p_end
__device__ __forceinline__ void put_dec(char *p_end, uint32_t v, uint32_t nd)
{
    for (uint32_t k = 0; k < nd; ++k) { const uint32_t q = v / 10u; *--p_end = (char)('0' + (v - q * 10u)); v = q; }
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

// bytes of the rows of this thread's cells; the workgroup's exclusive offsets
template <bool WRITE>
__global__ __launch_bounds__(FWG) void k_site_rows(const uint32_t *depth, uint32_t first_index, uint64_t n, uint32_t name_len,
                                                  const char *name, uint32_t *blk_bytes, const uint64_t *blk_off, char *text)
{
    __shared__ uint32_t wsum[FWG / 64];
    const uint64_t c0 = (uint64_t)blockIdx.x * FBLK + (uint64_t)threadIdx.x * FPER;
    uint32_t d[FPER], len[FPER], mine = 0;
#pragma unroll
    for (int k = 0; k < FPER; ++k) {
        const uint64_t c = c0 + k;
        d[k] = c < n ? depth[c] : 0u;
        len[k] = c < n ? name_len + 3u + dec_digits(first_index + (uint32_t)c) + dec_digits(d[k]) : 0u;
        mine += len[k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    if (!WRITE) {
        if (threadIdx.x == FWG - 1) blk_bytes[blockIdx.x] = base + mine;
        return;
    }
    char *p = text + blk_off[blockIdx.x] + base;
#pragma unroll 1
    for (int k = 0; k < FPER; ++k) {
        if (!len[k]) break;
        for (uint32_t j = 0; j < name_len; ++j) p[j] = name[j];
        const uint32_t idx = first_index + (uint32_t)(c0 + k);
        const uint32_t n1 = dec_digits(idx), n2 = dec_digits(d[k]);
        char *q = p + name_len;
        *q++ = '\t'; q += n1; put_dec(q, idx, n1);
        *q++ = '\t'; q += n2; put_dec(q, d[k], n2);
        *q = '\n';
        p += len[k];
    }
}

// exclusive scan of the workgroups' byte counts (one workgroup; a few thousand entries at most per call)
__global__ __launch_bounds__(FWG) void k_site_scan(const uint32_t *blk_bytes, uint32_t n_blk, uint64_t *blk_off)
{
    __shared__ unsigned long long wsum[FWG / 64];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blk; b0 += FWG) {
        const uint32_t i = b0 + threadIdx.x;
        const unsigned long long v = i < n_blk ? blk_bytes[i] : 0ull;
        unsigned long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(x, o); if ((int)(threadIdx.x & 63) >= o) x += y; }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        unsigned long long base = carry;
        for (int k = 0; k < wv; ++k) base += wsum[k];
        if (i < n_blk) blk_off[i] = base + x - v;
        __syncthreads();
        if (threadIdx.x == FWG - 1) carry = base + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) blk_off[n_blk] = carry;
}
} // namespace

uint32_t site_rows_blocks(uint64_t n) { return (uint32_t)((n + FBLK - 1) / FBLK); }

void launch_site_rows(hipStream_t st, const uint32_t *depth, uint32_t first_index, uint64_t n, uint32_t name_len, const char *dev_name,
                      uint32_t *blk_bytes, uint64_t *blk_off, char *text, bool write)
{
    const uint32_t nb = site_rows_blocks(n);
    if (!nb) return;
    if (!write) {
        hipLaunchKernelGGL((k_site_rows<false>), dim3(nb), dim3(FWG), 0, st, depth, first_index, n, name_len, dev_name, blk_bytes, blk_off, text);
        hipLaunchKernelGGL(k_site_scan, dim3(1), dim3(FWG), 0, st, blk_bytes, nb, blk_off);
    } else {
        hipLaunchKernelGGL((k_site_rows<true>), dim3(nb), dim3(FWG), 0, st, depth, first_index, n, name_len, dev_name, blk_bytes, blk_off, text);
    }
}
} // namespace pdk
